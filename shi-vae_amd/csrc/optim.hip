// Small streaming kernels around the step: layout shuffles between the padded scan records and the
// compact decoder input, the global gradient norm and the fused clip + Adam update
// (trainers.py:222-224: clip_grad_norm_(params, clip) then optimizer.step()).
#include <algorithm>
#include <cmath>

#include "common.h"

namespace shivae {

// embc[r][0:H | H:H+K | H+K:E] = emb[r][0:H | Hp:Hp+K | Hp+Kp:Hp+Kp+H]   (decoders.py:127 cat order)
__global__ void compact_emb_kernel(const float* __restrict__ emb, int Ep, int Hp, int Kp, int H, int K,
                                   float* __restrict__ embc, int ldc, int rows) {
    const int E = 2 * H + K;
    const size_t n = (size_t)rows * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / E;
        const int c = (int)(i % E);
        const int sc = c < H ? c : (c < H + K ? Hp + (c - H) : Hp + Kp + (c - H - K));
        embc[r * ldc + c] = emb[r * Ep + sc];
    }
}

__global__ void expand_add_emb_kernel(const float* __restrict__ dembc, int ldc, int H, int K,
                                      float* __restrict__ d_emb, int Ep, int Hp, int Kp, int rows) {
    const int E = 2 * H + K;
    const size_t n = (size_t)rows * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / E;
        const int c = (int)(i % E);
        const int sc = c < H ? c : (c < H + K ? Hp + (c - H) : Hp + Kp + (c - H - K));
        d_emb[r * Ep + sc] += dembc[r * ldc + c];
    }
}

// dst[r][dcol0 : dcol0+width] = src[r][col0 : col0+width]   (row strides lds / ldd)
__global__ void copy2d_kernel(const float* __restrict__ src, int lds, int col0, int width, float* __restrict__ dst,
                              int ldd, int dcol0, size_t rows) {
    const size_t n = rows * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / width;
        const int c = (int)(i % width);
        dst[r * ldd + dcol0 + c] = src[r * lds + col0 + c];
    }
}

// out[b][c][r] = in[b][r][c]: batched 2-D transpose through a 32x33 LDS tile (coalesced both ways).
// Used once per step to keep K-contiguous copies of the decoder weights, so that every input-gradient
// GEMM runs in the NT form (both operands K-contiguous), which is the fastest layout of gemm.hip.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C, size_t in_stride, size_t out_stride) {
    __shared__ float tile[32][33];
    const float* src = in + blockIdx.z * in_stride;
    float* dst = out + blockIdx.z * out_stride;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(size_t)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

int launch_transpose(const float* in, float* out, int R, int C, int batch, size_t in_stride, size_t out_stride,
                     hipStream_t stream) {
    if (R <= 0 || C <= 0 || batch <= 0) return SHIVAE_OK;
    SV_REQUIRE(batch <= 65535 && ceil_div(R, 32) <= 65535, "transpose grid too large");
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32), batch), dim3(256), 0, stream, in, out,
                       R, C, in_stride, out_stride);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

static int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 256 * 16); }

int launch_compact_emb(const float* emb, int Ep, int Hp, int Kp, int H, int K, float* embc, int ldc, int rows,
                       hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    hipLaunchKernelGGL(compact_emb_kernel, dim3(grid_for((size_t)rows * (2 * H + K))), dim3(256), 0, stream, emb,
                       Ep, Hp, Kp, H, K, embc, ldc, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_expand_add_emb(const float* dembc, int ldc, int H, int K, float* d_emb, int Ep, int Hp, int Kp, int rows,
                          hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    hipLaunchKernelGGL(expand_add_emb_kernel, dim3(grid_for((size_t)rows * (2 * H + K))), dim3(256), 0, stream,
                       dembc, ldc, H, K, d_emb, Ep, Hp, Kp, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_copy2d(const float* src, int lds, int col0, int width, float* dst, int ldd, int dcol0, int64_t rows,
                  hipStream_t stream) {
    if (rows <= 0 || width <= 0) return SHIVAE_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((size_t)rows * width)), dim3(256), 0, stream, src, lds, col0,
                       width, dst, ldd, dcol0, (size_t)rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_unpad_copy(const float* src, int ld, int col0, int width, float* dst, int64_t rows, hipStream_t stream) {
    return launch_copy2d(src, ld, col0, width, dst, width, 0, rows, stream);
}

// ---------------------------------------------------------------- global gradient norm (deterministic)
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n,
                                                             float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int nparts,
                                                           float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

int launch_sqnorm(const float* buf, int64_t n, float* scratch, float* out, hipStream_t stream) {
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256));
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(parts), dim3(256), 0, stream, buf, (size_t)n, scratch);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, stream, scratch, parts, out);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---- the same reduction for callers that cannot allocate (shivae_sqnorm_noalloc): the block partials live in a
// file-scope device array instead of a caller-provided buffer.  Every GPU holds its own copy of the array at its
// own address; the host-side name is only a handle for hipGetSymbolAddress, never a kernel argument.
static __device__ __align__(256) float scratch[1024];
static constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices] = {nullptr};

// Set-up for the CURRENT device: resolve its copy of the array once.  Called by shivae_init and shivae_plan_bind
// (both outside any stream capture); the entry point itself only falls back to it on a device neither has seen.
int init_sqnorm_noalloc() {
    int dev = -1;
    SV_CHECK_HIP(hipGetDevice(&dev));
    SV_REQUIRE(dev >= 0 && dev < kMaxDevices, "sqnorm_noalloc: device ordinal %d out of range", dev);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        SV_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        SV_REQUIRE(ptr != nullptr, "sqnorm_noalloc: no device address for the scratch array");
        g_scratch_ptr[dev] = static_cast<float*>(ptr);
    }
    return SHIVAE_OK;
}

int launch_sqnorm_noalloc(const float* buf, int64_t n, float* out, hipStream_t stream) {
    int dev = -1;
    SV_CHECK_HIP(hipGetDevice(&dev));
    SV_REQUIRE(dev >= 0 && dev < kMaxDevices, "sqnorm_noalloc: device ordinal %d out of range", dev);
    if (stream) {  // the array is per device: the stream must belong to the current one
        int sdev = dev;
        if (hipStreamGetDevice(stream, &sdev) == hipSuccess)
            SV_REQUIRE(sdev == dev, "sqnorm_noalloc: stream is on device %d, current device is %d", sdev, dev);
    }
    if (!g_scratch_ptr[dev]) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        SV_REQUIRE(cap == hipStreamCaptureStatusNone,
                   "sqnorm_noalloc: first use on device %d is inside a stream capture; call shivae_init() (or bind a "
                   "plan) on this device beforehand", dev);
        SV_TRY(init_sqnorm_noalloc());
    }
    // all `parts` partials are overwritten before the final kernel reads them: nothing has to start at zero
    return launch_sqnorm(buf, n, g_scratch_ptr[dev], out, stream);
}

// clip coefficient = min(clip / (norm + 1e-6), 1)  (torch clip_grad_norm_), then Adam (no weight
// decay, no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, const float* __restrict__ sqnorm, float clip,
                                 float step_size, float inv_sqrt_bc2, float b1, float b2, float eps) {
    const float norm = sqrtf(sqnorm[0]);
    const float coef = clip > 0.f ? fminf(clip / (norm + 1e-6f), 1.f) : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

int launch_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm, float clip,
                     float lr, float b1, float b2, float eps, int step, hipStream_t stream) {
    if (n <= 0) return SHIVAE_OK;
    SV_REQUIRE(step >= 1, "clip_adam: step must be >= 1");
    const double bc1 = 1.0 - std::pow((double)b1, step), bc2 = 1.0 - std::pow((double)b2, step);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, stream, p, g, m, v, (size_t)n,
                       sqnorm, clip, (float)(lr / bc1), (float)(1.0 / std::sqrt(bc2)), b1, b2, eps);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
