// Device-side pre/post pass either side of the train step: HeterogeneousScaler
// transform / inverse_transform (lib/scalers.py:85-183, 223-323) fused with Trainer.preprocess_batch
// (models/train.py:333-362: B,T,D -> T,B,X, float32 / bool) and utils.zero_fill (lib/utils.py:89-100).
// HBM-streaming, one thread per output element; writes are fully coalesced, reads of the narrower raw
// rows are served by L1/L2.
#include <algorithm>

#include "common.h"

namespace shivae {

struct ScaleParams {
    const void* in;
    const uint8_t* mask_in;
    const shivae_scale_attr* attrs;
    const int* col2attr;
    void* out;
    uint8_t* mask_out;
    long long total;
    int N, T, D, X, flags;
};

// Staged in LDS once per workgroup: per OUTPUT column one packed word (type | one-hot class << 2 | attribute
// << 16), per attribute its moments, so the only global accesses per element are the raw value, its mask
// byte and the two stores.
struct AttrLds {
    double mean, std;
    float inv;  // 1 / std
    int index;
};

// raw data[N][T][D] -> x[T][N][X] (flag TRANSPOSE) or [N][T][X]; mask replicated over one-hot columns.
// F64: float64 input scaled in float64 (bit-level match of the reference's numpy pass up to the final
// cast); float32 input: the subtraction of the mean is still done in float64 (no cancellation error),
// the rest in float32.  One wave per (n, t) row.  (Measured: instruction-issue bound at ~2.4 TB/s, not HBM
// bound; a 4-columns-per-lane variant with 16-B stores was slower.)
template <bool F64>
__global__ __launch_bounds__(256) void scale_transform_kernel(ScaleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
    AttrLds* al = reinterpret_cast<AttrLds*>(prep_smem);
    uint32_t* col = reinterpret_cast<uint32_t*>(al + p.D);
    for (int d = threadIdx.x; d < p.D; d += blockDim.x) {
        const shivae_scale_attr a = p.attrs[d];
        al[d] = AttrLds{a.mean, a.std, (float)(1.0 / a.std), a.index};
    }
    for (int xc = threadIdx.x; xc < p.X; xc += blockDim.x) {
        const int d = p.col2attr[xc];
        const shivae_scale_attr a = p.attrs[d];
        col[xc] = (uint32_t)a.type | ((uint32_t)(xc - a.xoff) << 2) | ((uint32_t)d << 16);
    }
    __syncthreads();
    const float* __restrict__ in32 = static_cast<const float*>(p.in);
    const double* __restrict__ in64 = static_cast<const double*>(p.in);
    const uint8_t* __restrict__ mi = p.mask_in;
    float* __restrict__ out = static_cast<float*>(p.out);
    uint8_t* __restrict__ mo = p.mask_out;
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)p.N * p.T, wstride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long nt = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); nt < rows; nt += wstride) {
        long long obase = nt * p.X;
        if (p.flags & SHIVAE_SCALE_TRANSPOSE) {
            const long long n = nt / p.T, t = nt % p.T;
            obase = (t * p.N + n) * p.X;
        }
        const long long ibase = nt * p.D;
#pragma unroll 4
        for (int xc = lane; xc < p.X; xc += 64) {
            const uint32_t pk = col[xc];
            const int type = pk & 3, cls = (pk >> 2) & 0x3fff;
            const AttrLds a = al[pk >> 16];
            float o;
            bool nan;
            if (F64) {  // the reference scales in float64 and casts to float32 afterwards (train.py:357)
                const double v = in64[ibase + a.index];
                nan = v != v;
                double od;
                if (type == SHIVAE_REAL) od = (v - a.mean) / a.std;                  // scalers.py:238-240
                else if (type == SHIVAE_POS) od = (log(v + 1.0) - a.mean) / a.std;   // scalers.py:271-275
                else if (type == SHIVAE_BIN) od = v;                                 // scalers.py:50-51 (no scaler)
                else od = nan ? v : ((int)v == cls ? 1.0 : 0.0);                     // scalers.py:308-315 (categories 0..R-1)
                o = (float)od;
            } else {
                const float v = in32[ibase + a.index];
                nan = v != v;
                if (type == SHIVAE_REAL) o = (float)((double)v - a.mean) * a.inv;
                else if (type == SHIVAE_POS) o = (float)((double)logf(v + 1.f) - a.mean) * a.inv;
                else if (type == SHIVAE_BIN) o = v;
                else o = nan ? v : ((int)v == cls ? 1.f : 0.f);
            }
            const bool m = mi ? mi[ibase + a.index] != 0 : nan;           // scalers.py:98-99
            if ((p.flags & SHIVAE_SCALE_ZERO_FILL) && m) o = 0.f;          // utils.py:89-100
            out[obase + xc] = o;
            mo[obase + xc] = m ? 1 : 0;
        }
    }
}

// x[T][N][X] (flag TRANSPOSE) or [N][T][X] -> data[N][T][D]; mask_out = mask of the attribute's FIRST column.
// One wave per (n, t) row, lanes stride the attributes.
__global__ __launch_bounds__(256) void scale_inverse_kernel(ScaleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
    shivae_scale_attr* at = reinterpret_cast<shivae_scale_attr*>(prep_smem);
    for (int d = threadIdx.x; d < p.D; d += blockDim.x) at[d] = p.attrs[d];
    __syncthreads();
    const float* __restrict__ in = static_cast<const float*>(p.in);
    const uint8_t* __restrict__ mi = p.mask_in;
    uint8_t* __restrict__ mo = p.mask_out;
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)p.N * p.T, wstride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long nt = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); nt < rows; nt += wstride) {
        long long row = nt;
        if (p.flags & SHIVAE_SCALE_TRANSPOSE) {
            const long long n = nt / p.T, t = nt % p.T;
            row = t * p.N + n;
        }
#pragma unroll 2
        for (int d = lane; d < p.D; d += 64) {
            const shivae_scale_attr a = at[d];
            const float* xr = in + row * p.X + a.xoff;
            double o;
            if (a.type == SHIVAE_REAL) o = (double)xr[0] * a.std + a.mean;                 // scalers.py:242-245
            else if (a.type == SHIVAE_POS) {                                                // scalers.py:277-283
                const double e = (double)xr[0] * a.std + a.mean;
                o = (p.flags & SHIVAE_SCALE_F64) ? exp(e) - 1.0 : (double)(expf((float)e) - 1.f);
            } else if (a.type == SHIVAE_BIN) o = xr[0];
            else {                                                                          // scalers.py:317-323
                int best = 0;
                float vb = xr[0];
                bool anynan = vb != vb;  // np.argmax returns the first NaN if there is one
                for (int c = 1; c < a.dim; ++c) {
                    const float v = xr[c];
                    if (!anynan && (v != v)) { best = c; anynan = true; }
                    else if (!anynan && v > vb) { vb = v; best = c; }
                }
                const float last = xr[a.dim - 1];
                o = (last != last) ? (double)last : (double)best;
            }
            if (p.flags & SHIVAE_SCALE_F64) static_cast<double*>(p.out)[nt * p.D + a.index] = o;
            else static_cast<float*>(p.out)[nt * p.D + a.index] = (float)o;
            if (mo) mo[nt * p.D + a.index] = mi ? mi[row * p.X + a.xoff] : (xr[0] != xr[0]);
        }
    }
}

static int launch_scale(bool inverse, const ScaleParams& p, hipStream_t stream) {
    if (p.total <= 0) return SHIVAE_OK;
    const long long rows = (long long)p.N * p.T;
    const int blocks = (int)std::min<long long>((rows + 3) / 4, 256 * 8);
    const size_t lds = inverse ? sizeof(shivae_scale_attr) * (size_t)p.D : sizeof(AttrLds) * (size_t)p.D + 4 * (size_t)p.X;
    SV_REQUIRE(p.D < 65536 && p.X - p.D < (1 << 14), "scale: too many attributes / classes");
    SV_REQUIRE(lds <= 160 * 1024, "scale: %d columns exceed the LDS descriptor table", inverse ? p.D : p.X);
    if (lds > 64 * 1024) {
        const void* k = inverse ? reinterpret_cast<const void*>(scale_inverse_kernel)
                                : ((p.flags & SHIVAE_SCALE_F64) ? reinterpret_cast<const void*>(scale_transform_kernel<true>)
                                                                : reinterpret_cast<const void*>(scale_transform_kernel<false>));
        SV_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (inverse) hipLaunchKernelGGL(scale_inverse_kernel, dim3(blocks), dim3(256), lds, stream, p);
    else if (p.flags & SHIVAE_SCALE_F64) hipLaunchKernelGGL(scale_transform_kernel<true>, dim3(blocks), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(scale_transform_kernel<false>, dim3(blocks), dim3(256), lds, stream, p);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae

using namespace shivae;

extern "C" {

int shivae_scale_transform(const void* data_dev, const uint8_t* mask_dev, int n, int t, int n_attr, int x_dim,
                           const shivae_scale_attr* attrs_dev, const int* col2attr_dev, float* x_dev,
                           uint8_t* mask_out_dev, int flags, shivae_stream stream) {
    SV_REQUIRE(data_dev && attrs_dev && col2attr_dev && x_dev && mask_out_dev, "scale_transform: null argument");
    SV_REQUIRE(n > 0 && t > 0 && n_attr > 0 && x_dim >= n_attr, "scale_transform: bad sizes");
    ScaleParams p{data_dev, mask_dev, attrs_dev, col2attr_dev, x_dev, mask_out_dev, (long long)n * t * x_dim,
                  n, t, n_attr, x_dim, flags};
    return launch_scale(false, p, static_cast<hipStream_t>(stream));
}

int shivae_scale_inverse(const float* x_dev, const uint8_t* mask_dev, int n, int t, int n_attr, int x_dim,
                         const shivae_scale_attr* attrs_dev, void* data_dev, uint8_t* mask_out_dev, int flags,
                         shivae_stream stream) {
    SV_REQUIRE(x_dev && attrs_dev && data_dev, "scale_inverse: null argument");
    SV_REQUIRE(n > 0 && t > 0 && n_attr > 0 && x_dim >= n_attr, "scale_inverse: bad sizes");
    ScaleParams p{x_dev, mask_dev, attrs_dev, nullptr, data_dev, mask_out_dev, (long long)n * t * n_attr,
                  n, t, n_attr, x_dim, flags};
    return launch_scale(true, p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
