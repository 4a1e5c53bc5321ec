// Likelihood-parameter heads of the per-attribute decoders (decoders.py:69-102): Linear(H -> nout) with
// nout = 1 (bin logit, real/pos mean with a fixed std), 2 (mean + std), C-1 (categorical logits).
//
// With one to eight output columns these are not matrix-core work: through 256x32 / 32x256 MFMA tiles they ran at
// 3-9 % tile utilisation (2-35 TFLOP/s) and cost a tenth of the step.  They are streaming passes over the head
// input (a2 for real/pos, the hidden head layer for bin/cat): HBM-bound, 16-B loads, no MFMA.
//
//   forward : P[row][j] = b[j] + sum_k hin[i][k] W[j][k]                      one read of hin
//   backward: ONE pass over hin that produces all four results of the old NN GEMM + TN GEMM + two column-sum
//             launches:  dhin[i][k] = [hin[i][k] > 0] sum_j dP[row][j] W[j][k]      (input gradient, ReLU-gated)
//                        gW[j][k]  += sum_i dP[row][j] hin[i][k]                     (weight gradient)
//                        gb[j]     += sum_i dP[row][j]                               (bias gradient)
//                        gprev[k]  += sum_i dhin[i][k]                               (bias gradient of the layer that
//                                                                                     produced hin)
// A lane owns 4 consecutive k of a 256-wide column block; a wave walks rows.  row = i, or idx[i] through the
// attribute's observed-row list (mask-aware decoder: hin / dhin are compact, P stays in chunk-row order).
#include <algorithm>

#include "common.h"

namespace shivae {

typedef const __attribute__((address_space(1))) float* hgbl_f32_ptr;
typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) hf32x4* hgbl_f32x4_ptr;

__device__ __forceinline__ float4 hload4(const float* p) {
    const hf32x4 t = *(hgbl_f32x4_ptr)p;
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void hstore4(float* p, float4 v) {
    hf32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(__attribute__((address_space(1))) hf32x4*)p = t;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct HeadParams {
    const HeadGroup* groups;
    int H, ldin, ldp, rows, use_idx, use_cnt, rows_per_wave;
};

// grid: (row blocks, groups); block = 256 threads = 4 waves, each wave takes rows_per_wave consecutive rows
template <int NOUT, int CB>  // CB = column blocks of 256 per row (H <= 256*CB)
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadParams p) {
    const HeadGroup g = p.groups[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = p.use_cnt ? min(*g.cnt, p.rows) : p.rows;
    const int stride = gridDim.x * 4 * p.rows_per_wave;  // (the grid covers the EXPECTED live rows: launch_heads)
    if (wave * p.rows_per_wave >= n) return;
    float4 w[NOUT][CB];
    bool live[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int k = c * 256 + lane * 4;
        live[c] = k < p.H;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) w[j][c] = live[c] ? hload4(g.W + (size_t)j * p.H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float b[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) b[j] = g.b[j];
    for (int r0 = wave * p.rows_per_wave; r0 < n; r0 += stride)
    for (int i = r0, r1 = min(n, r0 + p.rows_per_wave); i < r1; ++i) {
        float acc[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc[j] = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (!live[c]) continue;
            const float4 h = hload4(g.hin + (size_t)i * p.ldin + c * 256 + lane * 4);
#pragma unroll
            for (int j = 0; j < NOUT; ++j)
                acc[j] += h.x * w[j][c].x + h.y * w[j][c].y + h.z * w[j][c].z + h.w * w[j][c].w;
        }
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc[j] = wave_sum(acc[j]);
        if (lane == 0) {
            const int row = p.use_idx ? g.idx[i] : i;
            float* o = g.P + (size_t)row * p.ldp;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) o[j] = acc[j] + b[j];
        }
    }
}

// Sum over the 16 lanes of a DPP row (no LDS round trip): quad xor 1, quad xor 2, half-row mirror, row mirror.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

// Forward, fast form (H a multiple of 64): 16 lanes share a row (lane c owns columns 4c + 64 i), so a wave has four
// rows in flight per step and eight per unrolled iteration; the dot products are reduced inside the DPP row.
template <int NOUT, int NI>  // NI = H / 64
__global__ __launch_bounds__(256) void head_fwd16_kernel(HeadParams p) {
    const HeadGroup g = p.groups[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = lane & 15, sub = lane >> 4;
    const int n = p.use_cnt ? min(*g.cnt, p.rows) : p.rows;
    const int stride = gridDim.x * 4 * p.rows_per_wave;
    if (wave * p.rows_per_wave >= n) return;
    float4 w[NOUT][NI];
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) w[j][i] = hload4(g.W + (size_t)j * p.H + 4 * c + 64 * i);
    float b[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) b[j] = g.b[j];
    constexpr int U = 2;  // row quads per iteration
    for (int r0 = wave * p.rows_per_wave; r0 < n; r0 += stride)
    for (int base = r0, r1 = min(n, r0 + p.rows_per_wave); base < r1; base += 4 * U) {
        float4 h[U][NI];
        int row[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + 4 * u + sub;
            ok[u] = i < r1;
            const int ii = ok[u] ? i : r1 - 1;
            row[u] = p.use_idx ? g.idx[ii] : ii;
#pragma unroll
            for (int k = 0; k < NI; ++k) h[u][k] = hload4(g.hin + (size_t)ii * p.ldin + 4 * c + 64 * k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc[NOUT];
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < NI; ++k)
                    a += h[u][k].x * w[j][k].x + h[u][k].y * w[j][k].y + h[u][k].z * w[j][k].z + h[u][k].w * w[j][k].w;
                acc[j] = row16_sum(a);
            }
            if (c == 0 && ok[u]) {
                float* o = g.P + (size_t)row[u] * p.ldp;
#pragma unroll
                for (int j = 0; j < NOUT; ++j) o[j] = acc[j] + b[j];
            }
        }
    }
}

template <int NOUT, int CB>
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadParams p) {
    const HeadGroup g = p.groups[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = p.use_cnt ? min(*g.cnt, p.rows) : p.rows;
    const int stride = gridDim.x * 4 * p.rows_per_wave;
    if (wave * p.rows_per_wave >= n) return;
    float4 w[NOUT][CB], gw[NOUT][CB], cs[CB];
    bool live[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int k = c * 256 + lane * 4;
        live[c] = k < p.H;
        cs[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            w[j][c] = live[c] ? hload4(g.W + (size_t)j * p.H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            gw[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float gb[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) gb[j] = 0.f;
    constexpr int U = 4;  // rows in flight per wave (memory-level parallelism: 4 KB of loads + 4 KB of stores)
    for (int r0 = wave * p.rows_per_wave; r0 < n; r0 += stride)
    for (int base = r0, r1 = min(n, r0 + p.rows_per_wave); base < r1; base += U) {
        float d[U][NOUT];
        float4 h[U][CB];
        bool ok[U];
        size_t off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u;
            ok[u] = i < r1;
            const int ii = ok[u] ? i : r1 - 1;
            const int row = p.use_idx ? g.idx[ii] : ii;
            const float* dp = g.P + (size_t)row * p.ldp;  // (wave-uniform address: one broadcast load per column)
#pragma unroll
            for (int j = 0; j < NOUT; ++j) d[u][j] = *(hgbl_f32_ptr)(dp + j);
            off[u] = (size_t)ii * p.ldin + lane * 4;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                h[u][c] = live[c] ? hload4(g.hin + off[u] + c * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;  // (wave-uniform)
#pragma unroll
            for (int j = 0; j < NOUT; ++j) gb[j] += d[u][j];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                if (!live[c]) continue;
                const float4 hv = h[u][c];
                float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const float dj = d[u][j];
                    dh.x += dj * w[j][c].x; dh.y += dj * w[j][c].y; dh.z += dj * w[j][c].z; dh.w += dj * w[j][c].w;
                    gw[j][c].x += dj * hv.x; gw[j][c].y += dj * hv.y; gw[j][c].z += dj * hv.z; gw[j][c].w += dj * hv.w;
                }
                dh.x = hv.x > 0.f ? dh.x : 0.f; dh.y = hv.y > 0.f ? dh.y : 0.f;
                dh.z = hv.z > 0.f ? dh.z : 0.f; dh.w = hv.w > 0.f ? dh.w : 0.f;
                hstore4(g.dhin + off[u] + c * 256, dh);
                cs[c].x += dh.x; cs[c].y += dh.y; cs[c].z += dh.z; cs[c].w += dh.w;
            }
        }
    }
    // one atomic per (wave, column): ~100 waves per attribute
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        if (!live[c]) continue;
        const int k = c * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            float* q = g.gW + (size_t)j * p.H + k;
            atomicAdd(q, gw[j][c].x); atomicAdd(q + 1, gw[j][c].y); atomicAdd(q + 2, gw[j][c].z); atomicAdd(q + 3, gw[j][c].w);
        }
        float* q = g.gprev + k;
        atomicAdd(q, cs[c].x); atomicAdd(q + 1, cs[c].y); atomicAdd(q + 2, cs[c].z); atomicAdd(q + 3, cs[c].w);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) atomicAdd(g.gb + j, gb[j]);
    }
}

bool head_kernels_fit(int H, int nout, int ldin, int ldp) {
    (void)ldp;
    return nout >= 1 && nout <= 8 && H % 4 == 0 && ldin % 4 == 0 && H <= 512;
}

template <int NOUT, int CB>
static void launch_one(bool bwd, const HeadParams& p, dim3 grid, hipStream_t s) {
    if (bwd) { hipLaunchKernelGGL((head_bwd_kernel<NOUT, CB>), grid, dim3(256), 0, s, p); return; }
    if constexpr (NOUT <= 4) {  // (weights of the fast forward form: NOUT * H/16 registers)
        if (p.H == 256) { hipLaunchKernelGGL((head_fwd16_kernel<NOUT, 4>), grid, dim3(256), 0, s, p); return; }
        if (p.H == 128) { hipLaunchKernelGGL((head_fwd16_kernel<NOUT, 2>), grid, dim3(256), 0, s, p); return; }
        if (p.H == 64) { hipLaunchKernelGGL((head_fwd16_kernel<NOUT, 1>), grid, dim3(256), 0, s, p); return; }
    }
    hipLaunchKernelGGL((head_fwd_kernel<NOUT, CB>), grid, dim3(256), 0, s, p);
}
template <int CB>
static void launch_nout(int nout, bool bwd, const HeadParams& p, dim3 grid, hipStream_t s) {
    switch (nout) {
        case 1: launch_one<1, CB>(bwd, p, grid, s); break;
        case 2: launch_one<2, CB>(bwd, p, grid, s); break;
        case 3: launch_one<3, CB>(bwd, p, grid, s); break;
        case 4: launch_one<4, CB>(bwd, p, grid, s); break;
        case 5: launch_one<5, CB>(bwd, p, grid, s); break;
        case 6: launch_one<6, CB>(bwd, p, grid, s); break;
        case 7: launch_one<7, CB>(bwd, p, grid, s); break;
        default: launch_one<8, CB>(bwd, p, grid, s); break;
    }
}

// rows: the dense row count (and the bound of every group's live count); use_cnt / use_idx: honour the groups'
// device-side counts / row lists (mask-aware decoder)
int launch_heads(bool bwd, const HeadGroup* groups_dev, int ngroups, int nout, int H, int ldin, int ldp, int rows,
                 bool use_cnt, bool use_idx, hipStream_t stream, int live_rows) {
    if (ngroups <= 0 || rows <= 0) return SHIVAE_OK;
    SV_REQUIRE(head_kernels_fit(H, nout, ldin, ldp), "heads: H=%d nout=%d ld=%d not supported by the streaming kernels", H, nout, ldin);
    SV_REQUIRE(ngroups <= 65535, "heads: too many groups");
    HeadParams p;
    p.groups = groups_dev; p.H = H; p.ldin = ldin; p.ldp = ldp; p.rows = rows;
    p.use_idx = use_idx ? 1 : 0; p.use_cnt = use_cnt ? 1 : 0;
    // ~16 waves per CU over the whole launch, at least 64 rows per wave (amortises the weight registers and, in
    // the backward, the atomics at the end of a wave)
    // With device-side row counts a group's live rows are its FIRST cnt rows, so the rows per wave are sized by the
    // expected count (live_rows; 0 = all rows): sized by `rows`, a sparsely observed group keeps one or two waves busy
    // and the launch is latency-bound (cfg5, 10 % observed: 1.6 waves per CU, 0.5 TB/s).
    const long long want_waves = 256ll * 16;
    const int lrows = (use_cnt && live_rows > 0) ? std::min(rows, live_rows) : rows;
    int rpw = (int)std::max<long long>(64, ((long long)lrows * ngroups + want_waves - 1) / want_waves);
    rpw = std::min(rpw, 1024);
    p.rows_per_wave = rpw;
    // (the grid covers lrows; the waves stride on through whatever a group has beyond that.  Sized by `rows`, a
    // launch at 10 % observed spends most of its time dispatching workgroups that find nothing to do)
    dim3 grid((unsigned)ceil_div(ceil_div(lrows, rpw), 4), (unsigned)ngroups);
    if (H <= 256) launch_nout<1>(nout, bwd, p, grid, stream);
    else launch_nout<2>(nout, bwd, p, grid, stream);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
