// Fused, time-looped recurrent cell of the Shi-VAE (north_star kernels 1, 2 and 4).
//
// Reference: shivae.py:102-143 (the per-timestep loop), encoders.py:70-98 (GMEncoder.forward),
// samplers.py:23-62 (Gumbel-softmax / Gaussian reparameterisation), shivae.py:132-137 (phi_z, nn.LSTM
// single step).  Everything that does NOT depend on h_{t-1} has been hoisted into batched GEMMs
// (gemm.hip); what is left here is the strictly serial chain
//     h_{t-1} -> a_s -> s logits -> (Gumbel) s_t -> e1 -> e2 -> (mu, std) -> z_t -> phi_z -> LSTM -> h_t
// and, in the backward kernel, the same chain in reverse time (full BPTT: the reference never
// detaches h/c between its 50-step chunks, shivae.py:102-137).
//
// One workgroup owns a tile of 16 sequences and loops over all T steps; nothing is exchanged
// between workgroups.  Formulation: activations are kept TRANSPOSED, [feature][16 sequences], so a
// layer is  out[N,16] = W[N,Kin] * in[Kin,16]  on v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A
// operand and the activations as the B operand.  Consequences:
//   * weights are pre-packed (pack kernels below) in exact A-operand order: one MFMA's 64 weights
//     are 256 contiguous bytes, a lane's float4 feeds four consecutive MFMAs;
//   * the D layout (row = 4*(lane>>4)+r, col = lane&15) of one layer's output tile nt is exactly the
//     B-operand float4 of k-block nt of the next layer, so activations move between layers as one
//     ds_write_b128 / ds_read_b128 per lane in a "quad-interleaved" LDS image
//         idx(k, j) = ((k>>2)*16 + j)*4 + (k&3)          (conflict-free for both),
//   * the LSTM cell state c and the running dh/dc of the backward pass never leave registers: the
//     wave that owns hidden units [16*ht, 16*ht+16) computes all four gates of those units.
// Feature dims are padded to multiples of 16 (H->Hp, Z->Zp, K->Kp) with zero weights/biases, which
// keeps every padded lane at exactly 0 through ReLU, the LSTM cell (c=0 -> h=0) and the gradients.
#include <algorithm>
#include <cstdlib>

#include "cell_math.h"
#include "common.h"
#include "noise.h"

namespace shivae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int qidx(int tile, int q, int j) { return ((tile * 4 + q) * 16 + j) * 4; }

// acc[a] += Wp(tile nt[a]) * in, summed over KB k-blocks of 16.
// Software-pipelined: the weights stream from L2 every timestep (3.1 MB at H=256: they do not fit one CU's LDS), so
// each k-block's float4 would otherwise be requested right in front of the four MFMAs that consume it and the L2
// round trip (~700 cycles against 128 cycles of MFMA) would sit on the serial chain.  DEPTH k-blocks of weights
// are kept in flight (16 registers whatever NA is); the tail re-requests the last block instead of branching.
template <int NA>
__device__ __forceinline__ void mma_acc(f32x4 (&acc)[NA], const float* __restrict__ W, int KB, const int (&nt)[NA],
                                        const float* in, int lane) {
    constexpr int DEPTH = NA >= 4 ? 1 : (NA >= 2 ? 2 : 4);
    const int j = lane & 15, q = lane >> 4;
    const float4* Wl = reinterpret_cast<const float4*>(W) + lane;
    float4 wq[DEPTH][NA];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int a = 0; a < NA; ++a) wq[d][a] = Wl[(size_t)(nt[a] * KB + min(d, KB - 1)) * 64];
    for (int kb0 = 0; kb0 < KB; kb0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kb = kb0 + d;
            if (kb < KB) {  // (wave-uniform)
                const float4 b = *reinterpret_cast<const float4*>(in + qidx(kb, q, j));
                float4 w[NA];
                const int kn = min(kb + DEPTH, KB - 1);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    w[a] = wq[d][a];
                    wq[d][a] = Wl[(size_t)(nt[a] * KB + kn) * 64];
                }
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].x, b.x, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].y, b.y, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].z, b.z, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].w, b.w, acc[a], 0, 0, 0);
                }
            }
        }
    }
}

struct Src {
    const float* W;
    const float* in;
    int KB;
};

// out tiles 0..ntiles-1 are dealt round-robin to the waves; epi(nt, acc) gets the finished tile
template <int NSRC, class Epi>
__device__ __forceinline__ void layer(int ntiles, const Src (&src)[NSRC], int wid, int NW, int lane, Epi epi) {
    int t0 = wid;
    for (; t0 + 3 * NW < ntiles; t0 += 4 * NW) {
        const int nt[4] = {t0, t0 + NW, t0 + 2 * NW, t0 + 3 * NW};
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < NSRC; ++s) mma_acc<4>(acc, src[s].W, src[s].KB, nt, src[s].in, lane);
#pragma unroll
        for (int a = 0; a < 4; ++a) epi(nt[a], acc[a]);
    }
    for (; t0 < ntiles; t0 += NW) {
        const int nt[1] = {t0};
        f32x4 acc[1] = {{0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < NSRC; ++s) mma_acc<1>(acc, src[s].W, src[s].KB, nt, src[s].in, lane);
        epi(t0, acc[0]);
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
// Records go to HBM and are not read again by this kernel; the packed weights are re-read every timestep.  Streaming
// (non-temporal) stores / loads for the records keep them from pushing the 3 MB of weights out of the XCD's 4 MiB L2.
__device__ __forceinline__ void stg4(float* p, f32x4 v) {
    __builtin_nontemporal_store(v, (__attribute__((address_space(1))) f32x4*)p);
}
__device__ __forceinline__ f32x4 ldg4(const float* p) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) f32x4*)p);
}
// Transcendentals of the recurrent chain.  They sit on the serial critical path of every timestep (the cell alone is
// three sigmoids and two tanh per hidden unit, the Gumbel-softmax two logs, two exps and two divisions per class), and
// the libm forms cost 12-40 DEPENDENT vector instructions each -- for a small model, whose step is one wave's latency
// chain, about a third of the step.  The hardware-rate forms of cell_math.h (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp
// each) cost 2-6.  tanh(x) = sign(x) (1 - e) / (1 + e), e = exp(-2|x|): absolute error <= 2^-23 everywhere (relative
// error grows only where |x| < 1e-3, i.e. where the value itself is below 1e-3).  -DSV_SCAN_LIBM restores libm.
#ifdef SV_SCAN_LIBM
__device__ __forceinline__ float sc_exp(float v) { return expf(v); }
__device__ __forceinline__ float sc_rcp(float v) { return 1.f / v; }
__device__ __forceinline__ float sc_sigmoid(float v) { return sv_sigmoid(v); }
__device__ __forceinline__ float sc_tanh(float v) { return tanhf(v); }
__device__ __forceinline__ float sc_softplus(float v) { return sv_softplus(v); }
__device__ __forceinline__ float sc_softplus_grad_from_out(float sp) { return sv_softplus_grad_from_out(sp); }
__device__ __forceinline__ float sc_gumbel(float u) { return sv_gumbel(u); }
#else
__device__ __forceinline__ float sc_exp(float v) { return sv_fexp(v); }
__device__ __forceinline__ float sc_rcp(float v) { return sv_frcp(v); }
__device__ __forceinline__ float sc_sigmoid(float v) { return sv_frcp(1.f + sv_fexp(-v)); }
__device__ __forceinline__ float sc_tanh(float v) {
    const float e = sv_fexp(-2.f * fabsf(v));  // in (0, 1]
    return copysignf((1.f - e) * sv_frcp(1.f + e), v);
}
__device__ __forceinline__ float sc_softplus(float v) { return sv_softplus_fast(v); }
__device__ __forceinline__ float sc_softplus_grad_from_out(float sp) { return sv_softplus_grad_from_out_fast(sp); }
__device__ __forceinline__ float sc_gumbel(float u) { return -sv_flog(-sv_flog(u + 1e-20f) + 1e-20f); }
#endif
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ __forceinline__ f32x4 gate4(f32x4 v, f32x4 a) { return f32x4{a[0] > 0.f ? v[0] : 0.f, a[1] > 0.f ? v[1] : 0.f, a[2] > 0.f ? v[2] : 0.f, a[3] > 0.f ? v[3] : 0.f}; }

// ---- noise: the caller's tensors u[T][B][K] / eps[T][B][Z], or -- u == eps == nullptr -- the counter-based streams
// of noise.h under ScanArgs::seed.  Same numbers in the forward scan, the reverse scan and noise_fill_kernel.
__device__ __forceinline__ f32x4 sv_eps4(const ScanArgs& a, size_t row, int g, bool live) {  // eps[row][4g .. 4g+3]
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    if (!live) return e;
    if (a.eps) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.Z) e[r] = a.eps[row * a.Z + 4 * g + r];
    } else {
        const SvF4 t = sv_noise_eps4(a.seed, row, g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.Z) e[r] = t.v[r];
    }
    return e;
}
__device__ __forceinline__ float sv_u1(const ScanArgs& a, size_t row, int k) {  // u[row][k]
    if (a.u) return a.u[row * a.K + k];
    const SvF4 t = sv_noise_u4(a.seed, row, k >> 2);
    const int e = k & 3;
    return e == 0 ? t.v[0] : e == 1 ? t.v[1] : e == 2 ? t.v[2] : t.v[3];
}

// ---------------------------------------------------------------------------------- forward
// Per timestep the serial chain is  a_s -> logits -> s -> e1 -> e2 -> (mu,std,z) -> phi_z -> gates.
// Everything that needs only h_{t-1} (the h halves of a_s and e1, Whh h_{t-1}: half of the MFMA work)
// is issued in the first two phases, where it overlaps the one-wave logits/Gumbel-softmax stage; the
// operands that do not depend on the recurrence (xs_pre, ex_pre, eps, u) are requested at the top of
// the step so their HBM latency is off the chain.  A wave owns hidden tiles ht = wid + s*NW, s < MAXS.
// WLDS: the packed weights of a small model fit in LDS next to the activations; they are copied there once and
// every step's operand fetch is an LDS read instead of an L2 round trip on the serial chain.
template <int MAXS, int NTH, bool WLDS>
__global__ __launch_bounds__(NTH) void scan_fwd_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, NW = blockDim.x >> 6;
    const int j = lane & 15;
    const int Hp = a.Hp, Zp = a.Zp, Kp = a.Kp, Ep = a.Ep, B = a.B;
    const int HT = Hp >> 4, ZT = Zp >> 4, KT = Kp >> 4;
    const int bj = blockIdx.x * 16 + j;
    const bool live = bj < B;
    float* hb0 = smem;
    float* hb1 = hb0 + Hp * 16;
    float* bufA = hb1 + Hp * 16;
    float* bufB = bufA + Hp * 16;
    float* sbuf = bufB + Hp * 16;
    float* lgbuf = sbuf + Kp * 16;
    float* zbuf = lgbuf + Kp * 16;
    float* wl = zbuf + Zp * 16;
    if (WLDS)
        for (int i = tid; i < a.wlds_floats / 4; i += blockDim.x)
            reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(a.wpack)[i];
    const float* Wb = WLDS ? wl : a.wpack;
    const f32x4 zero4 = {0, 0, 0, 0};

    for (int i = tid; i < Hp * 16; i += blockDim.x) hb0[i] = 0.f;  // shivae.py:93 h0 = 0
    f32x4 creg[MAXS];
    int own[MAXS];      // owned hidden tile (clamped), valid if ownok
    bool ownok[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        creg[s] = zero4;  // shivae.py:94 c0 = 0
        ownok[s] = wid + s * NW < HT;
        own[s] = ownok[s] ? wid + s * NW : 0;
    }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < a.T; ++t) {
        size_t row = (size_t)t * B + (live ? bj : 0);
        int ln = lane;
        // (opaque to the optimiser: otherwise the per-lane addresses of every record array and of every weight image
        // are kept in registers across the whole time loop, two each, and the kernel spills)
        asm volatile("" : "+v"(row), "+v"(ln));
        const int j = ln & 15, q = ln >> 4;
        const float* h = cur ? hb1 : hb0;
        float* hn = cur ? hb0 : hb1;

        // operands independent of the recurrence: request now, use later
        f32x4 xs_pf[MAXS], ex_pf[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const bool ld = live && ownok[s];
            xs_pf[s] = ld ? ldg4(a.xs_pre + row * Hp + own[s] * 16 + 4 * q) : zero4;
            ex_pf[s] = ld ? ldg4(a.ex_pre + row * Hp + own[s] * 16 + 4 * q) : zero4;
        }
        float u_pf[4] = {0.5f, 0.5f, 0.5f, 0.5f};
        const f32x4 eps_pf = sv_eps4(a, row, wid * 4 + q, wid < ZT && live);
        if (wid == 0 && KT == 1 && live) {
            if (a.u) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q + r < a.K) u_pf[r] = a.u[row * a.K + 4 * q + r];
            } else {
                const SvF4 t = sv_noise_u4(a.seed, row, q);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q + r < a.K) u_pf[r] = t.v[r];
            }
        }

        // P0: a_s = relu(xs_pre + Wxs0[:, X:] h)   encoders.py:83 (x half hoisted);  e1 partial = We0[:, H+K:] h
        f32x4 e1acc[MAXS];
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) { acc[s] = zero4; e1acc[s] = zero4; }
            mma_acc<MAXS>(acc, Wb + a.w.xsh, HT, own, h, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const f32x4 v = relu4(acc[s] + xs_pf[s]);
                    st4(bufA + qidx(own[s], q, j), v);
                    if (live) stg4(a.as_ + row * Hp + own[s] * 16 + 4 * q, v);
                }
            mma_acc<MAXS>(e1acc, Wb + a.w.e0h, HT, own, h, ln);
        }
        __syncthreads();
        // P1: s logits = LeakyReLU(Wxs2 a_s + b) (encoders.py:56-57); s_probs = softmax(logits),
        //     s = softmax((logits + gumbel(u)) / temp) (encoders.py:87-88, samplers.py:23-30)
        if (KT == 1) {
            // K <= 16: the logits tile of wave 0 has k = 4q+r in lane (q, j): softmax over k is an in-lane
            // reduction over r plus two xor-shuffles over q -- no LDS round trip, no extra barrier
            if (wid == 0) {
                const int nt0[1] = {0};
                f32x4 lg[1] = {zero4};
                mma_acc<1>(lg, Wb + a.w.lg, HT, nt0, bufA, ln);
                const f32x4 bl = ld4(Wb + a.w.b_lg + 4 * q);
                float l[4], y[4], m1 = -INFINITY, m2 = -INFINITY;
                const float inv_temp = sc_rcp(a.temp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kv = 4 * q + r < a.K;
                    l[r] = sv_leaky(lg[0][r] + bl[r]);
                    y[r] = (l[r] + sc_gumbel(u_pf[r])) * inv_temp;
                    if (kv) { m1 = fmaxf(m1, l[r]); m2 = fmaxf(m2, y[r]); }
                }
                m1 = fmaxf(m1, __shfl_xor(m1, 16)); m1 = fmaxf(m1, __shfl_xor(m1, 32));
                m2 = fmaxf(m2, __shfl_xor(m2, 16)); m2 = fmaxf(m2, __shfl_xor(m2, 32));
                float e1v[4], e2v[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kv = 4 * q + r < a.K;
                    e1v[r] = kv ? sc_exp(l[r] - m1) : 0.f;
                    e2v[r] = kv ? sc_exp(y[r] - m2) : 0.f;
                    s1 += e1v[r]; s2 += e2v[r];
                }
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                f32x4 pv, sv, lv4;
                const float rs1 = sc_rcp(s1), rs2 = sc_rcp(s2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = e1v[r] * rs1; sv[r] = e2v[r] * rs2;
                    lv4[r] = 4 * q + r < a.K ? l[r] : 0.f;
                }
                st4(sbuf + qidx(0, q, j), sv);
                if (live) {
                    stg4(a.logits + row * Kp + 4 * q, lv4);
                    stg4(a.probs + row * Kp + 4 * q, pv);
                    stg4(a.emb + row * Ep + Hp + 4 * q, sv);
                }
            }
        } else {
            {
                const Src src[1] = {{Wb + a.w.lg, bufA, HT}};
                layer<1>(KT, src, wid, NW, ln, [&](int nt, f32x4 v) {
                    v += ld4(Wb + a.w.b_lg + nt * 16 + 4 * q);
                    st4(lgbuf + qidx(nt, q, j), f32x4{sv_leaky(v[0]), sv_leaky(v[1]), sv_leaky(v[2]), sv_leaky(v[3])});
                });
            }
            __syncthreads();
            if (tid < 16) {
                const int jj = tid, bb = blockIdx.x * 16 + jj;
                const bool lv = bb < B;
                const size_t r = (size_t)t * B + (lv ? bb : 0);
                float m1 = -INFINITY, m2 = -INFINITY;
                for (int k = 0; k < a.K; ++k) {
                    const float l = lgbuf[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)];
                    const float u = lv ? sv_u1(a, r, k) : 0.5f;
                    m1 = fmaxf(m1, l);
                    m2 = fmaxf(m2, (l + sv_gumbel(u)) / a.temp);
                }
                float s1 = 0.f, s2 = 0.f;
                for (int k = 0; k < a.K; ++k) {
                    const float l = lgbuf[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)];
                    const float u = lv ? sv_u1(a, r, k) : 0.5f;
                    s1 += expf(l - m1);
                    s2 += expf((l + sv_gumbel(u)) / a.temp - m2);
                }
                for (int k = 0; k < Kp; ++k) {
                    float sv = 0.f;
                    if (k < a.K) {
                        const float l = lgbuf[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)];
                        const float u = lv ? sv_u1(a, r, k) : 0.5f;
                        sv = expf((l + sv_gumbel(u)) / a.temp - m2) / s2;
                        if (lv) {
                            a.logits[r * Kp + k] = l;
                            a.probs[r * Kp + k] = expf(l - m1) / s1;
                            a.emb[r * Ep + Hp + k] = sv;
                        }
                    }
                    sbuf[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)] = sv;
                }
            }
        }
        //     meanwhile, off the chain: LSTM gate partials Whh h_{t-1} (gate order i,f,g,o; shivae.py:136)
        f32x4 g4[MAXS][4];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int nt[4] = {own[s], HT + own[s], 2 * HT + own[s], 3 * HT + own[s]};
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) g4[s][gi] = zero4;
            if (ownok[s]) mma_acc<4>(g4[s], Wb + a.w.hh, HT, nt, h, ln);
        }
        __syncthreads();
        // P2: e1 = relu(ex_pre + We0[:, H:H+K] s + We0[:, H+K:] h)    encoders.py:91 (phi_x part hoisted)
        {
            mma_acc<MAXS>(e1acc, Wb + a.w.e0s, KT, own, sbuf, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const f32x4 v = relu4(e1acc[s] + ex_pf[s]);
                    st4(bufB + qidx(own[s], q, j), v);
                    if (live) stg4(a.e1 + row * Hp + own[s] * 16 + 4 * q, v);
                }
        }
        __syncthreads();
        // S5: e2 = relu(We2 e1 + b)                         encoders.py:63-64
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.e2, HT, own, bufB, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const int n = own[s] * 16 + 4 * q;
                    const f32x4 v = relu4(acc[s] + ld4(Wb + a.w.b_e2 + n));
                    st4(bufA + qidx(own[s], q, j), v);
                    if (live) stg4(a.e2 + row * Hp + n, v);
                }
        }
        __syncthreads();
        // S6: mu, std = softplus(.), z = eps*std + mu       encoders.py:93-96, samplers.py:57-62
        for (int zt = wid; zt < ZT; zt += NW) {
            const int nt[1] = {zt};
            f32x4 am[1] = {zero4}, as[1] = {zero4};
            mma_acc<1>(am, Wb + a.w.mu, HT, nt, bufA, ln);
            mma_acc<1>(as, Wb + a.w.sd, HT, nt, bufA, ln);
            const int n = zt * 16 + 4 * q;
            const f32x4 mu = am[0] + ld4(Wb + a.w.b_mu + n);
            const f32x4 sp = as[0] + ld4(Wb + a.w.b_sd + n);
            f32x4 sd, z, ev = eps_pf;
            if (zt != wid) ev = sv_eps4(a, row, zt * 4 + q, live);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sd[r] = sc_softplus(sp[r]);
                z[r] = ev[r] * sd[r] + mu[r];
            }
            st4(zbuf + qidx(zt, q, j), z);
            if (live) {
                stg4(a.mu + row * Zp + n, mu);
                stg4(a.sd + row * Zp + n, sd);
                stg4(a.z + row * Zp + n, z);
            }
        }
        __syncthreads();
        // S7: phi_z = relu(Wz z + b)                        shivae.py:132
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.pz, ZT, own, zbuf, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const int n = own[s] * 16 + 4 * q;
                    const f32x4 v = relu4(acc[s] + ld4(Wb + a.w.b_pz + n));
                    st4(bufB + qidx(own[s], q, j), v);
                    if (live) stg4(a.emb + row * Ep + n, v);
                }
        }
        __syncthreads();
        // S8: LSTM cell, c in registers                      shivae.py:136
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            if (ownok[s]) {
                const int ht = own[s];
                const int nt[4] = {ht, HT + ht, 2 * HT + ht, 3 * HT + ht};
                mma_acc<4>(g4[s], Wb + a.w.ih, HT, nt, bufB, ln);
                const int n = ht * 16 + 4 * q;
#pragma unroll
                for (int gidx = 0; gidx < 4; ++gidx) g4[s][gidx] += ld4(Wb + a.w.b_g + gidx * Hp + n);
                f32x4 gi, gf, gg, go, cn, hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gi[r] = sc_sigmoid(g4[s][0][r]);
                    gf[r] = sc_sigmoid(g4[s][1][r]);
                    gg[r] = sc_tanh(g4[s][2][r]);
                    go[r] = sc_sigmoid(g4[s][3][r]);
                    cn[r] = gf[r] * creg[s][r] + gi[r] * gg[r];
                    hv[r] = go[r] * sc_tanh(cn[r]);
                }
                creg[s] = cn;
                st4(hn + qidx(ht, q, j), hv);
                if (live) {
                    float* gp = a.gates + row * 4 * Hp + n;
                    stg4(gp, gi); stg4(gp + Hp, gf); stg4(gp + 2 * Hp, gg); stg4(gp + 3 * Hp, go);
                    stg4(a.cnew + row * Hp + n, cn);
                    if (t + 1 < a.T) stg4(a.emb + (row + B) * Ep + Hp + Kp + n, hv);  // h_prev of step t+1
                    else if (a.h_last) stg4(a.h_last + (size_t)bj * Hp + n, hv);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------------------------- generation
// ShiVAE.sample_from_prior (shivae.py:156-214): per step  s_t = gumbel_softmax(log(1/K), temp);
// prior_t = relu(W[h, s] + b); z_t = eps*softplus(Ws prior_t + b) + (Wm prior_t + b); phi_z = relu(Wz z_t + b);
// LSTM step.  Same tile / operand scheme as the training scan; the decoder runs afterwards over all rows
// from the emb records [phi_z | s | h_{t-1}].  Records written: emb, z, mu (prior mean), sd (prior std),
// probs (the uniform prior 1/K, the reference's s_prior_t_probs).
template <int MAXS, int NTH>
__global__ __launch_bounds__(NTH) void scan_gen_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, NW = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int Hp = a.Hp, Zp = a.Zp, Kp = a.Kp, Ep = a.Ep, B = a.B;
    const int HT = Hp >> 4, ZT = Zp >> 4, KT = Kp >> 4;
    const int bj = blockIdx.x * 16 + j;
    const bool live = bj < B;
    float* hb0 = smem;
    float* hb1 = hb0 + Hp * 16;
    float* bufA = hb1 + Hp * 16;
    float* bufB = bufA + Hp * 16;
    float* sbuf = bufB + Hp * 16;
    float* zbuf = sbuf + 2 * Kp * 16;
    const float* Wb = a.wpack;
    const f32x4 zero4 = {0, 0, 0, 0};
    const float lq = logf(1.f / (float)a.K);  // torch.log(s_prior_t_probs), shivae.py:179-180

    for (int i = tid; i < Hp * 16; i += blockDim.x) hb0[i] = 0.f;  // shivae.py:170-171 h = c = 0
    f32x4 creg[MAXS];
    int own[MAXS];
    bool ownok[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        creg[s] = zero4;
        ownok[s] = wid + s * NW < HT;
        own[s] = ownok[s] ? wid + s * NW : 0;
    }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < a.T; ++t) {
        const size_t row = (size_t)t * B + (live ? bj : 0);
        const float* h = cur ? hb1 : hb0;
        float* hn = cur ? hb0 : hb1;
        // G0: s_t (one thread per sequence; K is small) -- samplers.py:23-30 on constant logits
        if (tid < 16) {
            const int jj = tid, bb = blockIdx.x * 16 + jj;
            const bool lv = bb < B;
            const size_t r = (size_t)t * B + (lv ? bb : 0);
            float m2 = -INFINITY;
            for (int k = 0; k < a.K; ++k) m2 = fmaxf(m2, (lq + sv_gumbel(lv ? sv_u1(a, r, k) : 0.5f)) / a.temp);
            float s2 = 0.f;
            for (int k = 0; k < a.K; ++k) s2 += expf((lq + sv_gumbel(lv ? sv_u1(a, r, k) : 0.5f)) / a.temp - m2);
            for (int k = 0; k < Kp; ++k) {
                float sv = 0.f;
                if (k < a.K) {
                    sv = expf((lq + sv_gumbel(lv ? sv_u1(a, r, k) : 0.5f)) / a.temp - m2) / s2;
                    if (lv) {
                        a.emb[r * Ep + Hp + k] = sv;
                        a.probs[r * Kp + k] = 1.f / (float)a.K;
                    }
                }
                sbuf[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)] = sv;
            }
        }
        //     meanwhile: the h halves of prior.0 and of the LSTM gates
        f32x4 pacc[MAXS], g4[MAXS][4];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) pacc[s] = zero4;
        mma_acc<MAXS>(pacc, Wb + a.w.p0h, HT, own, h, lane);
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int nt[4] = {own[s], HT + own[s], 2 * HT + own[s], 3 * HT + own[s]};
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) g4[s][gi] = zero4;
            if (ownok[s]) mma_acc<4>(g4[s], Wb + a.w.hh, HT, nt, h, lane);
        }
        __syncthreads();
        // G1: prior_t = relu(prior.0 [h, s] + b)             shivae.py:184
        mma_acc<MAXS>(pacc, Wb + a.w.p0s, KT, own, sbuf, lane);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (ownok[s]) {
                const int n = own[s] * 16 + 4 * q;
                st4(bufA + qidx(own[s], q, j), relu4(pacc[s] + ld4(Wb + a.w.b_p0 + n)));
            }
        __syncthreads();
        // G2: prior mean / std, z = eps*std + mean           shivae.py:185-188
        for (int zt = wid; zt < ZT; zt += NW) {
            const int nt[1] = {zt};
            f32x4 am[1] = {zero4}, as[1] = {zero4};
            mma_acc<1>(am, Wb + a.w.pm, HT, nt, bufA, lane);
            mma_acc<1>(as, Wb + a.w.ps, HT, nt, bufA, lane);
            const int n = zt * 16 + 4 * q;
            const f32x4 mu = am[0] + ld4(Wb + a.w.b_pm + n);
            const f32x4 sp = as[0] + ld4(Wb + a.w.b_ps + n);
            f32x4 sd, z;
            const f32x4 ev = sv_eps4(a, row, zt * 4 + q, live);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sd[r] = sc_softplus(sp[r]);
                z[r] = ev[r] * sd[r] + mu[r];
            }
            st4(zbuf + qidx(zt, q, j), z);
            if (live) {
                stg4(a.mu + row * Zp + n, mu);
                stg4(a.sd + row * Zp + n, sd);
                stg4(a.z + row * Zp + n, z);
            }
        }
        __syncthreads();
        // G3: phi_z = relu(Wz z + b)                         shivae.py:194
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.pz, ZT, own, zbuf, lane);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const int n = own[s] * 16 + 4 * q;
                    const f32x4 v = relu4(acc[s] + ld4(Wb + a.w.b_pz + n));
                    st4(bufB + qidx(own[s], q, j), v);
                    if (live) stg4(a.emb + row * Ep + n, v);
                }
        }
        __syncthreads();
        // G4: LSTM cell                                      shivae.py:198-199
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            if (ownok[s]) {
                const int ht = own[s];
                const int nt[4] = {ht, HT + ht, 2 * HT + ht, 3 * HT + ht};
                mma_acc<4>(g4[s], Wb + a.w.ih, HT, nt, bufB, lane);
                const int n = ht * 16 + 4 * q;
#pragma unroll
                for (int gidx = 0; gidx < 4; ++gidx) g4[s][gidx] += ld4(Wb + a.w.b_g + gidx * Hp + n);
                f32x4 cn, hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gi = sc_sigmoid(g4[s][0][r]), gf = sc_sigmoid(g4[s][1][r]);
                    const float gg = sc_tanh(g4[s][2][r]), go = sc_sigmoid(g4[s][3][r]);
                    cn[r] = gf * creg[s][r] + gi * gg;
                    hv[r] = go * sc_tanh(cn[r]);
                }
                creg[s] = cn;
                st4(hn + qidx(ht, q, j), hv);
                if (live) {
                    if (t == 0) stg4(a.emb + row * Ep + Hp + Kp + n, zero4);  // h_{-1} = 0 record of the first step
                    if (t + 1 < a.T) stg4(a.emb + (row + B) * Ep + Hp + Kp + n, hv);  // h_prev of step t+1
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---------------------------------------------------------------------------------- backward
template <int MAXS, int NTH, bool WLDS>
__global__ __launch_bounds__(NTH) void scan_bwd_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, NW = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int Hp = a.Hp, Zp = a.Zp, Kp = a.Kp, Ep = a.Ep, B = a.B;
    const int HT = Hp >> 4, ZT = Zp >> 4, KT = Kp >> 4;
    const int bj = blockIdx.x * 16 + j;
    const bool live = bj < B;
    float* dhb = smem;
    float* dgb = dhb + Hp * 16;
    float* bufA = dgb + 4 * Hp * 16;
    float* bufB = bufA + Hp * 16;
    float* dmub = bufB + Hp * 16;
    float* dsdb = dmub + Zp * 16;
    float* dsb = dsdb + Zp * 16;
    float* dlgb = dsb + Kp * 16;
    float* wl = dlgb + Kp * 16;
    if (WLDS)
        for (int i = tid; i < a.wlds_floats / 4; i += blockDim.x)
            reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(a.wpack)[i];
    const float* Wb = WLDS ? wl : a.wpack;
    const f32x4 zero4 = {0, 0, 0, 0};

    for (int i = tid; i < Hp * 16; i += blockDim.x) dhb[i] = 0.f;
    f32x4 dcreg[MAXS], dhacc[MAXS];
    int own[MAXS];      // owned hidden tile (clamped), valid if ownok
    bool ownok[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        dcreg[s] = zero4;
        ownok[s] = wid + s * NW < HT;
        own[s] = ownok[s] ? wid + s * NW : 0;
    }
    __syncthreads();

    // Every record this chain reads back (gates, cell states, the ReLU gates of the four hidden layers, the
    // gradients arriving from the decoder / prior / KL terms) is independent of the recurrence: the cell records of
    // step t-1 are requested while step t runs, everything else at the top of its step, so that no HBM round trip
    // sits between two phases of the serial chain.
    f32x4 c_gi[MAXS], c_gf[MAXS], c_gg[MAXS], c_go[MAXS], c_cn[MAXS], c_cp[MAXS];
    auto load_cell = [&](int t) {
        size_t row = (size_t)t * B + (live ? bj : 0);
        asm volatile("" : "+v"(row));
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            c_gi[s] = c_gf[s] = c_gg[s] = c_go[s] = c_cn[s] = c_cp[s] = zero4;
            if (live && ownok[s]) {
                const int n = own[s] * 16 + 4 * q;
                const float* gp = a.gates + row * 4 * Hp + n;
                c_gi[s] = ldg4(gp); c_gf[s] = ldg4(gp + Hp); c_gg[s] = ldg4(gp + 2 * Hp); c_go[s] = ldg4(gp + 3 * Hp);
                c_cn[s] = ldg4(a.cnew + row * Hp + n);
                if (t > 0) c_cp[s] = ldg4(a.cnew + (row - B) * Hp + n);
            }
        }
    };
    load_cell(a.T - 1);

    for (int t = a.T - 1; t >= 0; --t) {
        size_t row = (size_t)t * B + (live ? bj : 0);
        // (opaque to the optimiser: otherwise every record array gets a strength-reduced per-lane pointer of its own,
        // two registers each for ~25 arrays, and the kernel spills)
        asm volatile("" : "+v"(row));
        int ln = lane;  // (same for the per-lane weight addresses of the ten mma_acc call sites)
        asm volatile("" : "+v"(ln));
        const int j = ln & 15, q = ln >> 4;  // (shadow the kernel-scope copies: LDS / record offsets are recomputed, not kept)
        f32x4 g_pz[MAXS], d_pz[MAXS], g_e2[MAXS], g_e1[MAXS], g_as[MAXS], d_hx[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const bool ld = live && ownok[s];
            const int n = own[s] * 16 + 4 * q;
            g_pz[s] = ld ? ldg4(a.emb + row * Ep + n) : zero4;
            d_pz[s] = ld ? ldg4(a.d_emb + row * Ep + n) : zero4;
            g_e2[s] = ld ? ldg4(a.e2 + row * Hp + n) : zero4;
            g_e1[s] = ld ? ldg4(a.e1 + row * Hp + n) : zero4;
        }  // (g_as, d_hx: requested after B2, in the registers g_pz / d_pz leave)
        f32x4 z_dmu = zero4, z_dsd = zero4, z_sd = zero4, z_eps = zero4;  // z tile wid (further tiles load in place)
        if (wid < ZT && live) {
            const int n = wid * 16 + 4 * q;
            z_dmu = ldg4(a.dmu_ext + row * Zp + n);
            z_dsd = ldg4(a.dsd_ext + row * Zp + n);
            z_sd = ldg4(a.sd + row * Zp + n);
            z_eps = sv_eps4(a, row, wid * 4 + q, true);
        }

        // B1: LSTM cell backward on the owned hidden units (elementwise, registers)
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            dhacc[s] = zero4;
            if (ownok[s]) {
                const int ht = own[s], n = ht * 16 + 4 * q;
                const f32x4 dh = ld4(dhb + qidx(ht, q, j));
                const f32x4 gi = c_gi[s], gf = c_gf[s], gg = c_gg[s], go = c_go[s], cn = c_cn[s], cp = c_cp[s];
                f32x4 di, df, dg, dO, dc = dcreg[s];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tc = sc_tanh(cn[r]);
                    const float dct = dc[r] + dh[r] * go[r] * (1.f - tc * tc);
                    di[r] = dct * gg[r] * gi[r] * (1.f - gi[r]);
                    df[r] = dct * cp[r] * gf[r] * (1.f - gf[r]);
                    dg[r] = dct * gi[r] * (1.f - gg[r] * gg[r]);
                    dO[r] = dh[r] * tc * go[r] * (1.f - go[r]);
                    dc[r] = dct * gf[r];
                }
                dcreg[s] = dc;
                st4(dgb + qidx(ht, q, j), di);
                st4(dgb + qidx(HT + ht, q, j), df);
                st4(dgb + qidx(2 * HT + ht, q, j), dg);
                st4(dgb + qidx(3 * HT + ht, q, j), dO);
                if (live) {
                    float* gp = a.d_gates + row * 4 * Hp + n;
                    stg4(gp, di); stg4(gp + Hp, df); stg4(gp + 2 * Hp, dg); stg4(gp + 3 * Hp, dO);
                }
            }
        }
        if (t > 0) load_cell(t - 1);  // in flight for the rest of this step
        __syncthreads();
        // B2: dphi_z = Wih^T dgates + (decoder's dphi_z); through ReLU.   dh_prev += Whh^T dgates
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.t_ih, 4 * HT, own, dgb, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const int n = own[s] * 16 + 4 * q;
                    const f32x4 v = gate4(acc[s] + d_pz[s], g_pz[s]);
                    st4(bufB + qidx(own[s], q, j), v);
                    if (live) stg4(a.d_phiz_pre + row * Hp + n, v);
                }
            mma_acc<MAXS>(dhacc, Wb + a.w.t_hh, 4 * HT, own, dgb, ln);
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const bool ld = live && ownok[s];
            const int n = own[s] * 16 + 4 * q;
            g_as[s] = ld ? ldg4(a.as_ + row * Hp + n) : zero4;
            d_hx[s] = ld ? ldg4(a.d_emb + row * Ep + Hp + Kp + n) : zero4;
        }
        __syncthreads();
        // B3: dz = Wz^T dphi_z_pre; dmu = dz + dKL/dmu; dstd_pre = (dz*eps + dKL/dstd) * softplus'
        for (int zt = wid; zt < ZT; zt += NW) {
            const int nt[1] = {zt};
            f32x4 acc[1] = {zero4};
            mma_acc<1>(acc, Wb + a.w.t_pz, HT, nt, bufB, ln);
            const int n = zt * 16 + 4 * q;
            f32x4 xm = z_dmu, xs = z_dsd, sd = z_sd, ep = z_eps;
            if (zt != wid) {
                xm = xs = sd = ep = zero4;
                if (live) {
                    xm = ldg4(a.dmu_ext + row * Zp + n); xs = ldg4(a.dsd_ext + row * Zp + n); sd = ldg4(a.sd + row * Zp + n);
                    ep = sv_eps4(a, row, zt * 4 + q, true);
                }
            }
            f32x4 dmu = acc[0] + xm, dsd = zero4;
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dsd[r] = (acc[0][r] * ep[r] + xs[r]) * sc_softplus_grad_from_out(sd[r]);
            } else dmu = acc[0];
            st4(dmub + qidx(zt, q, j), dmu);
            st4(dsdb + qidx(zt, q, j), dsd);
            if (live) {
                stg4(a.d_mu + row * Zp + n, dmu);
                stg4(a.d_sd_pre + row * Zp + n, dsd);
            }
        }
        f32x4 k_s = zero4, k_ds = zero4, k_kl = zero4, k_lg = zero4;  // wave 0, K <= 16: the s row and what reaches it
        if (wid == 0 && KT == 1 && live) {  // (requested two phases ahead of B6, in the registers the z operands leave)
            k_s = ldg4(a.emb + row * Ep + Hp + 4 * q);
            k_ds = ldg4(a.d_emb + row * Ep + Hp + 4 * q);
            k_kl = ldg4(a.dlg_kl + row * Kp + 4 * q);
            k_lg = ldg4(a.logits + row * Kp + 4 * q);
        }
        __syncthreads();
        // B4: de2 = Wm^T dmu + Ws^T dstd_pre, through ReLU
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.t_mu, ZT, own, dmub, ln);
            mma_acc<MAXS>(acc, Wb + a.w.t_sd, ZT, own, dsdb, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const f32x4 v = gate4(acc[s], g_e2[s]);
                    st4(bufA + qidx(own[s], q, j), v);
                    if (live) stg4(a.d_e2_pre + row * Hp + own[s] * 16 + 4 * q, v);
                }
        }
        __syncthreads();
        // B5: de1 = We2^T de2_pre, through ReLU
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.t_e2, HT, own, bufA, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const f32x4 v = gate4(acc[s], g_e1[s]);
                    st4(bufB + qidx(own[s], q, j), v);
                    if (live) stg4(a.d_e1_pre + row * Hp + own[s] * 16 + 4 * q, v);
                }
        }
        __syncthreads();
        // B6: ds = We0_s^T de1_pre + (decoder/prior ds), then the Gumbel-softmax backward + the KL(s) gradient, through
        //     LeakyReLU;   dh_prev += We0_h^T de1_pre
        if (KT == 1) {
            // K <= 16: wave 0 holds the ds tile with k = 4q+r in lane (q, j): the softmax backward is an in-lane sum
            // over r plus two xor-shuffles over q, on operands requested at the top of the step
            if (wid == 0) {
                const int nt0[1] = {0};
                f32x4 ds[1] = {zero4};
                mma_acc<1>(ds, Wb + a.w.t_e0s, HT, nt0, bufB, ln);
                ds[0] += k_ds;
                float dot = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) dot += k_s[r] * ds[0][r];
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                f32x4 dl;
                const float inv_temp = sc_rcp(a.temp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = k_s[r] * (ds[0][r] - dot) * inv_temp + k_kl[r];
                    v *= k_lg[r] > 0.f ? 1.f : 0.01f;
                    dl[r] = (4 * q + r < a.K && live) ? v : 0.f;
                }
                st4(dlgb + qidx(0, q, j), dl);
                if (live) stg4(a.d_lg_pre + row * Kp + 4 * q, dl);
            }
            mma_acc<MAXS>(dhacc, Wb + a.w.t_e0h, HT, own, bufB, ln);
        } else {
            {
                const Src src[1] = {{Wb + a.w.t_e0s, bufB, HT}};
                layer<1>(KT, src, wid, NW, ln, [&](int nt, f32x4 v) {
                    if (live) v += ldg4(a.d_emb + row * Ep + Hp + nt * 16 + 4 * q);
                    st4(dsb + qidx(nt, q, j), v);
                });
            }
            mma_acc<MAXS>(dhacc, Wb + a.w.t_e0h, HT, own, bufB, ln);
            __syncthreads();
            if (tid < 16) {
                const int jj = tid, bb = blockIdx.x * 16 + jj;
                const bool lv = bb < B;
                const size_t r = (size_t)t * B + (lv ? bb : 0);
                float dot = 0.f;
                for (int k = 0; k < a.K; ++k)
                    dot += (lv ? a.emb[r * Ep + Hp + k] : 0.f) * dsb[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)];
                for (int k = 0; k < Kp; ++k) {
                    float dl = 0.f;
                    if (k < a.K && lv) {
                        const float sv = a.emb[r * Ep + Hp + k];
                        const float ds = dsb[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)];
                        dl = sv * (ds - dot) / a.temp + a.dlg_kl[r * Kp + k];
                        dl *= a.logits[r * Kp + k] > 0.f ? 1.f : 0.01f;
                        a.d_lg_pre[r * Kp + k] = dl;
                    }
                    dlgb[qidx(k >> 4, (k >> 2) & 3, jj) + (k & 3)] = dl;
                }
            }
        }
        __syncthreads();
        // B7: da_s = Wxs2^T dlogits_pre, through ReLU
        {
            f32x4 acc[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[s] = zero4;
            mma_acc<MAXS>(acc, Wb + a.w.t_lg, KT, own, dlgb, ln);
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (ownok[s]) {
                    const f32x4 v = gate4(acc[s], g_as[s]);
                    st4(bufA + qidx(own[s], q, j), v);
                    if (live) stg4(a.d_as_pre + row * Hp + own[s] * 16 + 4 * q, v);
                }
        }
        __syncthreads();
        // B8: dh_{t-1} = Whh^T.. + We0_h^T.. + Wxs0_h^T das_pre + (decoder/prior dh_prev)
        mma_acc<MAXS>(dhacc, Wb + a.w.t_xsh, HT, own, bufA, ln);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (ownok[s]) st4(dhb + qidx(own[s], q, j), dhacc[s] + d_hx[s]);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------- packing
__global__ void pack_weights_kernel(const PackJob* jobs, float* dst_base) {
    const PackJob jb = jobs[blockIdx.y];
    const int KB = jb.Ip >> 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < jb.Op * jb.Ip; e += gridDim.x * blockDim.x) {
        // e enumerates the PACKED order: (((nt*KB + kb)*64 + lane)*4 + c)
        const int c = e & 3, ln = (e >> 2) & 63, blk = e >> 8;
        const int kb = blk % KB, nt = blk / KB;
        const int o = nt * 16 + (ln & 15), i = kb * 16 + (ln >> 4) * 4 + c;
        const int ob = o / jb.oblk_pad, orr = o % jb.oblk_pad, ib = i / jb.iblk_pad, irr = i % jb.iblk_pad;
        float v = 0.f;
        if (orr < jb.oblk_true && irr < jb.iblk_true) {
            const int so = ob * jb.oblk_true + orr, si = ib * jb.iblk_true + irr;
            v = jb.transposed ? jb.src[(size_t)si * jb.ld + so] : jb.src[(size_t)so * jb.ld + si];
        }
        dst_base[jb.dst_off + e] = v;
    }
}

__global__ void pack_bias_kernel(const PackJob* jobs, float* dst_base) {
    const PackJob jb = jobs[blockIdx.y];
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < jb.Op; o += gridDim.x * blockDim.x) {
        const int ob = o / jb.oblk_pad, orr = o % jb.oblk_pad;
        float v = 0.f;
        if (orr < jb.oblk_true) {
            const int so = ob * jb.oblk_true + orr;
            v = jb.src[so] + (jb.src2 ? jb.src2[so] : 0.f);
        }
        dst_base[jb.dst_off + o] = v;
    }
}

int launch_pack(const PackJob* wjobs_dev, int nw, const PackJob* bjobs_dev, int nb, float* dst, hipStream_t stream) {
    if (nw > 0) hipLaunchKernelGGL(pack_weights_kernel, dim3(64, nw), dim3(256), 0, stream, wjobs_dev, dst);
    if (nb > 0) hipLaunchKernelGGL(pack_bias_kernel, dim3(4, nb), dim3(256), 0, stream, bjobs_dev, dst);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------------------------- launch
static int scan_geometry(const ScanArgs& a, int& nw, int& maxs, bool bwd = false) {
    // waves per workgroup: one per 16 hidden units up to 16 waves (4 per SIMD hide the L2 latency of
    // the streamed weights); each wave keeps ceil(HT/nw) cell-state tiles in registers
    const int HT = a.Hp / 16;
    nw = 1;
    while (nw * 2 <= HT && nw < 16) nw *= 2;
    static const int force = [] { const char* f = getenv("SHIVAE_SCAN_WAVES"); return f ? atoi(f) : 0; }();
    if (force == 1 || force == 2 || force == 4 || force == 8 || force == 16) nw = std::min(nw, force);
    // Both chains run best on 8 waves of up to 256 registers (B=4096, T=1000, H=256: forward 79.1 ms against 82.6 ms
    // on 16 waves, backward 93.2 against 104.3): with 128 registers per wave the prefetched records and the pipelined
    // weight fetch spill.
    else nw = std::min(nw, 8);
    (void)bwd;
    const int need = ceil_div(HT, nw);
    maxs = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    SV_REQUIRE(need <= 8, "scan: h_dim %d too large for the register-resident cell", a.H);
    return SHIVAE_OK;
}

size_t scan_fwd_lds_bytes(const ScanArgs& a) { return sizeof(float) * 16 * (4 * a.Hp + 2 * a.Kp + a.Zp); }
size_t scan_bwd_lds_bytes(const ScanArgs& a) { return sizeof(float) * 16 * (7 * a.Hp + 2 * a.Kp + 2 * a.Zp); }

template <int MAXS, int NTH, bool WLDS = false>
static int launch_scan_t(const ScanArgs& a, bool bwd, int nw, hipStream_t stream) {
    const size_t lds = (bwd ? scan_bwd_lds_bytes(a) : scan_fwd_lds_bytes(a)) + (WLDS ? sizeof(float) * a.wlds_floats : 0);
    SV_REQUIRE(lds <= 160 * 1024, "scan: needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = bwd ? scan_bwd_kernel<MAXS, NTH, WLDS> : scan_fwd_kernel<MAXS, NTH, WLDS>;
    if (lds > 64 * 1024)
        SV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.B, 16)), dim3(nw * 64), lds, stream, a);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_scan(const ScanArgs& a, bool bwd, hipStream_t stream) {
    if (a.T <= 0 || a.B <= 0) return SHIVAE_OK;
    int nw, maxs;
    SV_TRY(scan_geometry(a, nw, maxs, bwd));
    if (nw == 8) {
        switch (maxs) {
            case 1: return launch_scan_t<1, 512>(a, bwd, nw, stream);
            case 2: return launch_scan_t<2, 512>(a, bwd, nw, stream);
            case 4: return launch_scan_t<4, 512>(a, bwd, nw, stream);
            default: break;
        }
        set_error("scan: unsupported geometry nw=%d maxs=%d", nw, maxs);
        return SHIVAE_ERR_INVALID;
    }
    if (nw > 4) {
        switch (maxs) {
            case 1: return launch_scan_t<1, 1024>(a, bwd, nw, stream);
            case 2: return launch_scan_t<2, 1024>(a, bwd, nw, stream);
            default: break;
        }
        set_error("scan: unsupported geometry nw=%d maxs=%d", nw, maxs);
        return SHIVAE_ERR_INVALID;
    }
    if (maxs == 1) {
        // small model (h_dim <= 64): keep the packed weights in LDS when they fit beside the activations
        static const bool wlds = [] { const char* f = getenv("SHIVAE_SCAN_WLDS"); return !f || atoi(f) != 0; }();
        const size_t tot = (bwd ? scan_bwd_lds_bytes(a) : scan_fwd_lds_bytes(a)) + sizeof(float) * a.wlds_floats;
        if (wlds && a.wlds_floats > 0 && a.wlds_floats % 4 == 0 && tot <= 96 * 1024)
            return launch_scan_t<1, 256, true>(a, bwd, nw, stream);
    }
    switch (maxs) {
        case 1: return launch_scan_t<1, 256>(a, bwd, nw, stream);
        case 2: return launch_scan_t<2, 256>(a, bwd, nw, stream);
        case 4: return launch_scan_t<4, 256>(a, bwd, nw, stream);
        default: return launch_scan_t<8, 256>(a, bwd, nw, stream);
    }
}

template <int MAXS, int NTH>
static int launch_gen_t(const ScanArgs& a, int nw, hipStream_t stream) {
    const size_t lds = sizeof(float) * 16 * (4 * a.Hp + 2 * a.Kp + a.Zp);
    SV_REQUIRE(lds <= 160 * 1024, "scan: needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = scan_gen_kernel<MAXS, NTH>;
    if (lds > 64 * 1024)
        SV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.B, 16)), dim3(nw * 64), lds, stream, a);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_scan_gen(const ScanArgs& a, hipStream_t stream) {
    if (a.T <= 0 || a.B <= 0) return SHIVAE_OK;
    int nw, maxs;
    SV_TRY(scan_geometry(a, nw, maxs));
    if (nw > 4) {  // generation is not a hot path: one register budget (<= 8 waves) covers every geometry
        nw = 8;
        maxs = ceil_div(a.Hp / 16, nw) <= 1 ? 1 : ceil_div(a.Hp / 16, nw) <= 2 ? 2 : 4;
        switch (maxs) {
            case 1: return launch_gen_t<1, 512>(a, nw, stream);
            case 2: return launch_gen_t<2, 512>(a, nw, stream);
            default: return launch_gen_t<4, 512>(a, nw, stream);
        }
    }
    switch (maxs) {
        case 1: return launch_gen_t<1, 256>(a, nw, stream);
        case 2: return launch_gen_t<2, 256>(a, nw, stream);
        case 4: return launch_gen_t<4, 256>(a, nw, stream);
        default: return launch_gen_t<8, 256>(a, nw, stream);
    }
}

// ---------------------------------------------------------------------------------- noise as tensors
// The streams the scans draw in-kernel under `seed` (noise.h), written out: u[rows][K], eps[rows][Z].  For tests and
// for callers that want to look at, or replay, the noise of a step.
__global__ __launch_bounds__(256) void noise_fill_kernel(unsigned long long seed, long long rows, int K, int Z, float* __restrict__ u,
                                                         float* __restrict__ eps) {
    const int gk = (K + 3) >> 2, gz = (Z + 3) >> 2, per = gk + gz;
    const long long total = rows * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / per;
        const int g = (int)(i % per);
        if (g < gk) {
            if (!u) continue;
            const SvF4 t = sv_noise_u4(seed, (unsigned long long)row, g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < K) u[row * K + 4 * g + r] = t.v[r];
        } else {
            if (!eps) continue;
            const int gg = g - gk;
            const SvF4 t = sv_noise_eps4(seed, (unsigned long long)row, gg);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * gg + r < Z) eps[row * Z + 4 * gg + r] = t.v[r];
        }
    }
}

int launch_noise_fill(unsigned long long seed, long long rows, int K, int Z, float* u, float* eps, hipStream_t stream) {
    if (rows <= 0 || (K <= 0 && Z <= 0)) return SHIVAE_OK;
    const long long total = rows * (((K + 3) >> 2) + ((Z + 3) >> 2));
    const int grid = (int)std::min<long long>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(noise_fill_kernel, dim3(grid), dim3(256), 0, stream, seed, rows, K, Z, u, eps);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
