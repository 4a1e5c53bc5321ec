// Mask-aware heterogeneous likelihood + KL reduction kernels (north_star kernel 3) and the
// standalone reparameterisation kernels (north_star kernel 2).  HBM-streaming, fp32.
//
// Reference: base_missvae.py:233-285 heter_nll, :287-318 kl_q_z, :320-343 kl_q_s, :166-231
// compute_loss; samplers.py:23-62.
//
// Row = one (t, b) pair.  A wave handles 64/ap rows at a time with ap (a power of two) lanes per
// row striding over the attributes (resp. the z / s components); per-row partial sums are reduced
// with wave shuffles and written as per-timestep terms rowloss[row][6] =
// {nll_real, nll_pos, nll_bin, nll_cat, kl_z, kl_s}; a second, deterministic pass sums them over t.
// The backward of the loss wrt the decoder outputs / z statistics / s logits is fused in the same
// pass (one read of x, mask and the parameters serves both directions).
#include <algorithm>
#include <cstdlib>

#include "cell_math.h"
#include "common.h"

namespace shivae {

struct NllParams {
    const float* x; const uint8_t* mask; const float* P; float* dP; float* rowloss;
    const AttrDesc* attrs;
    int ldx, ldm, ldp, ldd, rows, D, learn_std, ap, want_grad;
    float gs;
};

__global__ __launch_bounds__(256) void heter_nll_kernel(NllParams p) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int rpw = 64 / p.ap, sub = lane / p.ap, al = lane % p.ap;
    for (int base = wave * rpw; base < p.rows; base += nwaves * rpw) {
        const int row = base + sub;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < p.rows) {
            const float* xr = p.x + (size_t)row * p.ldx;
            const uint8_t* mr = p.mask + (size_t)row * p.ldm;
            const float* pr = p.P + (size_t)row * p.ldp;
            float* dr = p.dP ? p.dP + (size_t)row * p.ldd : nullptr;
            for (int d = al; d < p.D; d += p.ap) {
                const AttrDesc a = p.attrs[d];
                // base_missvae.py:271-274: one mask column; for cat the LAST one-hot column
                const bool obs = mr[a.xoff + a.dim - 1] == 0;
                const float g = obs ? p.gs : 0.f;
                float nll;
                if (a.type <= 1) {  // real / pos: Normal(mu, std)
                    const float mu = pr[a.poff];
                    const float sd = p.learn_std ? sv_softplus(pr[a.poff + 1]) : SV_FIXED_STD;
                    float dmu, dsd;
                    nll = sv_nll_normal(xr[a.xoff], mu, sd, dmu, dsd);
                    if (p.want_grad) {
                        dr[a.poff] = obs ? dmu * g : 0.f;
                        if (p.learn_std) dr[a.poff + 1] = obs ? dsd * g * sv_softplus_grad_from_out(sd) : 0.f;
                    }
                } else if (a.type == 2) {  // bin: Bernoulli(logits)
                    float dl;
                    nll = sv_nll_bernoulli(xr[a.xoff], pr[a.poff], dl);
                    if (p.want_grad) dr[a.poff] = obs ? dl * g : 0.f;
                } else {  // cat: OneHotCategorical(logits = [0, theta_1..theta_{C-1}])
                    const int C = a.dim;
                    float m = 0.f, xbest = xr[a.xoff];
                    int idx = 0;
                    for (int c = 1; c < C; ++c) {
                        m = fmaxf(m, pr[a.poff + c - 1]);
                        const float xv = xr[a.xoff + c];
                        if (xv > xbest) { xbest = xv; idx = c; }  // first max, as torch.max(-1)[1]
                    }
                    float se = expf(-m);
                    for (int c = 1; c < C; ++c) se += expf(pr[a.poff + c - 1] - m);
                    const float lse = m + logf(se);
                    nll = lse - (idx == 0 ? 0.f : pr[a.poff + idx - 1]);
                    if (p.want_grad) {
                        const float inv = 1.f / se;
                        for (int c = 1; c < C; ++c) {
                            const float sm = expf(pr[a.poff + c - 1] - m) * inv;
                            dr[a.poff + c - 1] = obs ? (sm - (c == idx ? 1.f : 0.f)) * g : 0.f;
                        }
                    }
                }
                acc[a.type] += obs ? nll : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            for (int o = p.ap >> 1; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
        if (row < p.rows && al == 0) {
            float* rl = p.rowloss + (size_t)row * 6;
            rl[0] = acc[0]; rl[1] = acc[1]; rl[2] = acc[2]; rl[3] = acc[3];
        }
    }
}

static int pow2_lanes(int n) {
    int ap = 1;
    while (ap < n && ap < 64) ap <<= 1;
    return ap;
}

// Streaming form.  Unit of work = (row, group of GA consecutive attributes, GA a multiple of 64 chosen so
// that a group's spans fit a few KB): the attributes of a group occupy one contiguous span of the x
// row, of the mask row and of the parameter row.  A wave pulls the 16-B aligned supersets of the three
// spans into its private LDS slice with coalesced 16-B loads, each lane evaluates its attributes out of
// LDS (descriptor table in LDS, shared by the workgroup), the parameter gradients overwrite the staged
// parameters and leave as 16-B stores.  No workgroup barrier inside the loop: waves run independently,
// 24 per CU, which is what hides the HBM latency.  Transcendentals use the hardware-rate forms of
// cell_math.h.  Row sums: in-lane, then a fixed xor-shuffle tree (deterministic).
struct NllStreamParams {
    NllParams b;
    int X, P, GA;
    int xcap, pcap, mcap;  // LDS floats / floats / bytes per wave
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool FAST>
__global__ __launch_bounds__(256, 6) void heter_nll_stream_kernel(NllStreamParams q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nll_smem[];
    const NllParams& p = q.b;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int D = p.D, GA = q.GA;
    int4* at = reinterpret_cast<int4*>(nll_smem);  // {type, xoff, dim, poff}
    unsigned char* slice = nll_smem + (size_t)D * 16 + (size_t)wid * (4 * (q.xcap + q.pcap) + q.mcap);
    float* xs = reinterpret_cast<float*>(slice);
    float* ps = xs + q.xcap;
    unsigned char* ms = reinterpret_cast<unsigned char*>(ps + q.pcap);
    for (int i = tid; i < D; i += 256) {
        const AttrDesc ad = p.attrs[i];
        at[i] = make_int4(ad.type, ad.xoff, ad.dim, ad.poff);
    }
    __syncthreads();

    const int gw = blockIdx.x * 4 + wid, nw = gridDim.x * 4;
    for (int row = gw; row < p.rows; row += nw) {
        const float* xrow = p.x + (size_t)row * p.ldx;
        const float* prow = p.P + (size_t)row * p.ldp;
        const uint8_t* mrow = p.mask + (size_t)row * p.ldm;
        float* drow = p.want_grad ? p.dP + (size_t)row * p.ldd : nullptr;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int g0 = 0; g0 < D; g0 += GA) {
            const int g1 = min(g0 + GA, D);
            const int4 first = at[g0];
            const int x0 = first.y, p0 = first.w;
            const int x1 = g1 < D ? at[g1].y : q.X;
            const int p1 = g1 < D ? at[g1].w : q.P;
            // aligned supersets: element i of the span sits at LDS index delta + i
            const float* xsrc = xrow + x0;
            const float* psrc = prow + p0;
            const uint8_t* msrc = mrow + x0;
            const int dx = (int)((reinterpret_cast<uintptr_t>(xsrc) >> 2) & 3);
            const int dp = (int)((reinterpret_cast<uintptr_t>(psrc) >> 2) & 3);
            const int dm = (int)(reinterpret_cast<uintptr_t>(msrc) & 15);
            const int np4 = (dp + (p1 - p0) + 3) >> 2, nx4 = (dx + (x1 - x0) + 3) >> 2, nm4 = (dm + (x1 - x0) + 15) >> 4;
            {
                const float4* s4 = reinterpret_cast<const float4*>(psrc - dp);
                for (int i = lane; i < np4; i += 64) reinterpret_cast<float4*>(ps)[i] = s4[i];
                const float4* t4 = reinterpret_cast<const float4*>(xsrc - dx);
                for (int i = lane; i < nx4; i += 64) reinterpret_cast<float4*>(xs)[i] = t4[i];
                const uint4* u4 = reinterpret_cast<const uint4*>(msrc - dm);
                for (int i = lane; i < nm4; i += 64) reinterpret_cast<uint4*>(ms)[i] = u4[i];
            }
            wave_lds_sync();
            for (int d = g0 + lane; d < g1; d += 64) {
                const int4 ad = at[d];
                const int type = ad.x, dim = ad.z;
                const float* xr = xs + dx + (ad.y - x0);
                float* pr = ps + dp + (ad.w - p0);
                // base_missvae.py:271-274: one mask column; for cat the LAST one-hot column
                const bool obs = ms[dm + (ad.y - x0) + dim - 1] == 0;
                const float g = obs ? p.gs : 0.f;
                float nll;
                if (type <= 1) {  // real / pos: Normal(mu, std)
                    const float mu = pr[0];
                    const float sd = p.learn_std ? (FAST ? sv_softplus_fast(pr[1]) : sv_softplus(pr[1])) : SV_FIXED_STD;
                    float dmu, dsd;
                    nll = FAST ? sv_nll_normal_fast(xr[0], mu, sd, dmu, dsd) : sv_nll_normal(xr[0], mu, sd, dmu, dsd);
                    if (p.want_grad) {
                        pr[0] = obs ? dmu * g : 0.f;
                        if (p.learn_std) pr[1] = obs ? dsd * g * (FAST ? sv_softplus_grad_from_out_fast(sd) : sv_softplus_grad_from_out(sd)) : 0.f;
                    }
                } else if (type == 2) {  // bin: Bernoulli(logits)
                    float dl;
                    nll = FAST ? sv_nll_bernoulli_fast(xr[0], pr[0], dl) : sv_nll_bernoulli(xr[0], pr[0], dl);
                    if (p.want_grad) pr[0] = obs ? dl * g : 0.f;
                } else {  // cat: OneHotCategorical(logits = [0, theta_1..theta_{C-1}])
                    float m = 0.f, xbest = xr[0], tsel = 0.f;
                    int idx = 0;
#pragma unroll 4
                    for (int c = 1; c < dim; ++c) {
                        const float th = pr[c - 1], xv = xr[c];
                        m = fmaxf(m, th);
                        if (xv > xbest) { xbest = xv; idx = c; tsel = th; }  // first max, as torch.max(-1)[1]
                    }
                    float se = FAST ? sv_fexp(-m) : expf(-m);
#pragma unroll 4
                    for (int c = 1; c < dim; ++c) {
                        const float e = FAST ? sv_fexp(pr[c - 1] - m) : expf(pr[c - 1] - m);
                        pr[c - 1] = e;  // the staged copy now holds exp(theta - m)
                        se += e;
                    }
                    nll = m + (FAST ? sv_flog(se) : logf(se)) - tsel;
                    if (p.want_grad) {
                        const float inv = FAST ? sv_frcp(se) : 1.f / se;
#pragma unroll 4
                        for (int c = 1; c < dim; ++c) pr[c - 1] = obs ? (pr[c - 1] * inv - (c == idx ? 1.f : 0.f)) * g : 0.f;
                    }
                }
                const float v = obs ? nll : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += type == c ? v : 0.f;
            }
            wave_lds_sync();
            if (p.want_grad) {
                float* dsrc = drow + p0;
                const int dd = (int)((reinterpret_cast<uintptr_t>(dsrc) >> 2) & 3);
                if (dd == dp) {
                    // same 16-B phase as the staged copy: whole float4 pieces inside the span, scalars at its edges
                    float4* d4 = reinterpret_cast<float4*>(dsrc - dd);
                    for (int i = lane; i < np4; i += 64) {
                        const int lo = i * 4 - dp;  // span index of the piece's first element
                        if (lo >= 0 && lo + 4 <= p1 - p0) d4[i] = reinterpret_cast<const float4*>(ps)[i];
                        else
                            for (int c = 0; c < 4; ++c)
                                if (lo + c >= 0 && lo + c < p1 - p0) dsrc[lo + c] = ps[dp + lo + c];
                    }
                } else {
                    for (int i = lane; i < p1 - p0; i += 64) dsrc[i] = ps[dp + i];
                }
            }
            wave_lds_sync();  // the next staging overwrites this slice
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
        if (lane == 0) {
            float* rl = p.rowloss + (size_t)row * 6;
            rl[0] = acc[0]; rl[1] = acc[1]; rl[2] = acc[2]; rl[3] = acc[3];
        }
    }
}

int launch_heter_nll(const AttrDesc* attrs_dev, int D, int learn_std, const float* x, int ldx,
                     const uint8_t* mask, int ldm, const float* P, int ldp, float* dP, int ldd, int rows,
                     float gs, float* rowloss, int X, int Pw, int group, int xspan, int pspan, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    NllParams p;
    p.x = x; p.mask = mask; p.P = P; p.dP = dP; p.rowloss = rowloss; p.attrs = attrs_dev;
    p.ldx = ldx; p.ldm = ldm; p.ldp = ldp; p.ldd = ldd; p.rows = rows; p.D = D; p.learn_std = learn_std;
    p.ap = pow2_lanes(D); p.want_grad = dP != nullptr; p.gs = gs;
    // group: attributes per staging step; xspan / pspan: widest x / parameter span of a group (plan_create)
    NllStreamParams q;
    q.b = p; q.X = X; q.P = Pw; q.GA = group;
    q.xcap = round_up(xspan + 3, 4) + 4; q.pcap = round_up(pspan + 3, 4) + 4; q.mcap = round_up(xspan + 15, 16) + 16;
    const size_t lds = (size_t)D * 16 + 4 * ((size_t)4 * (q.xcap + q.pcap) + q.mcap);
    if (group > 0 && xspan > 0 && pspan > 0 && lds <= 64 * 1024) {
        // 4 rounds of the workgroups that fit the chip (LDS; 6 waves per SIMD): enough rows per wave to
        // amortise the descriptor table, short enough rounds that a mis-estimated residency costs a few
        // per cent.  (One-wave workgroups reading descriptors from L2 measured 15 % slower.)
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(6, (160 * 1024) / lds));
        // (whole rows per wave: every wave of the strided loop gets the same count, so a launch of a few tens of
        // thousands of rows has no half-empty last round)
        const int rpw = ceil_div(rows, 4 * 256 * per_cu * 4);
        const int grid = ceil_div(rows, 4 * rpw);
        static const bool exact = getenv("SHIVAE_NLL_EXACT") != nullptr;
        if (exact) hipLaunchKernelGGL(heter_nll_stream_kernel<false>, dim3(grid), dim3(256), lds, stream, q);
        else hipLaunchKernelGGL(heter_nll_stream_kernel<true>, dim3(grid), dim3(256), lds, stream, q);
        SV_CHECK_HIP(hipGetLastError());
        return SHIVAE_OK;
    }
    const int rows_per_block = 4 * (64 / p.ap);
    const int blocks = std::min(ceil_div(rows, rows_per_block), 256 * 8);
    hipLaunchKernelGGL(heter_nll_kernel, dim3(blocks), dim3(256), 0, stream, p);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- KL(q(z)||p(z)) + KL(q(s)||U)
struct KlParams {
    const float *mu, *sd, *mup, *sdp, *probs;
    float *dmu, *dsd, *dmup, *dsdp_pre, *dlg, *rowloss;
    int ldz, ldk, rows, Z, K, ap, want_grad;
    float kz;  // beta * grad_scale
};

__global__ __launch_bounds__(256) void kl_kernel(KlParams p) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int rpw = 64 / p.ap, sub = lane / p.ap, al = lane % p.ap;
    const float logq = logf(fminf(fmaxf(1.f / (float)p.K, SV_F32_EPS), 1.f - SV_F32_EPS));
    for (int base = wave * rpw; base < p.rows; base += nwaves * rpw) {
        const int row = base + sub;
        const bool live = row < p.rows;
        const size_t oz = (size_t)row * p.ldz, ok = (size_t)row * p.ldk;
        float klz = 0.f;
        if (live)
            for (int z = al; z < p.Z; z += p.ap) {
                float a, b, c, d;
                klz += sv_kl_normal(p.mu[oz + z], p.sd[oz + z], p.mup[oz + z], p.sdp[oz + z], a, b, c, d);
                if (p.want_grad) {
                    p.dmu[oz + z] = a * p.kz;
                    p.dsd[oz + z] = b * p.kz;
                    p.dmup[oz + z] = c * p.kz;
                    p.dsdp_pre[oz + z] = d * p.kz * sv_softplus_grad_from_out(p.sdp[oz + z]);
                }
            }
        // Categorical(probs): renormalise, clamp, KL against the uniform prior (base_missvae.py:332-343)
        float S = 0.f;
        if (live) for (int k = al; k < p.K; k += p.ap) S += p.probs[ok + k];
        for (int o = p.ap >> 1; o > 0; o >>= 1) S += __shfl_xor(S, o);
        float kls = 0.f, pd = 0.f;
        if (live)
            for (int k = al; k < p.K; k += p.ap) {
                const float pk = p.probs[ok + k] / S;
                const bool inr = pk >= SV_F32_EPS && pk <= 1.f - SV_F32_EPS;
                const float lp = logf(fminf(fmaxf(pk, SV_F32_EPS), 1.f - SV_F32_EPS));
                if (pk != 0.f) {
                    kls += pk * (lp - logq);
                    pd += pk * ((lp - logq) + (inr ? 1.f : 0.f));
                }
            }
        for (int o = p.ap >> 1; o > 0; o >>= 1) {
            klz += __shfl_xor(klz, o);
            kls += __shfl_xor(kls, o);
            pd += __shfl_xor(pd, o);
        }
        if (kls != kls) kls = 0.f;
        if (p.want_grad) {
            // d/dprobs, then through probs = softmax(logits): dlg_j = pi_j (dpi_j - sum_k pi_k dpi_k)
            float sdot = 0.f;
            if (live)
                for (int k = al; k < p.K; k += p.ap) {
                    const float pi = p.probs[ok + k], pk = pi / S;
                    const bool inr = pk >= SV_F32_EPS && pk <= 1.f - SV_F32_EPS;
                    const float lp = logf(fminf(fmaxf(pk, SV_F32_EPS), 1.f - SV_F32_EPS));
                    const float dkl = pk != 0.f ? (lp - logq) + (inr ? 1.f : 0.f) : 0.f;
                    sdot += pi * (dkl - pd) / S;
                }
            for (int o = p.ap >> 1; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
            if (live)
                for (int k = al; k < p.K; k += p.ap) {
                    const float pi = p.probs[ok + k], pk = pi / S;
                    const bool inr = pk >= SV_F32_EPS && pk <= 1.f - SV_F32_EPS;
                    const float lp = logf(fminf(fmaxf(pk, SV_F32_EPS), 1.f - SV_F32_EPS));
                    const float dkl = pk != 0.f ? (lp - logq) + (inr ? 1.f : 0.f) : 0.f;
                    p.dlg[ok + k] = p.kz * pi * ((dkl - pd) / S - sdot);
                }
        }
        if (live && al == 0) {
            p.rowloss[(size_t)row * 6 + 4] = klz;
            p.rowloss[(size_t)row * 6 + 5] = kls;
        }
    }
}

// 16-B form of kl_kernel: a lane owns 4 consecutive z (and 4 consecutive k) of a row, every load of a
// row group is in flight before the first use, every store is a float4.  Needs 16-B aligned bases and
// leading dimensions that are multiples of 4 with the rounded-up width inside the row.
__global__ __launch_bounds__(256) void kl_vec_kernel(KlParams p) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int rpw = 64 / p.ap, sub = lane / p.ap, al = lane % p.ap;
    const int Z4 = (p.Z + 3) >> 2, K4 = (p.K + 3) >> 2;
    const float logq = logf(fminf(fmaxf(1.f / (float)p.K, SV_F32_EPS), 1.f - SV_F32_EPS));
    for (int base = wave * rpw; base < p.rows; base += nwaves * rpw) {
        const int row = base + sub;
        const bool live = row < p.rows;
        const size_t oz = (size_t)row * p.ldz, ok = (size_t)row * p.ldk;
        float4 pk4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool hask = live && al < K4;
        if (hask) pk4 = *reinterpret_cast<const float4*>(p.probs + ok + al * 4);
        float klz = 0.f;
        if (live)
            for (int z4 = al; z4 < Z4; z4 += p.ap) {
                const size_t o = oz + z4 * 4;
                const float4 mq = *reinterpret_cast<const float4*>(p.mu + o), sq = *reinterpret_cast<const float4*>(p.sd + o);
                const float4 mp = *reinterpret_cast<const float4*>(p.mup + o), sp = *reinterpret_cast<const float4*>(p.sdp + o);
                const float mqv[4] = {mq.x, mq.y, mq.z, mq.w}, sqv[4] = {sq.x, sq.y, sq.z, sq.w};
                const float mpv[4] = {mp.x, mp.y, mp.z, mp.w}, spv[4] = {sp.x, sp.y, sp.z, sp.w};
                float ga[4], gb[4], gc[4], gd[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool v = z4 * 4 + c < p.Z;  // padded columns may hold anything
                    float a, b, cc, d;
                    const float kl = sv_kl_normal(mqv[c], sqv[c], mpv[c], spv[c], a, b, cc, d);
                    klz += v ? kl : 0.f;
                    ga[c] = v ? a * p.kz : 0.f;
                    gb[c] = v ? b * p.kz : 0.f;
                    gc[c] = v ? cc * p.kz : 0.f;
                    gd[c] = v ? d * p.kz * sv_softplus_grad_from_out(spv[c]) : 0.f;
                }
                if (p.want_grad) {
                    *reinterpret_cast<float4*>(p.dmu + o) = make_float4(ga[0], ga[1], ga[2], ga[3]);
                    *reinterpret_cast<float4*>(p.dsd + o) = make_float4(gb[0], gb[1], gb[2], gb[3]);
                    *reinterpret_cast<float4*>(p.dmup + o) = make_float4(gc[0], gc[1], gc[2], gc[3]);
                    *reinterpret_cast<float4*>(p.dsdp_pre + o) = make_float4(gd[0], gd[1], gd[2], gd[3]);
                }
            }
        // Categorical(probs): renormalise, clamp, KL against the uniform prior (base_missvae.py:332-343)
        const float piv[4] = {pk4.x, pk4.y, pk4.z, pk4.w};
        bool kv[4];
        float S = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { kv[c] = hask && al * 4 + c < p.K; S += kv[c] ? piv[c] : 0.f; }
        for (int o = p.ap >> 1; o > 0; o >>= 1) S += __shfl_xor(S, o);
        float kls = 0.f, pd = 0.f, dkl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float pk = kv[c] ? piv[c] / S : 0.f;
            const bool inr = pk >= SV_F32_EPS && pk <= 1.f - SV_F32_EPS;
            const float lp = logf(fminf(fmaxf(pk, SV_F32_EPS), 1.f - SV_F32_EPS));
            dkl[c] = (kv[c] && pk != 0.f) ? (lp - logq) + (inr ? 1.f : 0.f) : 0.f;
            if (kv[c] && pk != 0.f) { kls += pk * (lp - logq); pd += pk * dkl[c]; }
        }
        for (int o = p.ap >> 1; o > 0; o >>= 1) {
            klz += __shfl_xor(klz, o);
            kls += __shfl_xor(kls, o);
            pd += __shfl_xor(pd, o);
        }
        if (kls != kls) kls = 0.f;
        if (p.want_grad) {
            // d/dprobs, then through probs = softmax(logits): dlg_j = pi_j (dpi_j - sum_k pi_k dpi_k)
            float sdot = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) sdot += kv[c] ? piv[c] * (dkl[c] - pd) / S : 0.f;
            for (int o = p.ap >> 1; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
            if (hask) {
                float g[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) g[c] = kv[c] ? p.kz * piv[c] * ((dkl[c] - pd) / S - sdot) : 0.f;
                *reinterpret_cast<float4*>(p.dlg + ok + al * 4) = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
        if (live && al == 0) {
            p.rowloss[(size_t)row * 6 + 4] = klz;
            p.rowloss[(size_t)row * 6 + 5] = kls;
        }
    }
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int launch_kl(const float* mu, const float* sd, const float* mup, const float* sdp, int ldz,
              const float* probs, int ldk, float* dmu, float* dsd, float* dmup, float* dsdp_pre, float* dlg,
              int rows, int Z, int K, float kz, float* rowloss, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    KlParams p;
    p.mu = mu; p.sd = sd; p.mup = mup; p.sdp = sdp; p.probs = probs;
    p.dmu = dmu; p.dsd = dsd; p.dmup = dmup; p.dsdp_pre = dsdp_pre; p.dlg = dlg; p.rowloss = rowloss;
    p.ldz = ldz; p.ldk = ldk; p.rows = rows; p.Z = Z; p.K = K;
    p.want_grad = dmu != nullptr; p.kz = kz;
    const bool vec = ldz % 4 == 0 && ldk % 4 == 0 && round_up(Z, 4) <= ldz && round_up(K, 4) <= ldk && al16(mu) &&
                     al16(sd) && al16(mup) && al16(sdp) && al16(probs) &&
                     (!p.want_grad || (al16(dmu) && al16(dsd) && al16(dmup) && al16(dsdp_pre) && al16(dlg)));
    p.ap = vec ? pow2_lanes(std::max(ceil_div(Z, 4), ceil_div(K, 4))) : pow2_lanes(std::max(Z, K));
    const int rows_per_block = 4 * (64 / p.ap);
    const int blocks = std::min(ceil_div(rows, rows_per_block), 256 * 8);
    if (vec) hipLaunchKernelGGL(kl_vec_kernel, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(kl_kernel, dim3(blocks), dim3(256), 0, stream, p);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- sum over time, then over batch
// persum[b][6] = sum_t rowloss[(t*B + b)][6]   (fixed order -> deterministic)
__global__ void rowloss_sum_t_kernel(const float* __restrict__ rowloss, float* __restrict__ persum, int T, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*6
    if (i >= B * 6) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += rowloss[(size_t)t * B * 6 + i];
    persum[i] = s;
}

// losses[0..8] += {n_elbo, nll_loss, kld, kld_q_z, kld_q_s, nll_real, nll_pos, nll_bin, nll_cat}
// (base_missvae.py:211-230), each a batch mean with weight inv_batch = 1/B_total.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ persum, float* losses, int B,
                                                            float beta, float inv_batch) {
    __shared__ float red[6][256];
    float s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < B; b += 256)
        for (int c = 0; c < 6; ++c) s[c] += persum[b * 6 + c];
    for (int c = 0; c < 6; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int c = 0; c < 6; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float re = red[0][0] * inv_batch, po = red[1][0] * inv_batch, bi = red[2][0] * inv_batch,
                    ca = red[3][0] * inv_batch, kz = red[4][0] * inv_batch, ks = red[5][0] * inv_batch;
        const float nll = re + po + bi + ca;
        losses[0] += nll + beta * kz + beta * ks;
        losses[1] += nll;
        losses[2] += kz + ks;
        losses[3] += kz;
        losses[4] += ks;
        losses[5] += re;
        losses[6] += po;
        losses[7] += bi;
        losses[8] += ca;
    }
}

int launch_loss_reduce(const float* rowloss, float* persum, int T, int B, float beta, float inv_batch,
                       float* losses, hipStream_t stream) {
    hipLaunchKernelGGL(rowloss_sum_t_kernel, dim3(ceil_div(B * 6, 256)), dim3(256), 0, stream, rowloss, persum,
                       T, B);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, persum, losses, B, beta, inv_batch);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- Monte-Carlo imputation samples
// base_missvae.py:378-404 generate_heter_samples: per attribute the mean of n_samples draws from its
// likelihood (rounded for bin / cat), written in the x layout.  One thread per (row, attribute); draws come
// from a counter-based generator keyed by (seed, row, attribute, sample) so the result does not depend on
// the launch geometry.  The reference draws from torch's device generator: streams cannot match, the
// distribution of the result does (tests compare against the exact Normal / Binomial laws).
__device__ __forceinline__ uint64_t mc_mix(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float mc_u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.f / 16777216.f); }  // (0,1)

struct McParams {
    const AttrDesc* attrs;
    const float* P;
    float* xhat;
    long long rows;
    int D, ldp, ldx, learn_std, n;
    uint64_t seed;
};

__global__ __launch_bounds__(256) void mc_impute_kernel(McParams p) {
    const long long total = p.rows * p.D, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long row = i / p.D;
        const int d = (int)(i - row * p.D);
        const AttrDesc a = p.attrs[d];
        const float* pr = p.P + row * p.ldp + a.poff;
        float* xo = p.xhat + row * p.ldx + a.xoff;
        const uint64_t key = mc_mix(p.seed ^ mc_mix((uint64_t)i + 0x9E3779B97F4A7C15ull));
        if (a.type <= 1) {  // Normal(mu, std): mean of n draws
            const float mu = pr[0], sd = p.learn_std ? sv_softplus(pr[1]) : SV_FIXED_STD;
            float acc = 0.f;
            for (int s = 0; s < p.n; s += 2) {  // Box-Muller, two normals per counter
                const uint64_t r = mc_mix(key + (uint64_t)(s >> 1) * 0xD1342543DE82EF95ull);
                const float rad = sqrtf(-2.f * logf(mc_u01((uint32_t)r))), ang = 6.283185307179586f * mc_u01((uint32_t)(r >> 32));
                acc += rad * cosf(ang);
                if (s + 1 < p.n) acc += rad * sinf(ang);
            }
            xo[0] = mu + sd * acc / (float)p.n;
        } else if (a.type == 2) {  // Bernoulli(logits): round(mean of n draws)
            const float pb = sv_sigmoid(pr[0]);
            int cnt = 0;
            for (int s = 0; s < p.n; s += 2) {
                const uint64_t r = mc_mix(key + (uint64_t)(s >> 1) * 0xD1342543DE82EF95ull);
                cnt += mc_u01((uint32_t)r) < pb;
                if (s + 1 < p.n) cnt += mc_u01((uint32_t)(r >> 32)) < pb;
            }
            xo[0] = rintf((float)cnt / (float)p.n);  // torch.round: half to even
        } else {  // OneHotCategorical(logits [0, theta]): round(mean of n one-hot draws) per class
            const int C = a.dim;
            float m = 0.f;
            for (int c = 1; c < C; ++c) m = fmaxf(m, pr[c - 1]);
            float se = expf(-m);
            for (int c = 1; c < C; ++c) se += expf(pr[c - 1] - m);
            for (int c = 0; c < C; ++c) xo[c] = 0.f;  // counts, then rounded means
            for (int s = 0; s < p.n; ++s) {
                const uint64_t r = mc_mix(key + (uint64_t)s * 0xD1342543DE82EF95ull);
                const float u = mc_u01((uint32_t)r) * se;
                float cum = expf(-m);
                int k = 0;
                while (k + 1 < C && u >= cum) { cum += expf(pr[k] - m); ++k; }
                xo[k] += 1.f;
            }
            for (int c = 0; c < C; ++c) xo[c] = rintf(xo[c] / (float)p.n);
        }
    }
}

int launch_mc_impute(const AttrDesc* attrs_dev, int D, int learn_std, const float* P, int ldp, long long rows,
                     int n_samples, uint64_t seed, float* xhat, int ldx, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    McParams p{attrs_dev, P, xhat, rows, D, ldp, ldx, learn_std, n_samples, seed};
    const int blocks = (int)std::min<long long>((rows * D + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(mc_impute_kernel, dim3(blocks), dim3(256), 0, stream, p);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- standalone reparameterisation
// Gumbel-softmax over the K mixture components (samplers.py:28-30, encoders.py:87-88) and the
// Gaussian z = eps*std + mean (samplers.py:57-62).  One thread per row for the softmaxes.
__global__ void reparam_fwd_kernel(const float* logits, const float* u, int ldk, int K, float temp, float* s,
                                   float* probs, const float* mean, const float* sd, const float* eps, int ldz,
                                   int Z, float* z, int rows) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const size_t ok = (size_t)row * ldk, oz = (size_t)row * ldz;
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float l = logits[ok + k];
        m1 = fmaxf(m1, l);
        m2 = fmaxf(m2, (l + sv_gumbel(u[ok + k])) / temp);
    }
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const float l = logits[ok + k];
        s1 += expf(l - m1);
        s2 += expf((l + sv_gumbel(u[ok + k])) / temp - m2);
    }
    for (int k = 0; k < K; ++k) {
        const float l = logits[ok + k];
        probs[ok + k] = expf(l - m1) / s1;
        s[ok + k] = expf((l + sv_gumbel(u[ok + k])) / temp - m2) / s2;
    }
    for (int i = 0; i < Z; ++i) z[oz + i] = eps[oz + i] * sd[oz + i] + mean[oz + i];
}

// dlogits (through s only) = softmax-backward(ds)/temp ; dmean = dz ; dstd = dz*eps
__global__ void reparam_bwd_kernel(const float* s, const float* ds, int ldk, int K, float temp, float* dlogits,
                                   const float* eps, const float* dz, int ldz, int Z, float* dmean, float* dsd,
                                   int rows) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const size_t ok = (size_t)row * ldk, oz = (size_t)row * ldz;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += s[ok + k] * ds[ok + k];
    for (int k = 0; k < K; ++k) dlogits[ok + k] = s[ok + k] * (ds[ok + k] - dot) / temp;
    for (int i = 0; i < Z; ++i) {
        dmean[oz + i] = dz[oz + i];
        dsd[oz + i] = dz[oz + i] * eps[oz + i];
    }
}

int launch_reparam_fwd(const float* logits, const float* u, int ldk, int K, float temp, float* s, float* probs,
                       const float* mean, const float* sd, const float* eps, int ldz, int Z, float* z, int rows,
                       hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, logits, u, ldk, K,
                       temp, s, probs, mean, sd, eps, ldz, Z, z, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

int launch_reparam_bwd(const float* s, const float* ds, int ldk, int K, float temp, float* dlogits,
                       const float* eps, const float* dz, int ldz, int Z, float* dmean, float* dsd, int rows,
                       hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, s, ds, ldk, K, temp,
                       dlogits, eps, dz, ldz, Z, dmean, dsd, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- observed-row lists (mask-aware decoder)
// A missing attribute has no likelihood term (base_missvae.py:271-285: nll * mask), so nothing flows into its
// decoder network from that row: the decoder only has to run on the rows in which the attribute is observed.
// One workgroup per attribute compacts the chunk's row numbers in ascending order (ballot + popcount inside a
// wave, wave totals through LDS), which keeps every downstream summation order fixed by the data alone.
__global__ __launch_bounds__(1024) void observed_rows_kernel(const AttrDesc* __restrict__ attrs, const uint8_t* __restrict__ mask,
                                                             int ldm, int rows, int* __restrict__ idx, int ld_idx,
                                                             int* __restrict__ cnt, int* __restrict__ rank) {
    __shared__ int wtot[16];
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const AttrDesc a = attrs[d];
    const int col = a.xoff + a.dim - 1;
    int* out = idx + (size_t)d * ld_idx;
    int* rk = rank ? rank + (size_t)d * ld_idx : nullptr;  // the inverse: row -> its place in the list, or -1
    int run = 0;
    for (int base = 0; base < rows; base += 1024) {
        const int row = base + tid;
        const bool obs = row < rows && mask[(size_t)row * ldm + col] == 0;
        const unsigned long long b = __ballot(obs);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wid] = __popcll(b);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = wtot[w];
            woff += w < wid ? t : 0;
            tot += t;
        }
        if (obs) out[run + woff + before] = row;
        if (rk && row < rows) rk[row] = obs ? run + woff + before : -1;
        run += tot;
        __syncthreads();
    }
    if (tid == 0) cnt[d] = run;
}

int launch_observed_rows(const AttrDesc* attrs_dev, int D, const uint8_t* mask, int ldm, int rows, int* idx,
                         int ld_idx, int* cnt, hipStream_t stream, int* rank) {
    if (D <= 0) return SHIVAE_OK;
    SV_REQUIRE(rows >= 0 && rows <= ld_idx, "observed_rows: %d rows exceed the list stride %d", rows, ld_idx);
    hipLaunchKernelGGL(observed_rows_kernel, dim3(D), dim3(1024), 0, stream, attrs_dev, mask, ldm, rows, idx, ld_idx, cnt, rank);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- layer-0 input gradient: sum over attributes by gather
// The compact layer-0 backward leaves one product row per OBSERVED (row, attribute) pair: prod[i][d*blk .. +E) for the
// i-th observed row of attribute d.  The shared input's gradient is their sum over the attributes observed in a row
// (decoders.py:127 feeds every attribute's network the same input).  One wave per chunk row walks the attributes in
// ascending order -- 64 ranks per ballot, then one product row per set bit, two in flight -- and stores the sum once:
// no atomics, no zero fill, and a summation order fixed by the data alone.
template <int VEC>
__global__ __launch_bounds__(512) void l0_gather_kernel(const int* __restrict__ rank, int D, int ld_rank,
                                                        const float* __restrict__ prod, size_t ldp, int blk, int E,
                                                        float* __restrict__ out, int ldo, int rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 8 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int PER = 4;  // pieces per lane: E <= 64 * PER * VEC
    const int np = E / VEC;
    float acc[PER][VEC];
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;
    auto fetch = [&](const float* src, float (&t)[PER][VEC]) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = lane + 64 * j;
            if (q < np) {
                if constexpr (VEC == 4) {
                    const float4 x = reinterpret_cast<const float4*>(src)[q];
                    t[j][0] = x.x; t[j][1] = x.y; t[j][2] = x.z; t[j][3] = x.w;
                } else {
                    t[j][0] = src[q];
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) t[j][v] = 0.f;
            }
        }
    };
    auto add = [&](const float (&t)[PER][VEC]) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[j][v] += t[j][v];
    };
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int rk = d0 + lane < D ? rank[(size_t)(d0 + lane) * ld_rank + row] : -1;
        unsigned long long m = __ballot(rk >= 0);
        while (m) {
            const int b0 = __ffsll((long long)m) - 1;
            m &= m - 1;
            const bool two = m != 0;
            const int b1 = two ? __ffsll((long long)m) - 1 : b0;
            m &= m - 1;  // (0 & anything = 0 when there was no second bit)
            const int i0 = __shfl(rk, b0), i1 = __shfl(rk, b1);
            float t0[PER][VEC], t1[PER][VEC];
            fetch(prod + (size_t)i0 * ldp + (size_t)(d0 + b0) * blk, t0);
            if (two) fetch(prod + (size_t)i1 * ldp + (size_t)(d0 + b1) * blk, t1);
            add(t0);
            if (two) add(t1);
        }
    }
    float* o = out + (size_t)row * ldo;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = lane + 64 * j;
        if (q < np) {
            if constexpr (VEC == 4)
                reinterpret_cast<float4*>(o)[q] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            else
                o[q] = acc[j][0];
        }
    }
}

int launch_l0_gather(const int* rank, int D, int ld_rank, const float* prod, size_t ldp, int blk, int E, float* out,
                     int ldo, int rows, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    SV_REQUIRE(rows <= ld_rank && blk >= E, "l0_gather: %d rows exceed the rank stride %d, or block %d < E %d", rows, ld_rank, blk, E);
    const bool v4 = E % 4 == 0 && blk % 4 == 0 && ldp % 4 == 0 && ldo % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(prod) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    SV_REQUIRE(v4 ? E <= 1024 : E <= 256, "l0_gather: E = %d too wide", E);
    const dim3 grid(ceil_div(rows, 8)), block(512);
    if (v4)
        hipLaunchKernelGGL(l0_gather_kernel<4>, grid, block, 0, stream, rank, D, ld_rank, prod, ldp, blk, E, out, ldo, rows);
    else
        hipLaunchKernelGGL(l0_gather_kernel<1>, grid, block, 0, stream, rank, D, ld_rank, prod, ldp, blk, E, out, ldo, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// Zero the column blocks of the missing (row, attribute) pairs of Y[rows][D*blk] (float4 stores; blk % 4 == 0,
// 16-B aligned rows): the rows a ragged launch with a C row list does not store.  One workgroup per row: the first
// D threads look up their attribute's mask byte, the missing ones are listed in LDS, then all threads clear them.
__global__ __launch_bounds__(256) void zero_missing_blocks_kernel(const AttrDesc* __restrict__ attrs, int D,
                                                                  const uint8_t* __restrict__ mask, int ldm, float* __restrict__ Y,
                                                                  int ldy, int blk) {
    extern __shared__ int zm_list[];  // [D] missing attributes of this row, then the count
    __shared__ int zm_n;
    const int row = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) zm_n = 0;
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        const AttrDesc a = attrs[d];
        if (mask[(size_t)row * ldm + a.xoff + a.dim - 1] != 0) zm_list[atomicAdd(&zm_n, 1)] = d;
    }
    __syncthreads();
    const int n = zm_n, per = blk / 4;
    float4* yrow = reinterpret_cast<float4*>(Y + (size_t)row * ldy);
    for (int i = tid; i < n * per; i += 256) yrow[(size_t)zm_list[i / per] * per + i % per] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_zero_missing_blocks(const AttrDesc* attrs_dev, int D, const uint8_t* mask, int ldm, int rows, float* Y,
                               int ldy, int blk, hipStream_t stream) {
    if (D <= 0 || rows <= 0) return SHIVAE_OK;
    SV_REQUIRE(blk % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0, "zero_missing_blocks: 16-B pieces");
    hipLaunchKernelGGL(zero_missing_blocks_kernel, dim3(rows), dim3(256), (size_t)D * sizeof(int), stream, attrs_dev, D,
                       mask, ldm, Y, ldy, blk);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
