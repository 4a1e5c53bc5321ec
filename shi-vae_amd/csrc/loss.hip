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
                        dr[a.poff] = dmu * g;
                        if (p.learn_std) dr[a.poff + 1] = dsd * g * sv_softplus_grad_from_out(sd);
                    }
                } else if (a.type == 2) {  // bin: Bernoulli(logits)
                    float dl;
                    nll = sv_nll_bernoulli(xr[a.xoff], pr[a.poff], dl);
                    if (p.want_grad) dr[a.poff] = dl * g;
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
                            dr[a.poff + c - 1] = (sm - (c == idx ? 1.f : 0.f)) * g;
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

int launch_heter_nll(const AttrDesc* attrs_dev, int D, int learn_std, const float* x, int ldx,
                     const uint8_t* mask, int ldm, const float* P, int ldp, float* dP, int ldd, int rows,
                     float gs, float* rowloss, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    NllParams p;
    p.x = x; p.mask = mask; p.P = P; p.dP = dP; p.rowloss = rowloss; p.attrs = attrs_dev;
    p.ldx = ldx; p.ldm = ldm; p.ldp = ldp; p.ldd = ldd; p.rows = rows; p.D = D; p.learn_std = learn_std;
    p.ap = pow2_lanes(D); p.want_grad = dP != nullptr; p.gs = gs;
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

int launch_kl(const float* mu, const float* sd, const float* mup, const float* sdp, int ldz,
              const float* probs, int ldk, float* dmu, float* dsd, float* dmup, float* dsdp_pre, float* dlg,
              int rows, int Z, int K, float kz, float* rowloss, hipStream_t stream) {
    if (rows <= 0) return SHIVAE_OK;
    KlParams p;
    p.mu = mu; p.sd = sd; p.mup = mup; p.sdp = sdp; p.probs = probs;
    p.dmu = dmu; p.dsd = dsd; p.dmup = dmup; p.dsdp_pre = dsdp_pre; p.dlg = dlg; p.rowloss = rowloss;
    p.ldz = ldz; p.ldk = ldk; p.rows = rows; p.Z = Z; p.K = K; p.ap = pow2_lanes(std::max(Z, K));
    p.want_grad = dmu != nullptr; p.kz = kz;
    const int rows_per_block = 4 * (64 / p.ap);
    const int blocks = std::min(ceil_div(rows, rows_per_block), 256 * 8);
    hipLaunchKernelGGL(kl_kernel, dim3(blocks), dim3(256), 0, stream, p);
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

}  // namespace shivae
