// Scalar device math shared by the fused kernels and the standalone ones, so that the standalone
// kernels test exactly the arithmetic the fused scan uses.  Each function cites the reference line
// whose torch call it restates.
#pragma once
#include <hip/hip_runtime.h>

namespace shivae {

#define SV_F32_EPS 1.1920929e-07f          // torch.finfo(float32).eps (distributions/utils.py clamp_probs)
#define SV_HALF_LOG_2PI 0.91893853320467274f
#define SV_FIXED_STD 0.1f                   // decoders.py:22 fixed_std

// nn.Softplus(beta=1, threshold=20): encoders.py:68, shivae.py:61, decoders.py:74
__device__ __forceinline__ float sv_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }
// d softplus(a)/da from the OUTPUT: sigmoid(a) = 1 - exp(-softplus(a))
__device__ __forceinline__ float sv_softplus_grad_from_out(float sp) { return 1.f - expf(-sp); }
__device__ __forceinline__ float sv_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
// nn.LeakyReLU(0.01): encoders.py:57
__device__ __forceinline__ float sv_leaky(float v) { return v > 0.f ? v : 0.01f * v; }
// samplers.py:23-26: -log(-log(u + 1e-20) + 1e-20)
__device__ __forceinline__ float sv_gumbel(float u) { return -logf(-logf(u + 1e-20f) + 1e-20f); }

// Hardware-rate variants for the HBM-streaming likelihood kernel (v_exp_f32 / v_log_f32 / v_rcp_f32, each
// ~1 ulp; the exponent scaling adds <= |x| * 2^-24 relative): the libm forms cost 10-30 VALU
// instructions each and made that kernel instruction-bound at a third of the HBM rate.
__device__ __forceinline__ float sv_fexp(float v) { return __expf(v); }
__device__ __forceinline__ float sv_flog(float v) { return __logf(v); }
__device__ __forceinline__ float sv_frcp(float v) { return __frcp_rn(v); }
// log(1 + e), e >= 0, accurate for small e:  log(u) * e / (u - 1)  with u = fl(1 + e)
__device__ __forceinline__ float sv_flog1p_pos(float e) {
    const float u = 1.f + e, d = u - 1.f;
    return d == 0.f ? e : sv_flog(u) * e * sv_frcp(d);
}
__device__ __forceinline__ float sv_softplus_fast(float v) { return v > 20.f ? v : sv_flog1p_pos(sv_fexp(v)); }
__device__ __forceinline__ float sv_softplus_grad_from_out_fast(float sp) { return 1.f - sv_fexp(-sp); }
__device__ __forceinline__ float sv_nll_normal_fast(float x, float mu, float sd, float& dmu, float& dsd) {
    const float d = x - mu, inv = sv_frcp(sd), inv2 = inv * inv;
    dmu = -d * inv2;
    dsd = -d * d * inv2 * inv + inv;
    return d * d * 0.5f * inv2 + sv_flog(sd) + 0.91893853320467274f;
}
__device__ __forceinline__ float sv_nll_bernoulli_fast(float x, float l, float& dl) {
    const float e = sv_fexp(-fabsf(l));           // in (0, 1]
    const float r = sv_frcp(1.f + e);
    dl = (l >= 0.f ? r : e * r) - x;               // sigmoid(l) - x
    return fmaxf(l, 0.f) - l * x + sv_flog1p_pos(e);
}

// Normal(mu, sd).log_prob, negated (base_missvae.py:130,277)
__device__ __forceinline__ float sv_nll_normal(float x, float mu, float sd, float& dmu, float& dsd) {
    const float d = x - mu, inv = 1.f / sd, inv2 = inv * inv;
    dmu = -d * inv2;
    dsd = -d * d * inv2 * inv + inv;
    return d * d * 0.5f * inv2 + logf(sd) + SV_HALF_LOG_2PI;
}
// Bernoulli(logits=l).log_prob, negated = BCE-with-logits (base_missvae.py:135,277)
__device__ __forceinline__ float sv_nll_bernoulli(float x, float l, float& dl) {
    dl = sv_sigmoid(l) - x;
    return fmaxf(l, 0.f) - l * x + log1pf(expf(-fabsf(l)));
}

// KL(N(mq,sq) || N(mp,sp)) with the clamps of base_missvae.py:308-312 and torch's
// _kl_normal_normal; gradients wrt the UNclamped inputs (clamp passes gradient inside its range).
__device__ __forceinline__ float sv_kl_normal(float mq, float sq_in, float mp, float sp_in, float& dmq,
                                              float& dsq, float& dmp, float& dsp) {
    const float sq = fminf(fmaxf(sq_in, 1e-6f), 1e20f), sp = fminf(fmaxf(sp_in, 1e-6f), 1e20f);
    mq = fminf(fmaxf(mq, -1e20f), 1e20f);
    mp = fminf(fmaxf(mp, -1e20f), 1e20f);
    const float isp = 1.f / sp, ratio = sq * isp, r = ratio * ratio, d = (mq - mp) * isp, t1 = d * d;
    float kl = 0.5f * (r + t1 - 1.f - logf(r));
    if (kl != kl) kl = 0.f;  // base_missvae.py:317
    dmq = d * isp;
    dmp = -dmq;
    dsq = (sq_in >= 1e-6f && sq_in <= 1e20f) ? (sq * isp * isp - 1.f / sq) : 0.f;
    dsp = (sp_in >= 1e-6f && sp_in <= 1e20f) ? (-r * isp - t1 * isp + isp) : 0.f;
    return kl;
}

}  // namespace shivae
