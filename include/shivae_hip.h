/* shivae_hip.h -- C ABI of libshivae_hip.so: the Shi-VAE train step (ELBO forward + backward) as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (dbarrejon/Shi-VAE) has no FFI / plugin interface: its hot path is the Python
 * nn.Module API driven by Trainer.train_epoch (src/models/trainers.py:197-226).  This library sits
 * BELOW that API; each entry point cites the reference code it replaces.  The Python host layer in
 * shi-vae_amd/ (ShiVAE / Trainer with the reference's signatures) binds these with ctypes.
 *
 * Conventions
 *  - every pointer named *_dev / documented "device" is a HIP device pointer owned by the CALLER
 *    (torch allocates); the library allocates only small plan-time tables (freed in plan_destroy);
 *  - all arithmetic is fp32; masks are uint8 (1 = missing), as torch.bool in the reference;
 *  - tensors are time-major and contiguous: x[T][B][X], mask[T][B][X], u[T][B][K], eps[T][B][Z];
 *  - every call enqueues on the given hipStream_t and does not synchronise; one host thread per GPU;
 *  - return value: 0 = ok, <0 = error, text via shivae_last_error() (thread-local).
 */
#ifndef SHIVAE_HIP_H
#define SHIVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHIVAE_OK 0
#define SHIVAE_ERR_INVALID (-1)
#define SHIVAE_ERR_HIP (-2)
#define SHIVAE_ERR_UNBOUND (-3)

typedef struct shivae_plan shivae_plan;
typedef void* shivae_stream; /* hipStream_t */

/* attribute types, lib/utils.py:200-212 / data_types.csv "type" column */
enum { SHIVAE_REAL = 0, SHIVAE_POS = 1, SHIVAE_BIN = 2, SHIVAE_CAT = 3 };

/* Model + sizing description.  Mirrors ShiVAE.__init__(h_dim, z_dim, s_dim, n_layers=1, types_list,
 * learn_std) (src/models/shivae.py:25-70). */
typedef struct {
    int T;               /* time steps per call (exact) */
    int B;               /* sequences per call (exact; callers split larger batches) */
    int H, Z, K;         /* h_dim, z_dim, s_dim */
    int n_attr;          /* len(types_list) */
    const int* attr_type;   /* [n_attr] SHIVAE_REAL.. */
    const int* attr_nclass; /* [n_attr] one-hot width for SHIVAE_CAT (>=2), ignored otherwise */
    int learn_std;       /* decoders.py:71-86 */
    int chunk_rows;      /* rows (t,b pairs) of decoder activations kept at once; 0 = auto */
} shivae_config;

/* parameter tensors of the flat fp32 parameter buffer (state_dict key in the comment) */
enum {
    SHIVAE_P_PHIX0_W = 0, SHIVAE_P_PHIX0_B, /* encoder.phi_x.0   [H,X]      encoders.py:47-48 */
    SHIVAE_P_PHIX2_W, SHIVAE_P_PHIX2_B,     /* encoder.phi_x.2   [H,H]      encoders.py:50 */
    SHIVAE_P_XS0_W, SHIVAE_P_XS0_B,         /* encoder.phi_x_s.0 [H,X+H]    encoders.py:54 */
    SHIVAE_P_XS2_W, SHIVAE_P_XS2_B,         /* encoder.phi_x_s.2 [K,H]      encoders.py:56 */
    SHIVAE_P_E0_W, SHIVAE_P_E0_B,           /* encoder.enc.0     [H,2H+K]   encoders.py:61 */
    SHIVAE_P_E2_W, SHIVAE_P_E2_B,           /* encoder.enc.2     [H,H]      encoders.py:63 */
    SHIVAE_P_ENCMEAN_W, SHIVAE_P_ENCMEAN_B, /* encoder.enc_mean  [Z,H]      encoders.py:65 */
    SHIVAE_P_ENCSTD_W, SHIVAE_P_ENCSTD_B,   /* encoder.enc_std.0 [Z,H]      encoders.py:67 */
    SHIVAE_P_PHIZ_W, SHIVAE_P_PHIZ_B,       /* phi_z.0           [H,Z]      shivae.py:52 */
    SHIVAE_P_PRIOR0_W, SHIVAE_P_PRIOR0_B,   /* prior.0           [H,H+K]    shivae.py:57 */
    SHIVAE_P_PRIORMEAN_W, SHIVAE_P_PRIORMEAN_B, /* prior_mean    [Z,H]      shivae.py:59 */
    SHIVAE_P_PRIORSTD_W, SHIVAE_P_PRIORSTD_B,   /* prior_std.0   [Z,H]      shivae.py:60 */
    SHIVAE_P_RNN_WIH, SHIVAE_P_RNN_WHH,     /* rnn.weight_{ih,hh}_l0 [4H,H] shivae.py:68 */
    SHIVAE_P_RNN_BIH, SHIVAE_P_RNN_BHH,     /* rnn.bias_{ih,hh}_l0   [4H] */
    /* per attribute (attr = index into types_list), decoders.py:60-66 */
    SHIVAE_P_DEC0_W, SHIVAE_P_DEC0_B,       /* dec.decoder_dict.<v>.dec.0 [2H,2H+K] */
    SHIVAE_P_DEC2_W, SHIVAE_P_DEC2_B,       /* ...dec.2 [H,2H] */
    SHIVAE_P_DEC4_W, SHIVAE_P_DEC4_B,       /* ...dec.4 [H,H] */
    SHIVAE_P_HEAD0_W, SHIVAE_P_HEAD0_B,     /* bin: dec_p.0 / cat: dec_p_cat.0 [H,H]  decoders.py:89-102 */
    SHIVAE_P_HEADMEAN_W, SHIVAE_P_HEADMEAN_B, /* real/pos: dec_mean [1,H]             decoders.py:70,79 */
    SHIVAE_P_HEADSTD_W, SHIVAE_P_HEADSTD_B,   /* real/pos, learn_std: dec_std.0 [1,H] decoders.py:72,82 */
    SHIVAE_P_HEADOUT_W, SHIVAE_P_HEADOUT_B, /* bin: dec_p.2 [1,H] / cat: dec_p_cat.2 [C-1,H] */
    SHIVAE_P_COUNT
};

/* named per-(t,b) outputs of the last shivae_elbo_step, copied out UNPADDED as [T][B][width] */
enum {
    SHIVAE_O_Z = 0,      /* z_t sample        [Z]  (shivae.py:146 z_tup[0]) */
    SHIVAE_O_Z_MEAN,     /*                   [Z]  z_tup[1] */
    SHIVAE_O_Z_STD,      /*                   [Z]  z_tup[2] */
    SHIVAE_O_ZP_MEAN,    /* prior mean        [Z]  z_prior_tup[0] */
    SHIVAE_O_ZP_STD,     /* prior std         [Z]  z_prior_tup[1] */
    SHIVAE_O_S,          /* s_t (soft sample) [K]  s_tup[0] */
    SHIVAE_O_S_PROBS,    /*                   [K]  s_tup[1] */
    SHIVAE_O_DEC_PARAMS, /* raw decoder head outputs [P] (needs SHIVAE_STEP_KEEP_PARAMS); layout per
                            attribute in types_list order: real/pos -> mean[, pre-softplus std if
                            learn_std]; bin -> logit; cat -> theta_1..theta_{C-1} (theta_0 == 0) */
    SHIVAE_O_ROWLOSS,    /* per-timestep ELBO terms [6]: nll_real,nll_pos,nll_bin,nll_cat,kl_z,kl_s */
    SHIVAE_O_H_PREV,     /* h_{t-1}           [H] */
    SHIVAE_O_PHI_Z,      /*                   [H] */
    /* records of the recurrent scans (what autograd would keep for shivae.py:102-143), for kernel-level parity */
    SHIVAE_O_A_S,        /* relu(phi_x_s.0 [x,h])           [H]  encoders.py:83 */
    SHIVAE_O_LOGITS,     /* LeakyReLU(phi_x_s.2 a_s)        [K]  encoders.py:56-57 */
    SHIVAE_O_E1,         /* relu(enc.0 [phi_x,s,h])         [H]  encoders.py:91 */
    SHIVAE_O_E2,         /* relu(enc.2 e1)                  [H] */
    SHIVAE_O_GATES,      /* LSTM gates i,f,g,o after their nonlinearity [4H]  shivae.py:136 */
    SHIVAE_O_C_NEW,      /* c_t                             [H] */
    SHIVAE_O_D_GATES,    /* after shivae_scan_bwd / a training step: d loss / d gate pre-activations [4H] */
    SHIVAE_O_D_PHIZ_PRE, /* d / d (phi_z.0 z + b)           [H] */
    SHIVAE_O_D_MU,       /* d / d enc_mean output           [Z] */
    SHIVAE_O_D_STD_PRE,  /* d / d (enc_std.0 e2 + b)        [Z] */
    SHIVAE_O_D_E2_PRE,   /* d / d (enc.2 e1 + b)            [H] */
    SHIVAE_O_D_E1_PRE,   /* d / d (enc.0 [phi_x,s,h] + b)   [H] */
    SHIVAE_O_D_LOGITS_PRE, /* d / d (phi_x_s.2 a_s + b)     [K] */
    SHIVAE_O_D_AS_PRE,   /* d / d (phi_x_s.0 [x,h] + b)     [H] */
    SHIVAE_O_COUNT
};

#define SHIVAE_STEP_FORWARD_ONLY 1 /* no gradients (validation / imputation) */
#define SHIVAE_STEP_KEEP_PARAMS 2  /* keep all decoder parameters for SHIVAE_O_DEC_PARAMS */

const char* shivae_last_error(void);
/* build id string, e.g. "shivae_hip gfx950 <date>" */
const char* shivae_version(void);
/* Per-device set-up for the CURRENT device (today: the device address of the scratch array of shivae_sqnorm_noalloc).
 * shivae_plan_bind calls it too; call it yourself, outside any stream capture, on a device that never binds a plan.
 * No counterpart in the reference. */
int shivae_init(void);

/* ---- plan: model description, parameter layout, workspace ---------------------------------- */
/* replaces ShiVAE.__init__ bookkeeping (shivae.py:25-70, utils.py:229-241,291-315) */
int shivae_plan_create(const shivae_config* cfg, shivae_plan** out);
void shivae_plan_destroy(shivae_plan* plan);

/* total floats of the flat parameter (and gradient) buffer */
int64_t shivae_plan_num_params(const shivae_plan* plan);
/* float offset / shape of one parameter tensor inside the flat buffer; attr is ignored for the
 * non-decoder tensors.  Returns SHIVAE_ERR_INVALID if the tensor does not exist for that attribute. */
int shivae_plan_param_info(const shivae_plan* plan, int tensor_id, int attr, int64_t* offset, int* rows,
                           int* cols);
/* x_dim (sum of dims, utils.py:229-241) and P (decoder parameter columns per row) */
int shivae_plan_x_dim(const shivae_plan* plan);
int shivae_plan_p_dim(const shivae_plan* plan);
int shivae_plan_chunk_rows(const shivae_plan* plan);
/* Observed fraction (1 - missing rate of base_missvae.py:271-285's mask) of the first row chunk this plan trained
 * on, which picks the form of the decoder's layer-0 input gradient (SHIVAE_DEMB); < 0 before the
 * first training step or when SHIVAE_DEMB forces a form.  No counterpart in the reference. */
float shivae_plan_observed_fraction(const shivae_plan* plan);
/* Set that fraction instead of having the first training chunk measured (which costs one host wait inside that step
 * and cannot happen under stream capture): f in [0,1] picks the form (compact below 0.6), f < 0 asks for a new
 * measurement.  shivae_plan_bind resets it to "not measured".  With the compact form the layer-0 input gradient is
 * summed with fp32 atomics shared by the attributes: its summation order, hence its last bits, vary run to run. */
int shivae_plan_set_observed_fraction(shivae_plan* plan, float f);

/* Row capacity of the decoder's activation buffers, separately from the row chunk (default: equal).  A mask-aware
 * training chunk (base_missvae.py:271-285: only observed entries have likelihood terms) keeps each attribute's
 * activations for its observed rows only, so with most entries missing a chunk of chunk_rows rows fills a small part of
 * buffers sized for chunk_rows dense rows -- and those buffers are nearly all of the chunk's memory.  rows < chunk_rows
 * sizes them for `rows` rows per attribute: the same memory then carries a chunk several times larger (fewer, larger
 * launches).  A training chunk in which some attribute is observed in more than `rows` rows is detected on the host
 * before its launches go out (one host wait per chunk, only on plans with rows < chunk_rows) and run in pieces of
 * `rows` rows instead (shivae_plan_overflow_chunks counts them); dense passes (forward only, imputation, generation)
 * step `rows` rows at a time; such a plan always takes the compact form of the layer-0 input gradient.  Call before
 * shivae_plan_bind (it changes shivae_plan_workspace_bytes); rows <= 0 restores the default.  No counterpart in the
 * reference. */
int shivae_plan_set_activation_rows(shivae_plan* plan, int rows);
int shivae_plan_activation_rows(const shivae_plan* plan);
long long shivae_plan_overflow_chunks(const shivae_plan* plan);

size_t shivae_plan_workspace_bytes(const shivae_plan* plan);
/* attach caller-owned device memory: workspace (>= workspace_bytes, 256-B aligned; it is zeroed
 * here, synchronously), the flat parameter buffer and the flat gradient buffer */
int shivae_plan_bind(shivae_plan* plan, void* workspace_dev, size_t workspace_bytes, float* params_dev,
                     float* grads_dev);

/* ---- the step ------------------------------------------------------------------------------ */
/* replaces model(x, mask, beta, temp) + n_elbo.backward() (trainers.py:212-215,
 * base_missvae.py:144-231, shivae.py:72-154) for one contiguous batch of exactly cfg.B sequences.
 *   u, eps    : explicit noise (what samplers.py:25 torch.rand and :60 normal_() would draw), [T][B][K] and
 *               [T][B][Z]; or BOTH NULL: the scans draw it themselves (north_star kernel 2: "on-device Philox"),
 *               see shivae_plan_set_noise_seed -- nothing is materialised, forward and reverse scan regenerate the
 *               same numbers
 *   inv_batch : 1/B_total -- the batch mean of base_missvae.py:212; with micro-batches or data
 *               parallelism pass the GLOBAL batch so that gradients add up across calls/ranks
 *   losses_dev: 9 floats, ACCUMULATED (+=): n_elbo, nll_loss, kld, kld_q_z, kld_q_s, nll_real,
 *               nll_pos, nll_bin, nll_cat (base_missvae.py:215-230); caller zeroes them
 * Gradients are ACCUMULATED into the bound gradient buffer (zero it with shivae_zero). */
int shivae_elbo_step(shivae_plan* plan, const float* x_dev, const uint8_t* mask_dev, const float* u_dev,
                     const float* eps_dev, float beta, float temp, float inv_batch, int flags,
                     float* losses_dev, shivae_stream stream);

/* ---- the phases of the step (SURVEY.md 8b minimum set), for callers that compose their own ------------------
 * shivae_elbo_step == scan_fwd; prior_kl; decoder_nll; [scan_bwd;] loss_reduce on one bound plan and stream.
 * All of them work on the plan's workspace records; gradients are ACCUMULATED into the bound gradient buffer.
 *
 * shivae_scan_fwd -- north_star kernel (1): ShiVAE.feed2model's time loop without the prior and decoder
 *   (src/models/shivae.py:102-143; GMEncoder.forward src/models/encoders.py:70-98; Gumbel-softmax and Gaussian
 *   reparameterisation fused, src/models/samplers.py:23-62; nn.LSTM single step shivae.py:136): the x-only encoder
 *   GEMMs over all rows, then ONE persistent kernel looping over t.  Afterwards shivae_get_output gives Z, Z_MEAN,
 *   Z_STD, S, S_PROBS, PHI_Z, H_PREV and the records A_S, LOGITS, E1, E2, GATES, C_NEW. */
int shivae_scan_fwd(shivae_plan* plan, const float* x_dev, const float* u_dev, const float* eps_dev, float temp,
                    shivae_stream stream);
/* prior network + KL(q(z)||p(z)) + KL(q(s)||U) over all rows (shivae.py:124-127, base_missvae.py:287-343), rowloss
 * columns 4,5; without SHIVAE_STEP_FORWARD_ONLY also their gradients (scaled by beta*inv_batch), the prior network's
 * backward, and the START of this step's d_emb records (zero + the prior network's input gradient). */
int shivae_prior_kl(shivae_plan* plan, float beta, float inv_batch, int flags, shivae_stream stream);
/* HeterDecoder.forward over all rows (src/models/decoders.py:109-176) + heter_nll (base_missvae.py:233-285) in row
 * chunks, rowloss columns 0..3; without SHIVAE_STEP_FORWARD_ONLY each chunk's backward runs right behind its
 * forward (decoder activations never exist for more than one chunk, so forward and backward are ONE call): weight
 * gradients are accumulated, the input gradient is ADDED to the d_emb records (call shivae_prior_kl first, which
 * zeroes them). */
int shivae_decoder_nll(shivae_plan* plan, const float* x_dev, const uint8_t* mask_dev, float inv_batch, int flags,
                       shivae_stream stream);
/* north_star kernel (4): the reverse-time scan (what autograd does for shivae.py:102-143 -- full BPTT, h and c are
 * never detached) + the weight gradients of the recurrent cell and the phi_x backward, after shivae_scan_fwd on the
 * same plan, x, u, eps, temp.  The gradients that reach the chain from outside are either the records accumulated
 * by shivae_prior_kl / shivae_decoder_nll (pass NULL) or given here, unpadded:
 *   d_emb_dev   [T][B][2H+K]  d loss / d cat[phi_z_t, s_t, h_{t-1}]  (decoders.py:127 order)
 *   dmu_dev, dstd_dev [T][B][Z]  d loss / d z_mean, d z_std          (the KL term's part)
 *   dlogits_dev [T][B][K]     d loss / d s logits through s_probs
 * Afterwards shivae_get_output gives the D_* records. */
int shivae_scan_bwd(shivae_plan* plan, const float* x_dev, const float* u_dev, const float* eps_dev, float temp,
                    const float* d_emb_dev, const float* dmu_dev, const float* dstd_dev, const float* dlogits_dev,
                    shivae_stream stream);
/* base_missvae.py:211-230: per-sequence sums of the rowloss records over t, then the 9 batch means, ACCUMULATED
 * into losses_dev (as shivae_elbo_step) */
int shivae_loss_reduce(shivae_plan* plan, float beta, float inv_batch, float* losses_dev, shivae_stream stream);

/* ShiVAE.sample_from_prior (src/models/shivae.py:156-214): generation from the prior over plan.T steps for
 * plan.B sequences with explicit noise u[T][B][K] (Gumbel uniforms) and eps[T][B][Z] (or both NULL: in-kernel
 * noise, below).  Afterwards
 * shivae_get_output gives Z (the prior sample), Z_MEAN / Z_STD (prior mean / std), S, S_PROBS (the uniform
 * 1/K) and DEC_PARAMS. */
int shivae_sample_prior(shivae_plan* plan, const float* u_dev, const float* eps_dev, float temp, shivae_stream stream);

/* ---- in-kernel noise (samplers.py:23-30 torch.rand, samplers.py:57-62 normal_()) ---------------------------------
 * Every entry point above that takes (u_dev, eps_dev) accepts NULL for both; the scans then compute
 *   u[t][b][k], eps[t][b][n] = f(seed, row = t*B + b, index)
 * with Philox4x32-10 (key = seed; counter = (row, index / 4, stream); uniforms ((x >> 9) + 0.5) 2^-23 in (0,1), normals
 * by Box-Muller) where they need them: no noise tensor exists.  The seed is the plan's; set a NEW one before every
 * step (same seed = same noise), and leave it alone between shivae_scan_fwd and shivae_scan_bwd of one step.  Rows
 * are the plan's rows: two micro-batches or two ranks that must not share noise take different seeds. */
int shivae_plan_set_noise_seed(shivae_plan* plan, uint64_t seed);
/* the same streams written out: u_dev[rows][K], eps_dev[rows][Z] (either may be NULL).  A step with these tensors
 * equals the step with (NULL, NULL) under the same seed bit for bit. */
int shivae_noise_fill(uint64_t seed, int64_t rows, int K, int Z, float* u_dev, float* eps_dev, shivae_stream stream);

/* copy one named output of the last step into dst_dev as [T][B][width] (unpadded) */
int shivae_get_output(shivae_plan* plan, int which, float* dst_dev, shivae_stream stream);
int shivae_output_width(const shivae_plan* plan, int which);
/* per-sequence sums over time of the last step: [B][6] (order as SHIVAE_O_ROWLOSS) */
int shivae_get_persum(shivae_plan* plan, float* dst_dev, shivae_stream stream);

/* ---- individual kernels (exported for parity tests and for callers that compose their own step) */
/* base_missvae.py:233-285 heter_nll (eval_obs=False) forward, fused with d nll/d params.
 * params: raw head outputs [rows][ldp] in the SHIVAE_O_DEC_PARAMS layout; dparams may be NULL
 * (forward only) or alias params; rowloss: [rows][6], columns 0..3 written. */
int shivae_heter_nll(shivae_plan* plan, const float* x_dev, const uint8_t* mask_dev, const float* params_dev,
                     int ldp, float* dparams_dev, int rows, float grad_scale, float* rowloss_dev,
                     shivae_stream stream);
/* base_missvae.py:378-404 generate_heter_samples: xhat[rows][x_dim] = per attribute the mean of n_samples
 * draws from its likelihood (real/pos Normal, bin Bernoulli, cat one-hot), rounded for bin/cat.  Draws come
 * from a counter-based generator keyed by (seed, row, attribute, sample). */
int shivae_mc_impute(shivae_plan* plan, const float* params_dev, int ldp, int rows, int n_samples, uint64_t seed,
                     float* xhat_dev, shivae_stream stream);
/* base_missvae.py:287-343 kl_q_z + kl_q_s forward (rowloss columns 4,5), fused with their gradients
 * scaled by kl_scale (= beta*grad_scale): d/dmu_q, d/dstd_q, d/dmu_p, d/d(pre-softplus std_p) and
 * d/d(s logits) through s_probs = softmax(logits).  Gradient pointers may all be NULL. */
int shivae_kl_terms(const float* mu_q, const float* std_q, const float* mu_p, const float* std_p, int ldz,
                    const float* s_probs, int ldk, int rows, int Z, int K, float kl_scale, float* dmu_q,
                    float* dstd_q, float* dmu_p, float* dstd_p_pre, float* dlogits, float* rowloss_dev,
                    shivae_stream stream);
/* samplers.py:23-62: s_probs = softmax(logits), s = softmax((logits+gumbel(u))/temp), z = eps*std+mean */
int shivae_reparam_fwd(const float* logits, const float* u, int ldk, int K, float temp, float* s, float* s_probs,
                       const float* mean, const float* std, const float* eps, int ldz, int Z, float* z, int rows,
                       shivae_stream stream);
int shivae_reparam_bwd(const float* s, const float* ds, int ldk, int K, float temp, float* dlogits,
                       const float* eps, const float* dz, int ldz, int Z, float* dmean, float* dstd, int rows,
                       shivae_stream stream);
/* C = epilogue(op(A) op(B)) on the fp32 matrix cores; layout 0: A[M,K] B[N,K]^T (nn.Linear forward),
 * 1: A[M,K] B[K,N] (input gradient), 2: A[K,M]^T B[K,N] (weight gradient).  epi bits: 1 bias,
 * 2 relu, 4 softplus, 8 += C, 16 gate by aux>0, 32 atomic accumulate (alone; required if splitk>1) */
int shivae_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                int ldc, const float* bias, const float* aux, int ldaux, int epi, int splitk,
                shivae_stream stream);

/* Arithmetic of the 128x128-tile GEMMs (the decoder: 99 % of the step's flops).  0 (default): fp32-input MFMA
 * (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain).  1: every fp32 operand is split EXACTLY into three bf16 terms and
 * a product is accumulated in fp32 from the six bf16 MFMA products that matter (v_mfma_f32_32x32x16_bf16; the
 * dropped cross terms are <= 2^-26 of the product): same inputs, same outputs, fp32-level error, 2-3x the rate.
 * Also selectable with SHIVAE_GEMM_SPLIT=1.  No counterpart in the reference. */
int shivae_set_gemm_split(int on);
/* Split-mode arithmetic in full (tests/test_gpu_split_mode.py holds it against the fp32-input MFMA on adversarial
 * operands, both against fp64): x = hi + mid + lo exactly, each term the NEAREST bf16 of what is left; a*b is
 * accumulated as ah*bh in one fp32 accumulator and (ah*bm + am*bh + am*bm + ah*bl + al*bh) in a second one, added at
 * the end; the dropped terms are <= 2^-26 |a b|.  Measured error <= the fp32-input MFMA's on the same operands
 * (K up to 131 072, magnitudes spread over 2^+-40 in one row, cancellation of 12 bits, operands down to 2^-100).
 * Outside the contract: operands that are fp32 DENORMALS (|x| < 2^-126) -- the bf16 matrix cores flush denormal
 * inputs, so they count as zero or as their leading term only; and +-inf operands, which give NaN (inf - hi(inf))
 * where the fp32-input MFMA gives +-inf or NaN.  NaN propagates as NaN in both modes; every output that touches no
 * non-finite operand is unaffected.
 *
 * The decoder's weight operands are split ONCE per step into a bf16 image that the 128x256-tile kernel of
 * csrc/gemm_split.hip streams into LDS by DMA; exported for tests and for callers that compose their own step:
 *   shivae_presplit_bytes   -- bytes of the image of an [N,K] weight matrix
 *   shivae_presplit_weights -- W[N][ldw] (fp32, K used) -> image
 *   shivae_gemm_nt_presplit -- C[M,N] = epilogue(A[M,K] W[N,K]^T) with W given as its image; N % 256 == 0, K % 4 == 0,
 *                              K >= 32, M > 64, 16-B aligned operands; epi as shivae_gemm (no softplus). */
size_t shivae_presplit_bytes(int N, int K);
int shivae_presplit_weights(const float* W_dev, int N, int K, int ldw, void* image_dev, shivae_stream stream);
int shivae_gemm_nt_presplit(int M, int N, int K, const float* A, int lda, const void* image_dev, float* C, int ldc,
                            const float* bias, const float* aux, int ldaux, int epi, int splitk, shivae_stream stream);

/* ---- optimiser ------------------------------------------------------------------------------ */
int shivae_zero(float* buf_dev, int64_t n, shivae_stream stream);
/* sum of squares of a flat buffer -> *out_dev (overwritten); scratch_dev: >= 1024 floats */
int shivae_sqnorm(const float* buf_dev, int64_t n, float* scratch_dev, float* out_dev, shivae_stream stream);
/* shivae_sqnorm for callers that cannot allocate or pass a scratch buffer (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The block partials live in
 * a 4 KiB file-scope device array that every GPU the library runs on holds until the process ends; the entry point
 * calls no hipMalloc / hipFree.  n is limited to SHIVAE_SQNORM_NOALLOC_MAX elements (larger n is rejected).  All
 * calls on one GPU share that array: calls that overlap on one GPU (two streams) race -- serialise them.
 * The device of `stream` must be the current device (checked).  The array's device address is resolved by
 * shivae_init / shivae_plan_bind; a first use on a device that saw neither resolves it on the spot, which is refused
 * with an error when that first use is inside a stream capture. */
#define SHIVAE_SQNORM_NOALLOC_MAX ((int64_t)1 << 32)
int shivae_sqnorm_noalloc(const float* buf_dev, int64_t n, float* out_dev, shivae_stream stream);
/* torch.nn.utils.clip_grad_norm_(params, clip) followed by torch.optim.Adam.step()
 * (trainers.py:222-224) on flat buffers; sqnorm_dev holds the squared global gradient norm
 * (after any all-reduce).  step is the 1-based Adam step count. */
int shivae_clip_adam(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                     int64_t n, const float* sqnorm_dev, float clip, float lr, float beta1, float beta2,
                     float eps, int step, shivae_stream stream);

/* ---- data parallel (no counterpart in the reference, which is single-process) ----------------
 * Sum the flat gradient buffer over the ranks of an RCCL communicator, in place, in buckets of bucket_floats
 * (<= 0: one message), enqueued on `stream`: call it after the step on the step's stream, BEFORE
 * shivae_sqnorm / shivae_clip_adam (the global norm of trainers.py:222 is the norm of the summed gradient; every
 * rank ran its shard with inv_batch = 1/B_global).  rccl_comm is an ncclComm_t: the caller's own, or one made by
 * shivae_comm_init from the 128-byte id that rank 0 got from shivae_comm_unique_id and sent to the other ranks.
 * librccl is opened at the first of these calls (dlopen), not at load time. */
int shivae_comm_unique_id(void* id128_out);
int shivae_comm_init(void** rccl_comm_out, int world_size, int rank, const void* id128);
int shivae_comm_destroy(void* rccl_comm);
int shivae_allreduce_grads(float* grads_dev, int64_t n, void* rccl_comm, int64_t bucket_floats, shivae_stream stream);

/* ---- pre/post pass either side of the step ---------------------------- */
/* One attribute of a fitted HeterogeneousScaler (lib/scalers.py:27-58): type, first one-hot/x column
 * (utils.get_idx_transform, lib/utils.py:291-315), x columns (1, or nclass for cat), raw column
 * ('index'), and the fitted moments (StandardScaler / PositiveScaler .mean/.std, scalers.py:223-289;
 * unused for bin/cat). */
typedef struct {
    int type, xoff, dim, index;
    double mean, std;
} shivae_scale_attr;
enum { SHIVAE_SCALE_TRANSPOSE = 1, SHIVAE_SCALE_ZERO_FILL = 2, SHIVAE_SCALE_F64 = 4 };
/* HeterogeneousScaler.transform (scalers.py:85-131) fused with Trainer.preprocess_batch
 * (models/train.py:333-362) and utils.zero_fill (lib/utils.py:89-100): raw data[n][t][n_attr], f32 or
 * (flag F64, what the reference's numpy datasets hold) f64, scaled in f64 and stored as f32 like the
 * reference (NaN allowed) + mask[n][t][n_attr] u8 (NULL: mask = isnan(data)) -> x[t][n][x_dim] (TRANSPOSE) or
 * x[n][t][x_dim], mask replicated over one-hot columns; ZERO_FILL writes 0 at masked entries.
 * col2attr_dev[x_dim]: attribute of every x column.  Categories are the integers 0..nclass-1. */
int shivae_scale_transform(const void* data_dev, const uint8_t* mask_dev, int n, int t, int n_attr, int x_dim,
                           const shivae_scale_attr* attrs_dev, const int* col2attr_dev, float* x_dev,
                           uint8_t* mask_out_dev, int flags, shivae_stream stream);
/* HeterogeneousScaler.inverse_transform (scalers.py:133-183): x (layout as above) -> data[n][t][n_attr]
 * (f32, or f64 with flag F64);
 * mask_dev (x layout, may be NULL) -> mask_out_dev[n][t][n_attr] (may be NULL). */
int shivae_scale_inverse(const float* x_dev, const uint8_t* mask_dev, int n, int t, int n_attr, int x_dim,
                         const shivae_scale_attr* attrs_dev, void* data_dev, uint8_t* mask_out_dev, int flags,
                         shivae_stream stream);

/* ---- measurement support (no counterpart in the reference) ---------------------------------- */
/* Bracket every kernel launch of the following calls with HIP events on the launch stream.
 * shivae_profile_stop synchronises the device and writes up to max_entries records of 4 doubles
 * {kernel id, launches, total ms, total work} and returns the record count.  ids 0..17 are the
 * GEMM kernel variants (layout*6 + tile; work = flops), 18/19 the recurrent scan fwd/bwd (flops),
 * 20 the likelihood kernel, 21 the KL kernel (work = algorithmic bytes).  Not thread-safe. */
int shivae_profile_start(void);
/* 1 between shivae_profile_start and shivae_profile_stop (callers that capture HIP graphs must not capture then) */
int shivae_profile_active(void);
/* Diagnostics: while buf_dev is non-NULL every 128x128 GEMM workgroup writes 4 shader-clock stamps {start,
 * first K-tile staged, main loop done, epilogue done} to buf_dev[4*workgroup]. */
int shivae_debug_gemm_stamps(unsigned long long* buf_dev);
int shivae_profile_stop(double* out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* SHIVAE_HIP_H */
