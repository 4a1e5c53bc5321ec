// Shared declarations for the Shi-VAE HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shivae_hip.h"

namespace shivae {

void set_error(const char* fmt, ...);

#define SV_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            shivae::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,              \
                              hipGetErrorString(e__));                                  \
            return SHIVAE_ERR_HIP;                                                      \
        }                                                                               \
    } while (0)

#define SV_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            shivae::set_error(__VA_ARGS__);                                             \
            return SHIVAE_ERR_INVALID;                                                  \
        }                                                                               \
    } while (0)

#define SV_TRY(expr)                                                                    \
    do {                                                                                \
        int r__ = (expr);                                                               \
        if (r__ != SHIVAE_OK) return r__;                                               \
    } while (0)

// ---------------------------------------------------------------- per-kernel HIP-event profiler (plan.hip)
// ids 0..17: gemm_kernel variants = layout*6 + tile (0: 256x32, 1: 32x256, 2: 256x64, 3: 64x256, 4: 128x128, 5: 128x128 split-bf16)
enum ProfId { PROF_SCAN_FWD = 18, PROF_SCAN_BWD = 19, PROF_NLL = 20, PROF_KL = 21, PROF_HEADS_FWD = 22, PROF_HEADS_BWD = 23, PROF_SPLITW_NT = 24, PROF_SPLIT_TN = 25, PROF_COUNT = 26 };
void prof_begin(int id, double work, hipStream_t s);  // work: flops (gemm, scan) or algorithmic bytes (nll, kl)
void prof_end(hipStream_t s);
void prof_shape(int M, int N, int K, int G);  // optional detail for the next prof_begin

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------- GEMM (gemm.hip)
// One problem of a grouped GEMM: all groups share M,N,K and leading dimensions.
struct GemmGroup {
    const float* A;
    const float* B;
    float* C;
    const float* bias;  // [N] or null
    const float* aux;   // [M,N] activation whose sign gates the output (ReLU backward) or null
    float* csum;        // [N] column sums of the stored C are ADDED here (EPI_COLSUM: bias gradient) or null
    // Ragged rows (GemmDesc::ragged selects which of these a launch honours; all default to null):
    const int* cnt;     // device int: this group's live row count -- M for NT / NN, K for TN (RAG_CNT)
    const int* aidx;    // A is read through a row list: row i (NT/NN) or k-row i (TN) comes from A row aidx[i] (RAG_AIDX)
    const int* bidx;    // TN only: k-row i of B comes from B row bidx[i] (RAG_BIDX)
    const int* cidx;    // NT / NN: row i of the result is stored to C row cidx[i]; aux stays row i (RAG_CIDX)
};
enum GemmRagged { RAG_CNT = 1, RAG_AIDX = 2, RAG_BIDX = 4, RAG_CIDX = 8 };

enum GemmLayout {
    // C[M,N] = A[M,K] * B[N,K]^T   (forward linear: activations x weight[out,in])
    GEMM_NT = 0,
    // C[M,N] = A[M,K] * B[K,N]     (input gradient: dY x weight[out,in])
    GEMM_NN = 1,
    // C[M,N] = A[K,M]^T * B[K,N]   (weight gradient: dY^T x activations, K = rows)
    GEMM_TN = 2,
};

enum GemmEpi {
    EPI_BIAS = 1,      // + bias[n]
    EPI_RELU = 2,      // max(v,0)
    EPI_SOFTPLUS = 4,  // softplus, threshold 20
    EPI_ADDC = 8,      // + C_old (before the activation)
    EPI_MASK = 16,     // * (aux[m,n] > 0)   (after everything else)
    EPI_ATOMIC = 32,   // atomicAdd into C (split-K / cross-launch accumulation); excludes the others
    EPI_COLSUM = 64,   // csum[n] += sum_m C[m,n] of the values stored (fused bias gradient of an input-gradient GEMM)
};

struct GemmDesc {
    int layout;
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int epi;
    int splitk;     // >=1; >1 requires EPI_ATOMIC
    int unaligned;  // 1: some group base pointer is not 16-B aligned -> scalar global loads
    int c_aligned;  // 1: every group's C (and aux) base is 16-B aligned (group tables; inline groups are checked)
    int ragged;     // GemmRagged bits: per-group device-side row counts and row lists (the mask-aware decoder)
    int nimg;       // launch_gemm_split_nt: rows of the weight image B points into (0: = N)
};

extern unsigned long long* g_gemm_stamps;  // diagnostics: see shivae_debug_gemm_stamps
extern int g_gemm_split;                    // see shivae_set_gemm_split

// groups_dev: device array of ngroups GemmGroup, or nullptr with ngroups==1 and g0 used.
int launch_gemm(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                hipStream_t stream);

// ---- split-mode NT GEMM against pre-split weights (gemm_split.hip): GemmGroup::B = weight image
bool gemm_split_on();                              // SHIVAE_GEMM_SPLIT / shivae_set_gemm_split, resolved
size_t presplit_bytes(int N, int K);               // bytes of the image of one [N,K] weight matrix
// img[b] = three-plane bf16 image of W[b] ([N,K] row-major, leading dimension ld); images are presplit_bytes apart
int launch_presplit(const float* W, int N, int K, int ld, int batch, size_t in_stride, void* img, hipStream_t stream);
bool split_nt_fits(const GemmDesc& d);
int launch_gemm_split_nt(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                         hipStream_t stream);
// ---- split-mode TN GEMM on 128 x 256 x 16 tiles (gemm_split.hip): the weight gradients
bool split_tn_fits(const GemmDesc& d);
int launch_gemm_split_tn(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                         hipStream_t stream);

// bias gradient: out[g][n] += sum_m Y_g[m*ld + n]
struct ColsumGroup {
    const float* Y;
    float* out;
};
int launch_colsum(int M, int N, int ld, const ColsumGroup* groups_dev, int ngroups,
                  const ColsumGroup& g0, hipStream_t stream);

// ---------------------------------------------------------------- likelihood-parameter heads (heads.hip)
// One attribute's head Linear(H -> nout): streaming kernels for nout <= 8 (no MFMA).
struct HeadGroup {
    const float* hin;   // head input [rows][ldin] (a2 for real/pos, the hidden head layer for bin/cat)
    float* dhin;        // its gradient (backward), same layout
    const float* W;     // [nout][H]
    const float* b;     // [nout]
    float* gW;          // += weight gradient
    float* gb;          // += bias gradient
    float* gprev;       // [H] += column sums of dhin = bias gradient of the layer that produced hin
    float* P;           // this attribute's columns of the parameter rows [.][ldp]: outputs, then their gradients
    const int* cnt;     // live rows (mask-aware decoder) or null
    const int* idx;     // row i of hin belongs to parameter row idx[i], or null
};
bool head_kernels_fit(int H, int nout, int ldin, int ldp);
int launch_heads(bool bwd, const HeadGroup* groups_dev, int ngroups, int nout, int H, int ldin, int ldp, int rows,
                 bool use_cnt, bool use_idx, hipStream_t stream, int live_rows = 0);

struct AttrDesc;
// Row lists of the mask-aware decoder: for every attribute d, idx[d*ld_idx + i] = i-th row (ascending) of the
// chunk in which the attribute is observed, cnt[d] = how many (base_missvae.py:271-274 picks the mask column).
int launch_observed_rows(const AttrDesc* attrs_dev, int D, const uint8_t* mask, int ldm, int rows, int* idx,
                         int ld_idx, int* cnt, hipStream_t stream, int* rank = nullptr);
// out[row][0..E) = sum over the attributes d observed in the row of prod[rank[d][row]][d*blk .. +E)  (loss.hip)
int launch_l0_gather(const int* rank, int D, int ld_rank, const float* prod, size_t ldp, int blk, int E, float* out,
                     int ldo, int rows, hipStream_t stream);
// Zero Y[row][d*blk .. (d+1)*blk) for every (row, attribute d) that is missing (what a ragged launch that stores
// through a row list leaves untouched).
int launch_zero_missing_blocks(const AttrDesc* attrs_dev, int D, const uint8_t* mask, int ldm, int rows, float* Y,
                               int ldy, int blk, hipStream_t stream);


// ---------------------------------------------------------------- loss kernels (loss.hip)
// One attribute of types_list: where it lives in x (one-hot expanded, utils.py:291-315) and in the
// decoder-parameter row.
struct AttrDesc {
    int type;  // SHIVAE_REAL..SHIVAE_CAT
    int xoff;  // first x column
    int dim;   // x columns (1, or C for cat)
    int poff;  // first parameter column
    int nout;  // parameter columns: real/pos 1 (+1 if learn_std), bin 1, cat C-1
};

int launch_heter_nll(const AttrDesc* attrs_dev, int D, int learn_std, const float* x, int ldx,
                     const uint8_t* mask, int ldm, const float* P, int ldp, float* dP, int ldd, int rows,
                     float gs, float* rowloss, int X, int Pw, int group, int xspan, int pspan, hipStream_t stream);
// X, Pw: widths of x / P rows; group: attributes per staging step (multiple of 64); xspan, pspan: widest x / P span of a group
int launch_mc_impute(const AttrDesc* attrs_dev, int D, int learn_std, const float* P, int ldp, long long rows,
                     int n_samples, uint64_t seed, float* xhat, int ldx, hipStream_t stream);
int launch_kl(const float* mu, const float* sd, const float* mup, const float* sdp, int ldz,
              const float* probs, int ldk, float* dmu, float* dsd, float* dmup, float* dsdp_pre, float* dlg,
              int rows, int Z, int K, float kz, float* rowloss, hipStream_t stream);
int launch_loss_reduce(const float* rowloss, float* persum, int T, int B, float beta, float inv_batch,
                       float* losses, hipStream_t stream);
int launch_reparam_fwd(const float* logits, const float* u, int ldk, int K, float temp, float* s, float* probs,
                       const float* mean, const float* sd, const float* eps, int ldz, int Z, float* z, int rows,
                       hipStream_t stream);
int launch_reparam_bwd(const float* s, const float* ds, int ldk, int K, float temp, float* dlogits,
                       const float* eps, const float* dz, int ldz, int Z, float* dmean, float* dsd, int rows,
                       hipStream_t stream);

// ---------------------------------------------------------------- recurrent scan (scan.hip)
// float offsets into the packed-weight buffer; t_* are the transposed images used by the backward
struct ScanWOff {
    int xsh, lg, b_lg, e0s, e0h, e2, b_e2, mu, b_mu, sd, b_sd, pz, b_pz, ih, hh, b_g;
    int t_ih, t_hh, t_pz, t_mu, t_sd, t_e2, t_e0s, t_e0h, t_lg, t_xsh;
    int p0h, p0s, b_p0, pm, b_pm, ps, b_ps;  // prior network (generation scan only)
};

// Row-major record buffers, row = t*B + b, leading dims padded: Hp/Zp/Kp (multiples of 16),
// emb = [phi_z (Hp) | s (Kp) | h_prev (Hp)] with Ep = 2*Hp + Kp.  u/eps are the caller's unpadded
// [T][B][K] / [T][B][Z], or both null.
struct ScanArgs {
    int T, B, H, Z, K, Hp, Zp, Kp, Ep;
    float temp;
    const float* wpack;
    int wlds_floats;  // size of wpack if the kernel may keep all of it in LDS (small models), else 0
    ScanWOff w;
    const float *xs_pre, *ex_pre, *u, *eps;  // u == eps == nullptr: the scans draw the noise themselves (noise.h) under
    unsigned long long seed;                 // this seed -- the SAME numbers in the forward and the reverse scan
    float *emb, *as_, *e1, *e2, *logits, *probs, *mu, *sd, *z, *gates, *cnew, *h_last;
    const float *d_emb, *dmu_ext, *dsd_ext, *dlg_kl;
    float *d_gates, *d_phiz_pre, *d_mu, *d_sd_pre, *d_e2_pre, *d_e1_pre, *d_lg_pre, *d_as_pre;
};

// dst[packed] = W (or W^T) zero-padded; blocks model the 4 LSTM gates (i,f,g,o each padded H->Hp)
struct PackJob {
    const float* src;
    const float* src2;  // bias jobs: optional second addend (b_ih + b_hh)
    int ld, transposed;
    int Op, Ip;                // padded out / in extents (multiples of 16)
    int oblk_true, oblk_pad;   // out index o -> block o/oblk_pad, valid if o%oblk_pad < oblk_true
    int iblk_true, iblk_pad;
    int dst_off;
};

int launch_pack(const PackJob* wjobs_dev, int nw, const PackJob* bjobs_dev, int nb, float* dst, hipStream_t stream);
int launch_scan(const ScanArgs& a, bool bwd, hipStream_t stream);
// generation from the prior (shivae.py:156-214): writes emb, z, mu(=prior mean), sd(=prior std), probs(=1/K)
int launch_scan_gen(const ScanArgs& a, hipStream_t stream);
// u[rows][K] / eps[rows][Z] (either may be null) as the scans draw them under `seed`
int launch_noise_fill(unsigned long long seed, long long rows, int K, int Z, float* u, float* eps, hipStream_t stream);

// ---------------------------------------------------------------- misc (optim.hip)
int launch_compact_emb(const float* emb, int Ep, int Hp, int Kp, int H, int K, float* embc, int ldc, int rows,
                       hipStream_t stream);
int launch_expand_add_emb(const float* dembc, int ldc, int H, int K, float* d_emb, int Ep, int Hp, int Kp, int rows,
                          hipStream_t stream);
int launch_unpad_copy(const float* src, int ld, int col0, int width, float* dst, int64_t rows, hipStream_t stream);
int launch_copy2d(const float* src, int lds, int col0, int width, float* dst, int ldd, int dcol0, int64_t rows,
                  hipStream_t stream);
int launch_transpose(const float* in, float* out, int R, int C, int batch, size_t in_stride, size_t out_stride,
                     hipStream_t stream);
int launch_sqnorm(const float* buf, int64_t n, float* scratch, float* out, hipStream_t stream);
int launch_sqnorm_noalloc(const float* buf, int64_t n, float* out, hipStream_t stream);
int init_sqnorm_noalloc();
int launch_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm, float clip,
                     float lr, float b1, float b2, float eps, int step, hipStream_t stream);

}  // namespace shivae
