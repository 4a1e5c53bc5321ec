// Plan (model description, flat parameter layout, workspace carving, static group tables) and the
// host-side orchestration of one ELBO forward+backward step, exported through the C ABI of
// include/shivae_hip.h.
//
// Order of one step (reference: trainers.py:212-215 -> base_missvae.py:144-231 -> shivae.py:72-154):
//   pack scan weights | hoisted phi_x GEMMs | scan forward | per row-chunk: {compact emb, prior,
//   KL terms (+grads), decoder forward, masked NLL (+grads), decoder backward, prior backward,
//   scatter d_emb} | scan backward | batched weight gradients of the cell + phi_x backward | loss sums.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is looked up at run time (see rccl_api)

#include "common.h"

namespace shivae {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- HIP-event profiler: one (start, stop) pair per launch, resolved when the caller collects
// A ragged GEMM launch (mask-aware decoder) only multiplies the observed rows of every group: its work is the dense
// figure scaled by sum(cnt[attr of group]) / (groups * extent).  The counts live on the device; while profiling,
// each chunk's counts are copied to a pinned host block and the record keeps a pointer to it.
struct ProfRec { int id; double work; hipEvent_t a, b; int M, N, K, G; const std::vector<int>* rag_attr; const int* rag_cnt; int rag_ext; };
static int g_pm = 0, g_pn = 0, g_pk = 0, g_pg = 0;
void prof_shape(int M, int N, int K, int G) { g_pm = M; g_pn = N; g_pk = K; g_pg = G; }
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static const std::vector<int>* g_rag_attr = nullptr;
static const int* g_rag_cnt = nullptr;
static int g_rag_ext = 0;
static void prof_ragged(const std::vector<int>* attr, const int* cnt_host, int ext) { g_rag_attr = attr; g_rag_cnt = cnt_host; g_rag_ext = ext; }
void prof_begin(int id, double work, hipStream_t s) {
    if (!g_prof_on) return;
    ProfRec r{id, work, nullptr, nullptr, g_pm, g_pn, g_pk, g_pg, g_rag_attr, g_rag_cnt, g_rag_ext};
    g_pm = g_pn = g_pk = g_pg = 0;
    g_rag_attr = nullptr; g_rag_cnt = nullptr; g_rag_ext = 0;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
static bool prof_on() { return g_prof_on; }
// a plan is going away: records of ragged launches must not read its pinned counts any more
static void prof_forget_counts(const int* lo, const int* hi) {
    for (auto& r : g_prof)
        if (r.rag_cnt >= lo && r.rag_cnt < hi) { r.rag_cnt = nullptr; r.rag_attr = nullptr; }
}
void prof_end(hipStream_t s) {
    if (!g_prof_on || g_prof.empty()) return;
    (void)hipEventRecord(g_prof.back().b, s);
}

struct Tensor {
    int64_t off = -1;
    int rows = 0, cols = 0;
};

struct GroupTable {  // device table of GemmGroup + matching bias-gradient table
    GemmGroup* dev = nullptr;
    int n = 0;
    int unaligned = 0;
    int c_aligned = 1;
    std::vector<int> attr;  // attribute of each group (ragged launches: whose row count applies)
    GemmGroup* dev_img = nullptr;  // the same groups with B = the weight's pre-split image (gemm_split.hip), or null
};
struct ColTable {
    ColsumGroup* dev = nullptr;
    int n = 0;
};
struct HeadTable {
    HeadGroup* dev = nullptr;
    int n = 0;
    std::vector<int> attr;
};

}  // namespace shivae

using namespace shivae;

struct shivae_plan {
    int T, B, H, Z, K, X, D, P, E, Hp, Zp, Kp, Ep, E4, Z4, learn_std;
    int ldA0, ldA1, ldHH;  // leading dims of the decoder activations (+64 floats: no power-of-two row strides)
    int64_t R;
    int Rc;
    // Row capacity of the decoder's ACTIVATION buffers (a0 .. dhh), <= Rc.  A mask-aware training chunk keeps attribute
    // d's activations compactly (row i = its i-th observed row), so with most entries missing a chunk of Rc rows only
    // ever fills the first max_d cnt_d rows of them: shivae_plan_set_activation_rows lets the chunk be several times
    // larger than what the activations could hold densely.  Dense passes (forward only, imputation, generation,
    // SHIVAE_COMPACT=0) step through the rows Ra at a time; a training chunk whose counts exceed Ra is re-run in
    // pieces of Ra rows (n_overflow counts them).
    int Ra;
    int* cnt_guard = nullptr;  // pinned host copy of a chunk's counts for that check (allocated in shivae_plan_bind)
    long long n_overflow = 0;
    // Side streams for SMALL steps: the ~40 launches behind the reverse scan (the cell's weight and bias gradients, the
    // phi_x backward) are independent of each other and, on a model of a few thousand rows, a few microseconds each --
    // in one stream a chain of launch latencies.  Forked over the step's stream and these (events; the pattern a
    // stream capture records as parallel branches), joined before the step goes on.  SHIVAE_FANOUT=0: one stream.
    static constexpr int NSIDE = 3;
    hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NSIDE] = {nullptr, nullptr, nullptr};
    int nbc;  // attributes with a hidden head layer (bin/cat)
    int nll_group = 64, nll_xspan = 0, nll_pspan = 0;  // likelihood kernel: attributes per staging step, widest spans
    std::vector<AttrDesc> attrs;
    std::vector<int> bc_index;  // attr -> index among bin/cat, or -1
    std::vector<int> bc_attr;   // inverse
    // parameter layout
    Tensor glob[SHIVAE_P_COUNT];
    std::vector<Tensor> per_attr[SHIVAE_P_COUNT];
    int64_t n_params = 0;
    // workspace
    size_t ws_bytes = 0;
    std::map<std::string, size_t> ws_off;
    char* ws = nullptr;
    float *params = nullptr, *grads = nullptr;
    bool bound = false;
    int last_flags = 0;
    // device tables (library-owned)
    AttrDesc* attrs_dev = nullptr;
    PackJob *wjobs_dev = nullptr, *bjobs_dev = nullptr;
    int n_wjobs = 0, n_bjobs = 0;
    ScanWOff woff;
    size_t wpack_floats = 0;
    std::vector<void*> owned;  // device allocations to free
    // pinned host ring of per-chunk observed-row counts (D ints per slot), written only while the HIP-event profiler
    // is on: allocated once in shivae_plan_bind, never inside a step
    int* cnt_host = nullptr;
    int cnt_slots = 0, cnt_next = 0;
    // grouped-GEMM tables
    GroupTable b_l0_w, b_l0_wr;  // layer-0 weight gradient per attribute (ragged steps): main columns + the E % 128 remainder
    GroupTable b_l0_d, b_l0_dr;  // layer-0 input gradient per attribute over its observed rows, added into the chunk's rows
    GroupTable b_l0_p, b_l0_pr;  // the same products stored per attribute for launch_l0_gather (l0_prod)
    float* l0_prod = nullptr;    // [Ra][D * E4] over the dead a1 .. da2 buffers, or null when they are too small
    int l0_gather = 1;           // SHIVAE_L0D_GATHER=0: the atomic form
    int demb_mode = 2;           // SHIVAE_DEMB: 0 "dense" (one GEMM over all attributes' column blocks), 1 "compact", 2 "auto"
    float obs_frac = -1.f;       // auto: observed fraction of the first row chunk this plan trained on (< 0: not seen yet)
    unsigned long long noise_seed = 0;  // seed of the scans' in-kernel noise (calls with u == eps == nullptr)
    int l0_nmain = 0;
    GroupTable f_l0, f_l1, f_l2, f_h0, b_l1_w, b_l1_d, b_l2_w, b_l2_d, b_h0_w, b_h0_d;
    int compact = 1;  // mask-aware decoder in training steps (SHIVAE_COMPACT=0: dense)
    std::map<int, GroupTable> f_out, b_out_w, b_out_d;  // keyed by nout
    std::map<int, ColTable> c_out;
    std::map<int, HeadTable> h_out;  // the same heads for the streaming kernels of heads.hip (nout <= 8)
    int stream_heads = 1;            // SHIVAE_HEADS=gemm: heads through the grouped GEMMs (comparison runs)

    float* wsf(const char* name) const { return reinterpret_cast<float*>(ws + ws_off.at(name)); }
    float* prm(const Tensor& t) const { return params + t.off; }
    float* grd(const Tensor& t) const { return grads + t.off; }
};

namespace shivae {

static int64_t add_tensor(shivae_plan* p, Tensor& t, int rows, int cols) {
    p->n_params = round_up64(p->n_params, 4);
    t.off = p->n_params;
    t.rows = rows;
    t.cols = cols;
    p->n_params += (int64_t)rows * cols;
    return t.off;
}

static void build_layout(shivae_plan* p) {
    const int H = p->H, Z = p->Z, K = p->K, X = p->X, D = p->D, E = p->E;
    auto G = [&](int id, int r, int c) { add_tensor(p, p->glob[id], r, c); };
    G(SHIVAE_P_PHIX0_W, H, X); G(SHIVAE_P_PHIX0_B, 1, H);
    G(SHIVAE_P_PHIX2_W, H, H); G(SHIVAE_P_PHIX2_B, 1, H);
    G(SHIVAE_P_XS0_W, H, X + H); G(SHIVAE_P_XS0_B, 1, H);
    G(SHIVAE_P_XS2_W, K, H); G(SHIVAE_P_XS2_B, 1, K);
    G(SHIVAE_P_E0_W, H, E); G(SHIVAE_P_E0_B, 1, H);
    G(SHIVAE_P_E2_W, H, H); G(SHIVAE_P_E2_B, 1, H);
    G(SHIVAE_P_ENCMEAN_W, Z, H); G(SHIVAE_P_ENCMEAN_B, 1, Z);
    G(SHIVAE_P_ENCSTD_W, Z, H); G(SHIVAE_P_ENCSTD_B, 1, Z);
    G(SHIVAE_P_PHIZ_W, H, Z); G(SHIVAE_P_PHIZ_B, 1, H);
    G(SHIVAE_P_PRIOR0_W, H, H + K); G(SHIVAE_P_PRIOR0_B, 1, H);
    G(SHIVAE_P_PRIORMEAN_W, Z, H); G(SHIVAE_P_PRIORMEAN_B, 1, Z);
    G(SHIVAE_P_PRIORSTD_W, Z, H); G(SHIVAE_P_PRIORSTD_B, 1, Z);
    G(SHIVAE_P_RNN_WIH, 4 * H, H); G(SHIVAE_P_RNN_WHH, 4 * H, H);
    G(SHIVAE_P_RNN_BIH, 1, 4 * H); G(SHIVAE_P_RNN_BHH, 1, 4 * H);
    // decoder: every kind stacked over the attributes (uniform stride) so that layer 0 of ALL
    // attributes is one [D*2H, E] matrix and the heads' output rows form one [P, H] matrix.
    auto stack = [&](int id, int rows, int cols, int count) {
        Tensor base;
        add_tensor(p, base, rows * count, cols);
        p->glob[id] = base;
        p->per_attr[id].assign(D, Tensor());
        return base;
    };
    Tensor t;
    t = stack(SHIVAE_P_DEC0_W, 2 * H, E, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC0_W][d] = Tensor{t.off + (int64_t)d * 2 * H * E, 2 * H, E};
    t = stack(SHIVAE_P_DEC0_B, 1, 2 * H, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC0_B][d] = Tensor{t.off + (int64_t)d * 2 * H, 1, 2 * H};
    t = stack(SHIVAE_P_DEC2_W, H, 2 * H, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC2_W][d] = Tensor{t.off + (int64_t)d * H * 2 * H, H, 2 * H};
    t = stack(SHIVAE_P_DEC2_B, 1, H, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC2_B][d] = Tensor{t.off + (int64_t)d * H, 1, H};
    t = stack(SHIVAE_P_DEC4_W, H, H, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC4_W][d] = Tensor{t.off + (int64_t)d * H * H, H, H};
    t = stack(SHIVAE_P_DEC4_B, 1, H, D);
    for (int d = 0; d < D; ++d) p->per_attr[SHIVAE_P_DEC4_B][d] = Tensor{t.off + (int64_t)d * H, 1, H};
    t = stack(SHIVAE_P_HEAD0_W, H, H, std::max(p->nbc, 1));
    for (int d = 0; d < D; ++d)
        if (p->bc_index[d] >= 0) p->per_attr[SHIVAE_P_HEAD0_W][d] = Tensor{t.off + (int64_t)p->bc_index[d] * H * H, H, H};
    t = stack(SHIVAE_P_HEAD0_B, 1, H, std::max(p->nbc, 1));
    for (int d = 0; d < D; ++d)
        if (p->bc_index[d] >= 0) p->per_attr[SHIVAE_P_HEAD0_B][d] = Tensor{t.off + (int64_t)p->bc_index[d] * H, 1, H};
    // head output rows: [P, H] in attribute order; views by kind
    Tensor wout, bout;
    add_tensor(p, wout, p->P, H);
    add_tensor(p, bout, 1, p->P);
    for (int id : {SHIVAE_P_HEADMEAN_W, SHIVAE_P_HEADMEAN_B, SHIVAE_P_HEADSTD_W, SHIVAE_P_HEADSTD_B,
                   SHIVAE_P_HEADOUT_W, SHIVAE_P_HEADOUT_B})
        p->per_attr[id].assign(D, Tensor());
    p->glob[SHIVAE_P_HEADOUT_W] = wout;
    p->glob[SHIVAE_P_HEADOUT_B] = bout;
    for (int d = 0; d < D; ++d) {
        const AttrDesc& a = p->attrs[d];
        if (a.type <= SHIVAE_POS) {
            p->per_attr[SHIVAE_P_HEADMEAN_W][d] = Tensor{wout.off + (int64_t)a.poff * H, 1, H};
            p->per_attr[SHIVAE_P_HEADMEAN_B][d] = Tensor{bout.off + a.poff, 1, 1};
            if (p->learn_std) {
                p->per_attr[SHIVAE_P_HEADSTD_W][d] = Tensor{wout.off + (int64_t)(a.poff + 1) * H, 1, H};
                p->per_attr[SHIVAE_P_HEADSTD_B][d] = Tensor{bout.off + a.poff + 1, 1, 1};
            }
        } else {
            p->per_attr[SHIVAE_P_HEADOUT_W][d] = Tensor{wout.off + (int64_t)a.poff * H, a.nout, H};
            p->per_attr[SHIVAE_P_HEADOUT_B][d] = Tensor{bout.off + a.poff, 1, a.nout};
        }
    }
    p->n_params = round_up64(p->n_params, 4);
}

static void carve(shivae_plan* p, const char* name, size_t floats, size_t& cur) {
    p->ws_off[name] = cur;
    cur += round_up64((int64_t)(floats * sizeof(float)), 256);
}

static void build_workspace(shivae_plan* p) {
    size_t cur = 0;
    const size_t R = (size_t)p->R, Rc = (size_t)p->Rc, Hp = p->Hp, Zp = p->Zp, Kp = p->Kp, Ep = p->Ep;
    const size_t H = p->H, D = p->D;
    for (const char* n : {"p1", "phx", "xs_pre", "ex_pre", "as", "e1", "e2", "cnew", "d_phiz_pre", "d_e2_pre",
                          "d_e1_pre", "d_as_pre", "d_phx_pre", "d_p1_pre"})
        carve(p, n, R * Hp, cur);
    carve(p, "emb", R * Ep, cur);
    carve(p, "d_emb", R * Ep, cur);
    carve(p, "gates", R * 4 * Hp, cur);
    carve(p, "d_gates", R * 4 * Hp, cur);
    for (const char* n : {"mu", "sd", "z", "mup", "sdp", "dmu_ext", "dsd_ext", "d_mu", "d_sd_pre"}) carve(p, n, R * Zp, cur);
    for (const char* n : {"logits", "probs", "dlg_kl", "d_lg_pre"}) carve(p, n, R * Kp, cur);
    carve(p, "rowloss", R * 6, cur);
    carve(p, "persum", (size_t)p->B * 6, cur);
    carve(p, "pfull", R * p->P, cur);
    // chunk-local
    carve(p, "embc", Rc * p->E4, cur);
    carve(p, "dembc", Rc * p->E4, cur);
    const size_t Ra = (size_t)p->Ra;  // (activation rows: shivae_plan::Ra)
    carve(p, "a0", Ra * p->ldA0, cur);
    carve(p, "da0", Ra * p->ldA0, cur);
    for (const char* n : {"a1", "a2", "da1", "da2"}) carve(p, n, Ra * p->ldA1, cur);
    carve(p, "hh", Ra * p->ldHH, cur);
    carve(p, "dhh", Ra * p->ldHH, cur);
    carve(p, "pbuf", Rc * p->P, cur);
    carve(p, "rowidx", D * Rc, cur);  // int: per attribute, the chunk's observed rows (launch_observed_rows)
    carve(p, "rowcnt", D, cur);       // int: how many
    carve(p, "rowrank", D * Rc, cur);  // int: the inverse of rowidx (row -> place in the attribute's list, or -1)
    carve(p, "wpack", p->wpack_floats, cur);
    // K-contiguous (transposed) copies of the decoder weights for the input-gradient GEMMs
    carve(p, "wt0", (size_t)p->E * D * 2 * H, cur);
    carve(p, "wt2", D * 2 * H * H, cur);
    carve(p, "wt4", D * H * H, cur);
    carve(p, "wth0", (size_t)std::max(p->nbc, 1) * H * H, cur);
    // pre-split bf16 images of the same weights and of their transposes (split mode, gemm_split.hip)
    auto img = [&](const char* name, size_t count, int N, int K) { carve(p, name, count * presplit_bytes(N, K) / sizeof(float), cur); };
    img("wi_f0", D, 2 * p->H, p->E); img("wi_f2", D, p->H, 2 * p->H); img("wi_f4", D, p->H, p->H);
    img("wi_fh0", std::max(p->nbc, 1), p->H, p->H);
    img("wi_t0", 1, p->E, p->D * 2 * p->H); img("wi_t2", D, 2 * p->H, p->H); img("wi_t4", D, p->H, p->H);
    img("wi_th0", std::max(p->nbc, 1), p->H, p->H);
    p->ws_bytes = cur;
}

// ------------------------------------------------------------ packed scan weights: offsets + jobs
static void build_pack_plan(shivae_plan* p, std::vector<PackJob>& wj, std::vector<PackJob>& bj, bool with_ptrs) {
    const int H = p->H, Z = p->Z, K = p->K, X = p->X, Hp = p->Hp, Zp = p->Zp, Kp = p->Kp, E = p->E;
    int cur = 0;
    auto W = [&](int& off, const Tensor& t, int col0, int O, int Op, int I, int Ip, bool transposed, int oblocks,
                 int iblocks) {
        // (O, I) = true out/in extents of the PACKED operator; for transposed jobs the source is [I rows, O cols]
        off = cur;
        PackJob j;
        memset(&j, 0, sizeof(j));
        j.src = with_ptrs ? p->params + t.off + col0 : nullptr;
        j.ld = t.cols;
        j.transposed = transposed;
        j.Op = Op * oblocks; j.Ip = Ip * iblocks;
        j.oblk_true = O; j.oblk_pad = Op; j.iblk_true = I; j.iblk_pad = Ip;
        j.dst_off = cur;
        wj.push_back(j);
        cur += j.Op * j.Ip;
    };
    auto Bv = [&](int& off, const Tensor& t, const Tensor* t2, int O, int Op, int oblocks) {
        off = cur;
        PackJob j;
        memset(&j, 0, sizeof(j));
        j.src = with_ptrs ? p->params + t.off : nullptr;
        j.src2 = (with_ptrs && t2) ? p->params + t2->off : nullptr;
        j.Op = Op * oblocks; j.oblk_true = O; j.oblk_pad = Op;
        j.iblk_true = j.iblk_pad = 1;
        j.dst_off = cur;
        bj.push_back(j);
        cur += round_up(j.Op, 4);
    };
    ScanWOff& w = p->woff;
    const Tensor* g = p->glob;
    // forward images: out = rows of W
    W(w.xsh, g[SHIVAE_P_XS0_W], X, H, Hp, H, Hp, false, 1, 1);          // Wxs0[:, X:]   (encoders.py:54)
    W(w.lg, g[SHIVAE_P_XS2_W], 0, K, Kp, H, Hp, false, 1, 1);
    Bv(w.b_lg, g[SHIVAE_P_XS2_B], nullptr, K, Kp, 1);
    W(w.e0s, g[SHIVAE_P_E0_W], H, H, Hp, K, Kp, false, 1, 1);           // We0[:, H:H+K] (encoders.py:61)
    W(w.e0h, g[SHIVAE_P_E0_W], H + K, H, Hp, H, Hp, false, 1, 1);       // We0[:, H+K:]
    W(w.e2, g[SHIVAE_P_E2_W], 0, H, Hp, H, Hp, false, 1, 1);
    Bv(w.b_e2, g[SHIVAE_P_E2_B], nullptr, H, Hp, 1);
    W(w.mu, g[SHIVAE_P_ENCMEAN_W], 0, Z, Zp, H, Hp, false, 1, 1);
    Bv(w.b_mu, g[SHIVAE_P_ENCMEAN_B], nullptr, Z, Zp, 1);
    W(w.sd, g[SHIVAE_P_ENCSTD_W], 0, Z, Zp, H, Hp, false, 1, 1);
    Bv(w.b_sd, g[SHIVAE_P_ENCSTD_B], nullptr, Z, Zp, 1);
    W(w.pz, g[SHIVAE_P_PHIZ_W], 0, H, Hp, Z, Zp, false, 1, 1);
    Bv(w.b_pz, g[SHIVAE_P_PHIZ_B], nullptr, H, Hp, 1);
    W(w.ih, g[SHIVAE_P_RNN_WIH], 0, H, Hp, H, Hp, false, 4, 1);         // 4 gate blocks of rows
    W(w.hh, g[SHIVAE_P_RNN_WHH], 0, H, Hp, H, Hp, false, 4, 1);
    Bv(w.b_g, g[SHIVAE_P_RNN_BIH], &g[SHIVAE_P_RNN_BHH], H, Hp, 4);
    // transposed images for the backward: out = columns of W
    W(w.t_ih, g[SHIVAE_P_RNN_WIH], 0, H, Hp, H, Hp, true, 1, 4);
    W(w.t_hh, g[SHIVAE_P_RNN_WHH], 0, H, Hp, H, Hp, true, 1, 4);
    W(w.t_pz, g[SHIVAE_P_PHIZ_W], 0, Z, Zp, H, Hp, true, 1, 1);
    W(w.t_mu, g[SHIVAE_P_ENCMEAN_W], 0, H, Hp, Z, Zp, true, 1, 1);
    W(w.t_sd, g[SHIVAE_P_ENCSTD_W], 0, H, Hp, Z, Zp, true, 1, 1);
    W(w.t_e2, g[SHIVAE_P_E2_W], 0, H, Hp, H, Hp, true, 1, 1);
    W(w.t_e0s, g[SHIVAE_P_E0_W], H, K, Kp, H, Hp, true, 1, 1);
    W(w.t_e0h, g[SHIVAE_P_E0_W], H + K, H, Hp, H, Hp, true, 1, 1);
    W(w.t_lg, g[SHIVAE_P_XS2_W], 0, H, Hp, K, Kp, true, 1, 1);
    W(w.t_xsh, g[SHIVAE_P_XS0_W], X, H, Hp, H, Hp, true, 1, 1);
    // prior network for the generation scan (shivae.py:56-61,184-186)
    W(w.p0h, g[SHIVAE_P_PRIOR0_W], 0, H, Hp, H, Hp, false, 1, 1);       // prior.0[:, :H] acts on h
    W(w.p0s, g[SHIVAE_P_PRIOR0_W], H, H, Hp, K, Kp, false, 1, 1);       // prior.0[:, H:] acts on s
    Bv(w.b_p0, g[SHIVAE_P_PRIOR0_B], nullptr, H, Hp, 1);
    W(w.pm, g[SHIVAE_P_PRIORMEAN_W], 0, Z, Zp, H, Hp, false, 1, 1);
    Bv(w.b_pm, g[SHIVAE_P_PRIORMEAN_B], nullptr, Z, Zp, 1);
    W(w.ps, g[SHIVAE_P_PRIORSTD_W], 0, Z, Zp, H, Hp, false, 1, 1);
    Bv(w.b_ps, g[SHIVAE_P_PRIORSTD_B], nullptr, Z, Zp, 1);
    (void)E;
    p->wpack_floats = cur;
}

template <class T>
static int upload(shivae_plan* p, const std::vector<T>& host, T** dev) {
    *dev = nullptr;
    if (host.empty()) return SHIVAE_OK;
    SV_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(T)));
    p->owned.push_back(*dev);
    SV_CHECK_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SHIVAE_OK;
}

static int make_table(shivae_plan* p, const std::vector<GemmGroup>& g, GroupTable& t,
                      const std::vector<int>& attr = std::vector<int>()) {
    t.n = (int)g.size();
    t.attr = attr;
    t.unaligned = 0;
    t.c_aligned = 1;
    for (const GemmGroup& x : g)
    {
        if ((reinterpret_cast<uintptr_t>(x.A) | reinterpret_cast<uintptr_t>(x.B)) & 15) t.unaligned = 1;
        if ((reinterpret_cast<uintptr_t>(x.C) | reinterpret_cast<uintptr_t>(x.aux)) & 15) t.c_aligned = 0;
    }
    return upload(p, g, &t.dev);
}
static int make_ctable(shivae_plan* p, const std::vector<ColsumGroup>& g, ColTable& t) {
    t.n = (int)g.size();
    return upload(p, g, &t.dev);
}

static int build_tables(shivae_plan* p) {
    const int H = p->H, D = p->D;
    float *a0 = p->wsf("a0"), *a1 = p->wsf("a1"), *a2 = p->wsf("a2"), *hh = p->wsf("hh"), *pb = p->wsf("pbuf");
    float *da0 = p->wsf("da0"), *da1 = p->wsf("da1"), *da2 = p->wsf("da2"), *dhh = p->wsf("dhh");
    auto PA = [&](int id, int d) { return p->params + p->per_attr[id][d].off; };
    auto GA = [&](int id, int d) { return p->grads + p->per_attr[id][d].off; };
    // Row lists of the mask-aware decoder (launch_observed_rows): every group carries its attribute's count and
    // list; a launch says through GemmDesc::ragged which of them it honours, so one table serves both forms.
    int* rowidx = reinterpret_cast<int*>(p->wsf("rowidx"));
    int* rowcnt = reinterpret_cast<int*>(p->wsf("rowcnt"));
    const float* embc = p->wsf("embc");
    std::vector<GemmGroup> fl0, fl1, fl2, fh0, wl0, wl0r, dl0, dl0r, wl1, dl1, wl2, dl2, wh0, dh0;
    std::vector<GemmGroup> fl0i, fl1i, fl2i, fh0i, dl0i, dl1i, dl2i, dh0i, pl0i;  // B = pre-split weight image
    std::vector<GemmGroup> pl0, pl0r;
    auto IMG = [&](const char* name, size_t idx, int N, int K) {
        return reinterpret_cast<const float*>(p->ws + p->ws_off.at(name) + idx * presplit_bytes(N, K));
    };
    auto twin = [&](std::vector<GemmGroup>& dst, const GemmGroup& src, const float* image) {
        GemmGroup t = src;
        t.B = image;
        dst.push_back(t);
    };
    float* dembc = p->wsf("dembc");
    {
        // a few columns past a multiple of 128 (E = 528) go to a skinny launch of their own, as in gemm1
        const int rem = p->E % 128;
        p->l0_nmain = (p->E > 128 && rem > 0 && rem <= 32) ? p->E - rem : p->E;
        // Product rows of the compact layer-0 input gradient (launch_l0_gather): by then the layer-1/2 and head
        // activations and their gradients (a1, a2, da1, da2: carved back to back) are dead for the chunk.
        const size_t room = (p->ws_off.at("hh") - p->ws_off.at("a1")) / sizeof(float);
        p->l0_prod = ((size_t)p->Ra * D * p->E4 <= room && p->E <= 1024) ? a1 : nullptr;
    }
    std::vector<int> all, bc;
    std::map<int, std::vector<GemmGroup>> fo, wo, dO;
    std::map<int, std::vector<int>> ko;
    std::map<int, std::vector<ColsumGroup>> co;
    std::map<int, std::vector<HeadGroup>> ho;
    for (int d = 0; d < D; ++d) {
        const AttrDesc& a = p->attrs[d];
        const int* cnt = rowcnt + d;
        const int* list = rowidx + (size_t)d * p->Rc;
        // {A, B, C, bias, aux, csum} + {cnt, aidx, bidx, cidx}
        auto G = [&](const float* A, const float* B, float* C, const float* bias, const float* aux, float* csum,
                     const int* aidx = nullptr, const int* bidx = nullptr, const int* cidx = nullptr) {
            return GemmGroup{A, B, C, bias, aux, csum, cnt, aidx, bidx, cidx};
        };
        float *a0d = a0 + (size_t)d * 2 * H, *a1d = a1 + (size_t)d * H, *a2d = a2 + (size_t)d * H;
        float *da0d = da0 + (size_t)d * 2 * H, *da1d = da1 + (size_t)d * H, *da2d = da2 + (size_t)d * H;
        all.push_back(d);
        // trunk (decoders.py:60-66).  Layer 0 as a group: the shared input gathered through the attribute's rows.
        fl0.push_back(G(embc, PA(SHIVAE_P_DEC0_W, d), a0d, PA(SHIVAE_P_DEC0_B, d), nullptr, nullptr, list));
        // its weight gradient: the shared input comes through the row list (bidx); da0 too (aidx) when it sits in
        // chunk rows (SHIVAE_DEMB=dense, see dl1)
        wl0.push_back(G(da0d, embc, GA(SHIVAE_P_DEC0_W, d), nullptr, nullptr, nullptr, list, list));
        if (p->l0_nmain < p->E)
            wl0r.push_back(G(da0d, embc + p->l0_nmain, GA(SHIVAE_P_DEC0_W, d) + p->l0_nmain, nullptr, nullptr, nullptr, list, list));
        // its input gradient over the observed rows: da0_d [cnt, 2H] x W0_d [2H, E], added into the chunk's rows (cidx)
        dl0.push_back(G(da0d, PA(SHIVAE_P_DEC0_W, d), dembc, nullptr, nullptr, nullptr, nullptr, nullptr, list));
        if (p->l0_nmain < p->E)
            dl0r.push_back(G(da0d, PA(SHIVAE_P_DEC0_W, d) + p->l0_nmain, dembc + p->l0_nmain, nullptr, nullptr, nullptr, nullptr, nullptr, list));
        if (p->l0_prod) {
            float* pd0 = p->l0_prod + (size_t)d * p->E4;
            pl0.push_back(G(da0d, PA(SHIVAE_P_DEC0_W, d), pd0, nullptr, nullptr, nullptr));
            if (p->l0_nmain < p->E)
                pl0r.push_back(G(da0d, PA(SHIVAE_P_DEC0_W, d) + p->l0_nmain, pd0 + p->l0_nmain, nullptr, nullptr, nullptr));
        }
        fl1.push_back(G(a0d, PA(SHIVAE_P_DEC2_W, d), a1d, PA(SHIVAE_P_DEC2_B, d), nullptr, nullptr));
        fl2.push_back(G(a1d, PA(SHIVAE_P_DEC4_W, d), a2d, PA(SHIVAE_P_DEC4_B, d), nullptr, nullptr));
        twin(fl0i, fl0.back(), IMG("wi_f0", d, 2 * H, p->E));
        // compact layer-0 input gradient against attribute d's K range of the transposed stack's image (rows = E,
        // K-tiles d*2H/32 ...: the image is K-tile major, so a K range is a contiguous slab)
        if ((2 * H) % 32 == 0) {
            const float* slab = reinterpret_cast<const float*>(reinterpret_cast<const char*>(IMG("wi_t0", 0, p->E, D * 2 * H)) +
                                                               (size_t)d * (2 * H / 32) * 3 * p->E * 64);
            twin(dl0i, dl0.back(), slab);
            if (p->l0_prod) twin(pl0i, pl0.back(), slab);
        }
        twin(fl1i, fl1.back(), IMG("wi_f2", d, H, 2 * H));
        twin(fl2i, fl2.back(), IMG("wi_f4", d, H, H));
        wl2.push_back(G(da2d, a1d, GA(SHIVAE_P_DEC4_W, d), nullptr, nullptr, nullptr));
        dl2.push_back(G(da2d, p->wsf("wt4") + (size_t)d * H * H, da1d, nullptr, a1d, GA(SHIVAE_P_DEC2_B, d)));
        wl1.push_back(G(da1d, a0d, GA(SHIVAE_P_DEC2_W, d), nullptr, nullptr, nullptr));
        // (cidx: with SHIVAE_DEMB=dense the layer-0 backward is dense over the chunk, and a ragged launch scatters da0
        // back to chunk rows)
        dl1.push_back(G(da1d, p->wsf("wt2") + (size_t)d * 2 * H * H, da0d, nullptr, a0d,
                        p->grads + p->glob[SHIVAE_P_DEC0_B].off + (size_t)d * 2 * H, nullptr, nullptr, list));
        twin(dl2i, dl2.back(), IMG("wi_t4", d, H, H));
        twin(dl1i, dl1.back(), IMG("wi_t2", d, 2 * H, H));
        // heads (decoders.py:69-102)
        float* wout = p->params + p->glob[SHIVAE_P_HEADOUT_W].off + (size_t)a.poff * H;
        float* bout = p->params + p->glob[SHIVAE_P_HEADOUT_B].off + a.poff;
        float* gwout = p->grads + p->glob[SHIVAE_P_HEADOUT_W].off + (size_t)a.poff * H;
        float* gbout = p->grads + p->glob[SHIVAE_P_HEADOUT_B].off + a.poff;
        float* pd = pb + a.poff;
        const float* hin = a2d;
        float *dhin = da2d, *hmask = a2d, *dhbias = GA(SHIVAE_P_DEC4_B, d);
        if (p->bc_index[d] >= 0) {
            const int jx = p->bc_index[d];
            float *hhj = hh + (size_t)jx * H, *dhhj = dhh + (size_t)jx * H;
            bc.push_back(d);
            fh0.push_back(G(a2d, PA(SHIVAE_P_HEAD0_W, d), hhj, PA(SHIVAE_P_HEAD0_B, d), nullptr, nullptr));
            wh0.push_back(G(dhhj, a2d, GA(SHIVAE_P_HEAD0_W, d), nullptr, nullptr, nullptr));
            dh0.push_back(G(dhhj, p->wsf("wth0") + (size_t)jx * H * H, da2d, nullptr, a2d, GA(SHIVAE_P_DEC4_B, d)));
            twin(fh0i, fh0.back(), IMG("wi_fh0", jx, H, H));
            twin(dh0i, dh0.back(), IMG("wi_th0", jx, H, H));
            hin = hhj; dhin = dhhj; hmask = hhj; dhbias = GA(SHIVAE_P_HEAD0_B, d);
        }
        // tables are keyed by +nout (input = hh, bin/cat) or -nout (input = a2, real/pos): the two
        // inputs have different leading dimensions, and a grouped launch shares them.
        // The parameter rows (pbuf) stay in chunk-row order for the likelihood kernel: the head output is stored
        // through the row list (cidx) and its gradient is read back through it (aidx).
        const int key = p->bc_index[d] >= 0 ? a.nout : -a.nout;
        ko[key].push_back(d);
        fo[key].push_back(G(hin, wout, pd, bout, nullptr, nullptr, nullptr, nullptr, list));
        wo[key].push_back(G(pd, hin, gwout, nullptr, nullptr, nullptr, list));
        co[key].push_back({pd, gbout});
        dO[key].push_back(G(pd, wout, dhin, nullptr, hmask, dhbias, list));
        ho[key].push_back(HeadGroup{hin, dhin, wout, bout, gwout, gbout, dhbias, pd, cnt, list});
    }
    SV_TRY(make_table(p, fl0, p->f_l0, all));
    SV_TRY(make_table(p, wl0, p->b_l0_w, all)); SV_TRY(make_table(p, wl0r, p->b_l0_wr, all));
    SV_TRY(make_table(p, dl0, p->b_l0_d, all)); SV_TRY(make_table(p, dl0r, p->b_l0_dr, all));
    if (p->l0_prod) {
        SV_TRY(make_table(p, pl0, p->b_l0_p, all)); SV_TRY(make_table(p, pl0r, p->b_l0_pr, all));
        SV_TRY(upload(p, pl0i, &p->b_l0_p.dev_img));
    }
    SV_TRY(make_table(p, fl1, p->f_l1, all)); SV_TRY(make_table(p, fl2, p->f_l2, all)); SV_TRY(make_table(p, fh0, p->f_h0, bc));
    SV_TRY(make_table(p, wl1, p->b_l1_w, all)); SV_TRY(make_table(p, dl1, p->b_l1_d, all));
    SV_TRY(make_table(p, wl2, p->b_l2_w, all)); SV_TRY(make_table(p, dl2, p->b_l2_d, all));
    SV_TRY(make_table(p, wh0, p->b_h0_w, bc)); SV_TRY(make_table(p, dh0, p->b_h0_d, bc));
    SV_TRY(upload(p, fl0i, &p->f_l0.dev_img)); SV_TRY(upload(p, fl1i, &p->f_l1.dev_img));
    SV_TRY(upload(p, fl2i, &p->f_l2.dev_img)); SV_TRY(upload(p, fh0i, &p->f_h0.dev_img));
    SV_TRY(upload(p, dl0i, &p->b_l0_d.dev_img));
    SV_TRY(upload(p, dl1i, &p->b_l1_d.dev_img)); SV_TRY(upload(p, dl2i, &p->b_l2_d.dev_img));
    SV_TRY(upload(p, dh0i, &p->b_h0_d.dev_img));
    for (auto& kv : fo) SV_TRY(make_table(p, kv.second, p->f_out[kv.first], ko[kv.first]));
    for (auto& kv : wo) SV_TRY(make_table(p, kv.second, p->b_out_w[kv.first], ko[kv.first]));
    for (auto& kv : dO) SV_TRY(make_table(p, kv.second, p->b_out_d[kv.first], ko[kv.first]));
    for (auto& kv : co) SV_TRY(make_ctable(p, kv.second, p->c_out[kv.first]));
    for (auto& kv : ho) {
        HeadTable& t = p->h_out[kv.first];
        t.n = (int)kv.second.size();
        t.attr = ko[kv.first];
        SV_TRY(upload(p, kv.second, &t.dev));
    }
    return SHIVAE_OK;
}

// Zero fills and device-to-device copies inside a step.  Eager: hipMemsetAsync / hipMemcpyAsync.  Under stream capture
// (or SHIVAE_ZERO_KERNEL=1): plain store kernels, so that a captured step consists of kernel nodes only.  Measured
// reason (tools/graph_check_steps.py, ROCm 7.0): with the d_emb fill as a memset node, a replayed training step of a
// small model now and then hands the reverse scan garbage of magnitude 1e10-1e14 (every gradient behind it is wrong,
// the forward and its losses are not) -- the 77th replay of the cfg2 bench run, the 40th of another; the eager step on
// the same parameters and noise is right, and 600 replays with the fill as a kernel node are all right.
__global__ __launch_bounds__(256) void zero_fill_kernel(float4* __restrict__ p4, size_t n4, float* __restrict__ tail, int ntail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p4[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
static bool kernel_nodes_only(hipStream_t s) {
    static const int force = [] { const char* f = getenv("SHIVAE_ZERO_KERNEL"); return f ? atoi(f) : -1; }();
    if (force >= 0) return force != 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    return cap != hipStreamCaptureStatusNone;
}
__global__ __launch_bounds__(256) void zero2d_kernel(float* __restrict__ p, size_t ld, int cols, size_t rows) {
    const size_t total = rows * (size_t)cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
// columns [0, cols) of `rows` rows of leading dimension ld
static int zero2d_async(float* buf, size_t ld, int cols, size_t rows, hipStream_t s) {
    if (cols <= 0 || rows == 0) return SHIVAE_OK;
    if (!kernel_nodes_only(s)) {
        SV_CHECK_HIP(hipMemset2DAsync(buf, ld * sizeof(float), 0, (size_t)cols * sizeof(float), rows, s));
        return SHIVAE_OK;
    }
    const int grid = (int)std::min<size_t>((rows * (size_t)cols + 255) / 256, 4096);
    hipLaunchKernelGGL(zero2d_kernel, dim3(grid), dim3(256), 0, s, buf, ld, cols, rows);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}
static int copy_async(float* dst, const float* src, size_t floats, hipStream_t s) {
    if (floats == 0) return SHIVAE_OK;
    if (!kernel_nodes_only(s) || floats >= ((size_t)1 << 31)) {
        SV_CHECK_HIP(hipMemcpyAsync(dst, src, floats * sizeof(float), hipMemcpyDeviceToDevice, s));
        return SHIVAE_OK;
    }
    return launch_copy2d(src, (int)floats, 0, (int)floats, dst, (int)floats, 0, 1, s);
}
static int zero_async(float* buf, size_t floats, hipStream_t s) {
    if (floats == 0) return SHIVAE_OK;
    if (!kernel_nodes_only(s) || (reinterpret_cast<uintptr_t>(buf) & 15)) {
        SV_CHECK_HIP(hipMemsetAsync(buf, 0, floats * sizeof(float), s));
        return SHIVAE_OK;
    }
    const size_t n4 = floats / 4;
    const int grid = (int)std::min<size_t>((n4 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<float4*>(buf), n4, buf + n4 * 4, (int)(floats & 3));
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// steps of at most this many (t, b) rows fan their independent launches out (shivae_plan::side); SHIVAE_FANOUT_ROWS
static int64_t fanout_max_rows() {
    static const int64_t v = [] { const char* f = getenv("SHIVAE_FANOUT_ROWS"); return f ? atoll(f) : (int64_t)65536; }();
    return v;
}

// Fork / join of a small step's stream with the plan's side streams (shivae_plan::side): side() hands out a side stream
// (forking on first use: the side streams wait for everything issued on `s` so far), join() makes `s` wait for all of
// them.  Launches given to a side stream must not be read by anything issued on `s` before the next join.
struct Fan {
    shivae_plan* p;
    hipStream_t s;
    bool on, forked = false;
    int turn = 0;
    Fan(shivae_plan* plan, hipStream_t stream) : p(plan), s(stream) {
        static const bool fan_on = [] { const char* f = getenv("SHIVAE_FANOUT"); return !f || atoi(f) != 0; }();
        on = fan_on && p->side[0] && p->R <= fanout_max_rows() && !prof_on();
    }
    hipStream_t side() {
        if (!on) return s;
        if (!forked) {
            if (hipEventRecord(p->ev_fork, s) != hipSuccess) { on = false; return s; }
            for (int i = 0; i < shivae_plan::NSIDE; ++i)
                if (hipStreamWaitEvent(p->side[i], p->ev_fork, 0) != hipSuccess) { on = false; return s; }
            forked = true;
        }
        const int i = turn;
        turn = (turn + 1) % shivae_plan::NSIDE;
        return p->side[i];
    }
    // the side streams also wait for what has been issued on `s` since the fork
    void sync_sides() {
        if (!on) return;
        forked = false;
        turn = 0;
        (void)side();
        turn = 0;
    }
    int join() {
        if (!forked) return SHIVAE_OK;
        forked = false;
        for (int i = 0; i < shivae_plan::NSIDE; ++i) {
            SV_CHECK_HIP(hipEventRecord(p->ev_join[i], p->side[i]));
            SV_CHECK_HIP(hipStreamWaitEvent(s, p->ev_join[i], 0));
        }
        return SHIVAE_OK;
    }
};

// ------------------------------------------------------------ GEMM helpers
static int splitk_for(int M, int N, int K, int groups) {
    const int tiles = ceil_div(M, 128) * ceil_div(N, 128) * groups;
    int sk = ceil_div(1024, tiles);
    // K chunk per workgroup: long enough to amortise the tile's prologue and atomic epilogue when there are many
    // tiles; with a handful of tiles (small models: one 256x32 tile per weight gradient) the launch is a latency
    // chain of K-tiles in a few workgroups, so shorter chunks on more CUs win (cfg1: 43 -> 14 us per launch)
    // (64: cfg1 2.53 -> 2.49 ms/step, cfg2 2.10 -> 2.04 against 128; 32 is no faster)
    static const int tiny_chunk = [] { const char* f = getenv("SHIVAE_TN_MINCHUNK"); return f && atoi(f) >= 32 ? atoi(f) : 64; }();
    const int min_chunk = tiles <= 4 ? tiny_chunk : (tiles <= 32 ? 256 : 512);
    sk = std::min(sk, ceil_div(K, min_chunk));
    sk = std::min(sk, 65535 / std::max(groups, 1));
    return std::max(sk, 1);
}

// Split of a long K for a GEMM with few output tiles: the smallest split whose blocks fill whole rounds of
// the chip's 512 resident workgroups (256 CUs x 2) to >= 92 %, K chunks of at least min_chunk.
static int splitk_fill(int M, int N, int K, int min_chunk) {
    const int tiles = ceil_div(M, 128) * ceil_div(N, 128);
    const int max_sk = std::max(1, std::min(K / min_chunk, 64));
    int best = 1;
    double best_eff = 0.0;
    for (int sk = 1; sk <= max_sk; ++sk) {
        const int blocks = tiles * sk;
        const double eff = (double)blocks / ((double)ceil_div(blocks, 512) * 512.0);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = sk; }
        if (eff >= 0.92) break;
    }
    return best;
}

static bool use_addc() {
    static const bool on = [] { const char* f = getenv("SHIVAE_GEMM_ADDC"); return !f || atoi(f) != 0; }();
    return on;
}

static int gemm1(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* aux, int ldaux, int epi, hipStream_t s, const void* Bimg = nullptr,
                 int nimg = 0) {
    // A few columns past a multiple of 128 (E = 2H+K = 528 at H=256, K=16) would cost a whole extra
    // 128-wide tile column: run them as a separate skinny launch instead.
    const int rem = N % 128;
    if (N > 128 && rem > 0 && rem <= 32 && M > 64) {
        const int Nm = N - rem;
        SV_TRY(gemm1(layout, M, Nm, K, A, lda, B, ldb, C, ldc, bias, aux, ldaux, epi, s, Bimg, nimg));
        const float* B2 = layout == GEMM_NT ? B + (size_t)Nm * ldb : B + Nm;
        if (epi == 0 && K >= 4096) {
            // long-K skinny remainder: too few tiles to fill the chip -> zero the columns, split K, add atomically
            SV_TRY(zero2d_async(C + Nm, (size_t)ldc, rem, (size_t)M, s));
            GemmDesc d{layout, M, rem, K, lda, ldb, ldc, 0, EPI_ATOMIC, std::min(32, ceil_div(K, 2048)), 0, 0};
            return launch_gemm(d, nullptr, 1, GemmGroup{A, B2, C + Nm, nullptr, nullptr}, s);
        }
        return gemm1(layout, M, rem, K, A, lda, B2, ldb, C + Nm, ldc, bias ? bias + Nm : nullptr,
                     aux ? aux + Nm : nullptr, ldaux, epi, s);
    }
    GemmDesc d{layout, M, N, K, lda, ldb, ldc, ldaux, epi, 1, 0, 0};
    if (epi == EPI_ATOMIC) {
        d.splitk = splitk_for(M, N, K, 1);
        // one workgroup per output tile: accumulate with a plain 16-B read-modify-write instead of atomics
        if (d.splitk == 1 && use_addc()) d.epi = EPI_ADDC;
    }
    if (epi == 0 && K >= 8192) {
        // long K but too few output tiles to fill 256 CUs x 4 blocks: zero C, split K, accumulate atomically
        int sk = splitk_fill(M, N, K, 4096);
        // Very long K (the layer-0 input gradient, K = D*2H = 131072): the column tiles that share a row slab of A
        // stream 67 MB of it each; started together they drift apart further than an L2 holds, and the slab is
        // fetched once per column tile (measured 3.3x the operand's bytes).  Shorter K ranges keep the siblings
        // together (B=256, T=100 chunk: 27.7 -> 26.2 ms for this GEMM; SHIVAE_LONGK_CHUNK=0 restores one K range).
        static const int long_k_chunk = [] { const char* f = getenv("SHIVAE_LONGK_CHUNK"); return f ? atoi(f) : 16384; }();
        if (long_k_chunk > 0 && K >= 4 * long_k_chunk) sk = std::max(sk, std::min(64, K / long_k_chunk));
        if (sk > 1) {
            SV_TRY(zero2d_async(C, (size_t)ldc, N, (size_t)M, s));
            d.epi = EPI_ATOMIC;
            d.splitk = sk;
        }
    }
    GemmGroup g{A, B, C, bias, aux};
    if (Bimg && gemm_split_on()) {  // split mode against the weight's pre-split image (gemm_split.hip)
        d.nimg = nimg;
        if (split_nt_fits(d) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(aux)) & 15) == 0) {
            g.B = static_cast<const float*>(Bimg);
            return launch_gemm_split_nt(d, nullptr, 1, g, s);
        }
    }
    return launch_gemm(d, nullptr, 1, g, s);
}
// rag_cnt: pinned host copy of the chunk's row counts while profiling (else null); see prof_ragged
static const int* g_chunk_cnt_host = nullptr;
static int gemmg(int layout, int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int epi, const GroupTable& t,
                 hipStream_t s, int ragged = 0, bool shared_c = false, int nimg = 0) {
    if (t.n == 0) return SHIVAE_OK;
    GemmDesc d{layout, M, N, K, lda, ldb, ldc, ldaux, epi, 1, t.unaligned, t.c_aligned, ragged, nimg};
    // (shared_c: the groups add into the same C, so the adds stay atomic even with one workgroup per tile)
    if (epi == EPI_ATOMIC && !shared_c) {
        d.splitk = splitk_for(M, N, K, t.n);
        if (d.splitk == 1 && use_addc()) d.epi = EPI_ADDC;
    }
    if ((ragged & RAG_CNT) && g_chunk_cnt_host) prof_ragged(&t.attr, g_chunk_cnt_host, layout == GEMM_TN ? K : M);
    if (t.dev_img && gemm_split_on() && t.c_aligned && split_nt_fits(d))
        return launch_gemm_split_nt(d, t.dev_img, t.n, GemmGroup{}, s);
    return launch_gemm(d, t.dev, t.n, GemmGroup{}, s);
}
// streaming heads (heads.hip) with their profile record: algorithmic bytes = one read of the head input (+ one
// write of its gradient in the backward) + the parameter columns
static int heads(bool bwd, const HeadTable& t, int nout, int H, int ldin, int ldp, int rows, bool rag, hipStream_t s,
                 float obs_frac = -1.f, int chunk_rows = 0) {
    if (t.n == 0) return SHIVAE_OK;
    if (rag && g_chunk_cnt_host) prof_ragged(&t.attr, g_chunk_cnt_host, rows);
    prof_begin(bwd ? PROF_HEADS_BWD : PROF_HEADS_FWD, (double)t.n * rows * 4.0 * ((bwd ? 2.0 : 1.0) * H + nout), s);
    // (expected live rows per attribute: 1.25 x the observed fraction this plan measured, when it has one)
    // (`rows` bounds a group's live count; with narrow activation buffers it is smaller than the chunk the fraction refers to)
    const int live = (rag && obs_frac >= 0.f) ? std::max(256, (int)(1.25f * obs_frac * (chunk_rows > 0 ? chunk_rows : rows)) + 64) : 0;
    SV_TRY(launch_heads(bwd, t.dev, t.n, nout, H, ldin, ldp, rows, rag, rag, s, live));
    prof_end(s);
    return SHIVAE_OK;
}
static int colsum1(int M, int N, int ld, const float* Y, float* out, hipStream_t s) {
    return launch_colsum(M, N, ld, nullptr, 1, ColsumGroup{Y, out}, s);
}
static int colsumg(int M, int N, int ld, const ColTable& t, hipStream_t s) {
    if (t.n == 0) return SHIVAE_OK;
    return launch_colsum(M, N, ld, t.dev, t.n, ColsumGroup{}, s);
}

static ScanArgs scan_args(shivae_plan* p, const float* u, const float* eps, float temp) {
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.T = p->T; a.B = p->B; a.H = p->H; a.Z = p->Z; a.K = p->K;
    a.Hp = p->Hp; a.Zp = p->Zp; a.Kp = p->Kp; a.Ep = p->Ep; a.temp = temp;
    a.wpack = p->wsf("wpack"); a.w = p->woff;
    a.wlds_floats = (int)p->wpack_floats;
    a.xs_pre = p->wsf("xs_pre"); a.ex_pre = p->wsf("ex_pre"); a.u = u; a.eps = eps; a.seed = p->noise_seed;
    a.emb = p->wsf("emb"); a.as_ = p->wsf("as"); a.e1 = p->wsf("e1"); a.e2 = p->wsf("e2");
    a.logits = p->wsf("logits"); a.probs = p->wsf("probs"); a.mu = p->wsf("mu"); a.sd = p->wsf("sd");
    a.z = p->wsf("z"); a.gates = p->wsf("gates"); a.cnew = p->wsf("cnew"); a.h_last = nullptr;
    a.d_emb = p->wsf("d_emb"); a.dmu_ext = p->wsf("dmu_ext"); a.dsd_ext = p->wsf("dsd_ext"); a.dlg_kl = p->wsf("dlg_kl");
    a.d_gates = p->wsf("d_gates"); a.d_phiz_pre = p->wsf("d_phiz_pre"); a.d_mu = p->wsf("d_mu");
    a.d_sd_pre = p->wsf("d_sd_pre"); a.d_e2_pre = p->wsf("d_e2_pre"); a.d_e1_pre = p->wsf("d_e1_pre");
    a.d_lg_pre = p->wsf("d_lg_pre"); a.d_as_pre = p->wsf("d_as_pre");
    return a;
}

// ------------------------------------------------------------ the step, phase by phase
// Each phase is also an entry point of the C ABI (shivae_scan_fwd, shivae_prior_kl, shivae_decoder_nll,
// shivae_scan_bwd); shivae_elbo_step is their composition.
static double scan_flops(const shivae_plan* p) {
    // serial-chain flops per (t,b): SURVEY.md 8d (k2 h-part, k2 out, k4 s+h parts, enc.2, mean/std, phi_z, LSTM)
    const double H = p->H, Z = p->Z, K = p->K;
    return 2.0 * (H * H + H * K + (H + K) * H + H * H + 2.0 * H * Z + Z * H + 8.0 * H * H) * (double)p->R;
}

// Phase 1: packed scan weights, the x-only encoder GEMMs over all rows, the forward scan
// (encoders.py:70-98, shivae.py:102-143).
static int phase_scan_fwd(shivae_plan* p, const float* x, const float* u, const float* eps, float temp, hipStream_t s) {
    const int H = p->H, X = p->X, E = p->E, Hp = p->Hp;
    const int R = (int)p->R;
    const Tensor* g = p->glob;
    const int NT = GEMM_NT;
    // scan weights in MFMA operand order (parameters change every optimiser step)
    SV_TRY(launch_pack(p->wjobs_dev, p->n_wjobs, p->bjobs_dev, p->n_bjobs, p->wsf("wpack"), s));
    // 1. h-independent encoder parts, batched over all T*B rows (encoders.py:82, x/phi_x halves of :83,:91)
    float *p1 = p->wsf("p1"), *phx = p->wsf("phx"), *xs_pre = p->wsf("xs_pre"), *ex_pre = p->wsf("ex_pre");
    SV_TRY(gemm1(NT, R, H, X, x, X, p->prm(g[SHIVAE_P_PHIX0_W]), X, p1, Hp, p->prm(g[SHIVAE_P_PHIX0_B]), nullptr, 0,
                 EPI_BIAS | EPI_RELU, s));
    SV_TRY(gemm1(NT, R, H, H, p1, Hp, p->prm(g[SHIVAE_P_PHIX2_W]), H, phx, Hp, p->prm(g[SHIVAE_P_PHIX2_B]), nullptr, 0,
                 EPI_BIAS | EPI_RELU, s));
    SV_TRY(gemm1(NT, R, H, X, x, X, p->prm(g[SHIVAE_P_XS0_W]), X + H, xs_pre, Hp, p->prm(g[SHIVAE_P_XS0_B]), nullptr, 0,
                 EPI_BIAS, s));
    SV_TRY(gemm1(NT, R, H, H, phx, Hp, p->prm(g[SHIVAE_P_E0_W]), E, ex_pre, Hp, p->prm(g[SHIVAE_P_E0_B]), nullptr, 0,
                 EPI_BIAS, s));

    // the serial chain
    ScanArgs sa = scan_args(p, u, eps, temp);
    prof_begin(PROF_SCAN_FWD, scan_flops(p), s);
    SV_TRY(launch_scan(sa, false, s));
    prof_end(s);
    return SHIVAE_OK;
}

// Phase 2: prior network + both KL terms (+ their gradients and the prior network's backward) over ALL rows
// (shivae.py:124-127, base_missvae.py:287-343).  With gradients it starts the d_emb records of this step: zero,
// plus the prior network's input gradient; the decoder phase adds its part.
static int phase_prior_kl(shivae_plan* p, float beta, float inv_batch, bool grad, hipStream_t s) {
    const int H = p->H, Z = p->Z, K = p->K;
    const int Hp = p->Hp, Zp = p->Zp, Kp = p->Kp, Ep = p->Ep;
    const int R = (int)p->R;
    const Tensor* g = p->glob;
    const int NT = GEMM_NT, NN = GEMM_NN, TN = GEMM_TN;
    float *emb = p->wsf("emb"), *d_emb = p->wsf("d_emb"), *rowloss = p->wsf("rowloss");
    if (grad) SV_TRY(zero_async(d_emb, (size_t)R * Ep, s));
    Fan fan(p, s);

    const float *Wp0 = p->prm(g[SHIVAE_P_PRIOR0_W]), *Wpm = p->prm(g[SHIVAE_P_PRIORMEAN_W]),
                *Wps = p->prm(g[SHIVAE_P_PRIORSTD_W]);
    const float kz = beta * inv_batch;
    // Blocks of <= 4M rows keep the GEMM grids legal.  Scratch is borrowed from record buffers that are only
    // written after this point (by the backward scan / phi_x backward).
    {
        float *prc_all = p->wsf("d_phx_pre"), *dprc_all = p->wsf("d_p1_pre");
        float *dmup_all = p->wsf("d_mu"), *dsdp_all = p->wsf("d_sd_pre");
        float *mup = p->wsf("mup"), *sdp = p->wsf("sdp");
        const int PB = 1 << 22;
        for (int r0 = 0; r0 < R; r0 += PB) {
            const int rb = std::min(PB, R - r0);
            const float *emb_s = emb + (size_t)r0 * Ep + Hp, *emb_h = emb + (size_t)r0 * Ep + Hp + Kp;
            float *prc = prc_all + (size_t)r0 * Hp, *dprc = dprc_all + (size_t)r0 * Hp;
            float *mupc = mup + (size_t)r0 * Zp, *sdpc = sdp + (size_t)r0 * Zp;
            float *dmupc = dmup_all + (size_t)r0 * Zp, *dsdpc = dsdp_all + (size_t)r0 * Zp;
            // prior.0 takes cat[h, s] (shivae.py:124): columns [0,H) of its weight act on h, [H,H+K) on s
            SV_TRY(gemm1(NT, rb, H, H, emb_h, Ep, Wp0, H + K, prc, Hp, nullptr, nullptr, 0, 0, s));
            SV_TRY(gemm1(NT, rb, H, K, emb_s, Ep, Wp0 + H, H + K, prc, Hp, p->prm(g[SHIVAE_P_PRIOR0_B]), nullptr, 0,
                         EPI_BIAS | EPI_ADDC | EPI_RELU, s));
            SV_TRY(gemm1(NT, rb, Z, H, prc, Hp, Wpm, H, mupc, Zp, p->prm(g[SHIVAE_P_PRIORMEAN_B]), nullptr, 0, EPI_BIAS, s));
            SV_TRY(gemm1(NT, rb, Z, H, prc, Hp, Wps, H, sdpc, Zp, p->prm(g[SHIVAE_P_PRIORSTD_B]), nullptr, 0,
                         EPI_BIAS | EPI_SOFTPLUS, s));
            // algorithmic bytes per row: mu, std (q and p) 16Z + probs 4K; the five gradients double it
            prof_begin(PROF_KL, (double)rb * (16.0 * Z + 4.0 * K) * (grad ? 2.0 : 1.0), s);
            SV_TRY(launch_kl(p->wsf("mu") + (size_t)r0 * Zp, p->wsf("sd") + (size_t)r0 * Zp, mupc, sdpc, Zp,
                             p->wsf("probs") + (size_t)r0 * Kp, Kp, grad ? p->wsf("dmu_ext") + (size_t)r0 * Zp : nullptr,
                             grad ? p->wsf("dsd_ext") + (size_t)r0 * Zp : nullptr, grad ? dmupc : nullptr,
                             grad ? dsdpc : nullptr, grad ? p->wsf("dlg_kl") + (size_t)r0 * Kp : nullptr, rb, Z, K, kz,
                             rowloss + (size_t)r0 * 6, s));
            prof_end(s);
            if (!grad) continue;
            // Small steps (Fan): the weight / bias gradients only read what is already there and go to the side streams;
            // the step's stream carries the chain dmu, dsd -> dprc -> d_emb.
            SV_TRY(gemm1(TN, Z, H, rb, dmupc, Zp, prc, Hp, p->grd(g[SHIVAE_P_PRIORMEAN_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, fan.side()));
            SV_TRY(gemm1(TN, Z, H, rb, dsdpc, Zp, prc, Hp, p->grd(g[SHIVAE_P_PRIORSTD_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, fan.side()));
            SV_TRY(colsum1(rb, Z, Zp, dmupc, p->grd(g[SHIVAE_P_PRIORMEAN_B]), fan.side()));
            SV_TRY(colsum1(rb, Z, Zp, dsdpc, p->grd(g[SHIVAE_P_PRIORSTD_B]), fan.side()));
            SV_TRY(gemm1(NN, rb, H, Z, dmupc, Zp, Wpm, H, dprc, Hp, nullptr, nullptr, 0, 0, s));
            SV_TRY(gemm1(NN, rb, H, Z, dsdpc, Zp, Wps, H, dprc, Hp, nullptr, prc, Hp, EPI_ADDC | EPI_MASK, s));
            fan.sync_sides();  // (dprc is ready for them, too)
            float* gWp0 = p->grd(g[SHIVAE_P_PRIOR0_W]);
            SV_TRY(gemm1(TN, H, H, rb, dprc, Hp, emb_h, Ep, gWp0, H + K, nullptr, nullptr, 0, EPI_ATOMIC, fan.side()));
            SV_TRY(gemm1(TN, H, K, rb, dprc, Hp, emb_s, Ep, gWp0 + H, H + K, nullptr, nullptr, 0, EPI_ATOMIC, fan.side()));
            SV_TRY(colsum1(rb, H, Hp, dprc, p->grd(g[SHIVAE_P_PRIOR0_B]), fan.side()));
            // into the (zeroed) d_emb records; the decoder chunks add their part below
            float* de = d_emb + (size_t)r0 * Ep;
            SV_TRY(gemm1(NN, rb, H, H, dprc, Hp, Wp0, H + K, de + Hp + Kp, Ep, nullptr, nullptr, 0, 0, s));
            SV_TRY(gemm1(NN, rb, K, H, dprc, Hp, Wp0 + H, H + K, de + Hp, Ep, nullptr, nullptr, 0, 0, s));
        }
    }
    // (the side launches read scratch borrowed from records that the reverse scan writes: done before the phase ends)
    SV_TRY(fan.join());
    return SHIVAE_OK;
}

// Phase 3: the decoder and its likelihood, in row chunks (decoders.py:109-176, base_missvae.py:233-285), each
// chunk's backward right behind its forward (activations only ever exist for one chunk); input gradients are
// ADDED to the d_emb records, weight gradients to the bound gradient buffer.
static int phase_decoder(shivae_plan* p, const float* x, const uint8_t* mask, float inv_batch, int flags, hipStream_t s) {
    const bool grad = !(flags & SHIVAE_STEP_FORWARD_ONLY);
    const bool keep = flags & SHIVAE_STEP_KEEP_PARAMS;
    const int H = p->H, K = p->K, X = p->X, D = p->D, E = p->E, P = p->P;
    const int Hp = p->Hp, Kp = p->Kp, Ep = p->Ep, E4 = p->E4;
    const int R = (int)p->R;
    const Tensor* g = p->glob;
    const int NT = GEMM_NT, NN = GEMM_NN, TN = GEMM_TN;
    float *emb = p->wsf("emb"), *d_emb = p->wsf("d_emb"), *rowloss = p->wsf("rowloss");
    const bool split = gemm_split_on();
    if (split) {  // the step's weights as three bf16 planes, once (gemm_split.hip)
        SV_TRY(launch_presplit(p->prm(g[SHIVAE_P_DEC0_W]), 2 * H, E, E, D, (size_t)2 * H * E, p->wsf("wi_f0"), s));
        SV_TRY(launch_presplit(p->prm(g[SHIVAE_P_DEC2_W]), H, 2 * H, 2 * H, D, (size_t)H * 2 * H, p->wsf("wi_f2"), s));
        SV_TRY(launch_presplit(p->prm(g[SHIVAE_P_DEC4_W]), H, H, H, D, (size_t)H * H, p->wsf("wi_f4"), s));
        if (p->nbc > 0) SV_TRY(launch_presplit(p->prm(g[SHIVAE_P_HEAD0_W]), H, H, H, p->nbc, (size_t)H * H, p->wsf("wi_fh0"), s));
    }
    if (grad) {  // K-contiguous copies of the decoder weights for the input-gradient GEMMs
        SV_TRY(launch_transpose(p->prm(g[SHIVAE_P_DEC0_W]), p->wsf("wt0"), D * 2 * H, E, 1, 0, 0, s));
        SV_TRY(launch_transpose(p->prm(g[SHIVAE_P_DEC2_W]), p->wsf("wt2"), H, 2 * H, D, (size_t)H * 2 * H, (size_t)H * 2 * H, s));
        SV_TRY(launch_transpose(p->prm(g[SHIVAE_P_DEC4_W]), p->wsf("wt4"), H, H, D, (size_t)H * H, (size_t)H * H, s));
        if (p->nbc > 0)
            SV_TRY(launch_transpose(p->prm(g[SHIVAE_P_HEAD0_W]), p->wsf("wth0"), H, H, p->nbc, (size_t)H * H, (size_t)H * H, s));
        if (split) {
            SV_TRY(launch_presplit(p->wsf("wt0"), E, D * 2 * H, D * 2 * H, 1, 0, p->wsf("wi_t0"), s));
            SV_TRY(launch_presplit(p->wsf("wt2"), 2 * H, H, H, D, (size_t)2 * H * H, p->wsf("wi_t2"), s));
            SV_TRY(launch_presplit(p->wsf("wt4"), H, H, H, D, (size_t)H * H, p->wsf("wi_t4"), s));
            if (p->nbc > 0) SV_TRY(launch_presplit(p->wsf("wth0"), H, H, H, p->nbc, (size_t)H * H, p->wsf("wi_th0"), s));
        }
    }
    float *embc = p->wsf("embc"), *dembc = p->wsf("dembc");
    float *a0 = p->wsf("a0"), *pbuf = p->wsf("pbuf");
    float *da0 = p->wsf("da0");
    const int ldA0 = p->ldA0, ldA1 = p->ldA1, ldHH = p->ldHH;
    const bool rag = grad && !keep && p->compact;
    const int rcnt = rag ? RAG_CNT : 0;
    int* rowidx = reinterpret_cast<int*>(p->wsf("rowidx"));
    int* rowcnt = reinterpret_cast<int*>(p->wsf("rowcnt"));
    int* rowrank = reinterpret_cast<int*>(p->wsf("rowrank"));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);  // (no host wait inside a stream capture)
    // Narrow activation buffers (Ra < Rc): only the compact forms fit them, and a chunk's counts are checked on the
    // host before its launches go out.  Dense passes, and captures (no host wait), step Ra rows at a time.
    const bool narrow = p->Ra < p->Rc;
    const int step_rows = (rag && cap == hipStreamCaptureStatusNone) ? p->Rc : p->Ra;
    auto observed_rows = [&](int r0, int rc) {
        return launch_observed_rows(p->attrs_dev, D, mask + (size_t)r0 * X, X, rc, rowidx, p->Rc, rowcnt, s,
                                    p->l0_prod && p->l0_gather ? rowrank : nullptr);
    };
    // one row chunk [r0, r0 + rc); lists_ready: observed_rows(r0, rc) has already run
    auto chunk = [&](int r0, int rc, bool lists_ready) -> int {
        // rr: row extent of everything that is indexed compactly (bounds every group's live count)
        const int rr = rag ? std::min(rc, p->Ra) : rc;
        SV_TRY(launch_compact_emb(emb + (size_t)r0 * Ep, Ep, Hp, Kp, H, K, embc, E4, rc, s));
        if (rag) {
            // Mask-aware decoder: attribute d's network only runs on the rows in which d is observed (their
            // likelihood terms are the only ones in the loss, base_missvae.py:271-285).  Activations of group d
            // are stored compactly -- row i of its column block = its i-th observed row of the chunk.
            if (!lists_ready) SV_TRY(observed_rows(r0, rc));
            g_chunk_cnt_host = nullptr;
            if (prof_on() && p->cnt_host) {
                // (a ring of 4 steps' worth of chunks: a profile session longer than that re-uses slots, and its
                // older ragged records are then scaled by a later chunk's counts)
                int* blk = p->cnt_host + (size_t)(p->cnt_next++ % p->cnt_slots) * D;
                SV_CHECK_HIP(hipMemcpyAsync(blk, rowcnt, (size_t)D * sizeof(int), hipMemcpyDeviceToHost, s));
                g_chunk_cnt_host = blk;
            }
            // (no host wait inside a stream capture: the dense form is taken there)
            if (p->demb_mode == 2 && p->obs_frac < 0.f && cap == hipStreamCaptureStatusNone) {
                // which form of the layer-0 input gradient pays depends on how much is observed (see below): measured
                // once per plan, on its first training chunk (the only host wait in a step, and only in that one)
                std::vector<int> cnt(D);
                SV_CHECK_HIP(hipMemcpyAsync(cnt.data(), rowcnt, (size_t)D * sizeof(int), hipMemcpyDeviceToHost, s));
                SV_CHECK_HIP(hipStreamSynchronize(s));
                double tot = 0;
                for (int c : cnt) tot += c;
                p->obs_frac = (float)(tot / ((double)D * rc));
            }
            SV_TRY(gemmg(NT, rr, 2 * H, E, E4, E, ldA0, 0, EPI_BIAS | EPI_RELU, p->f_l0, s, RAG_CNT | RAG_AIDX));
        } else if (split && p->f_l0.dev_img && (2 * H) % 256 == 0 && E % 4 == 0 && rc > 64) {
            // dense rows, split mode: the same grouped launch over all rows (the weight images are per attribute)
            SV_TRY(gemmg(NT, rc, 2 * H, E, E4, E, ldA0, 0, EPI_BIAS | EPI_RELU, p->f_l0, s, 0));
        } else {
            // decoder forward: layer 0 of all attributes is one GEMM (shared input, stacked weights)
            SV_TRY(gemm1(NT, rc, D * 2 * H, E, embc, E4, p->prm(g[SHIVAE_P_DEC0_W]), E, a0, ldA0, p->prm(g[SHIVAE_P_DEC0_B]),
                         nullptr, 0, EPI_BIAS | EPI_RELU, s));
        }
        SV_TRY(gemmg(NT, rr, H, 2 * H, ldA0, 2 * H, ldA1, 0, EPI_BIAS | EPI_RELU, p->f_l1, s, rcnt));
        SV_TRY(gemmg(NT, rr, H, H, ldA1, H, ldA1, 0, EPI_BIAS | EPI_RELU, p->f_l2, s, rcnt));
        SV_TRY(gemmg(NT, rr, H, H, ldA1, H, ldHH, 0, EPI_BIAS | EPI_RELU, p->f_h0, s, rcnt));
        for (auto& kv : p->f_out) {
            const int key = kv.first, nout = key < 0 ? -key : key;
            // (ragged: stored through the row list, so pbuf stays in chunk-row order; the rows of missing entries
            // keep finite leftovers -- zeros or older parameters -- which the likelihood kernel masks out)
            const int ldin = key < 0 ? ldA1 : ldHH;
            if (p->stream_heads && head_kernels_fit(H, nout, ldin, P)) {
                SV_TRY(heads(false, p->h_out[key], nout, H, ldin, P, rr, rag, s, p->obs_frac, rc));
                continue;
            }
            SV_TRY(gemmg(NT, rr, nout, H, ldin, H, P, 0, EPI_BIAS, kv.second, s, rag ? RAG_CNT | RAG_CIDX : 0));
        }
        if (keep) SV_TRY(copy_async(p->wsf("pfull") + (size_t)r0 * P, pbuf, (size_t)rc * P, s));
        // algorithmic bytes per row: x (4X) + mask (X) + parameters (4P), + the gradient write (4P) in training
        prof_begin(PROF_NLL, (double)rc * (5.0 * X + 4.0 * P * (grad ? 2.0 : 1.0)), s);
        SV_TRY(launch_heter_nll(p->attrs_dev, D, p->learn_std, x + (size_t)r0 * X, X, mask + (size_t)r0 * X, X, pbuf, P,
                                grad ? pbuf : nullptr, P, rc, inv_batch, rowloss + (size_t)r0 * 6, X, P, p->nll_group, p->nll_xspan, p->nll_pspan, s));
        prof_end(s);
        if (!grad) return SHIVAE_OK;

        // ---- decoder backward (pbuf now holds d nll / d head outputs) ----
        for (auto& kv : p->b_out_w) {
            const int key = kv.first, nout = key < 0 ? -key : key, ldin = key < 0 ? ldA1 : ldHH;
            if (p->stream_heads && head_kernels_fit(H, nout, ldin, P)) {
                // one streaming pass: input gradient (ReLU-gated), weight and bias gradients, and the bias
                // gradient of the layer that produced the head input (heads.hip)
                SV_TRY(heads(true, p->h_out[key], nout, H, ldin, P, rr, rag, s, p->obs_frac, rc));
                continue;
            }
            // (ragged: the gradient rows of the observed entries are read through the row list; the dense column
            // sums below are unaffected, the gradient is zero where the entry is missing)
            SV_TRY(gemmg(TN, nout, H, rr, P, ldin, H, 0, EPI_ATOMIC, kv.second, s, rag ? RAG_CNT | RAG_AIDX : 0));
            SV_TRY(colsumg(rc, nout, P, p->c_out[key], s));
            // d(head input) = dP * Wout, gated by the head input's ReLU; its column sums are the bias gradient
            // of the layer that produced the head input (EPI_COLSUM, here and in the trunk below)
            SV_TRY(gemmg(NN, rr, H, nout, P, H, ldin, ldin, EPI_MASK | EPI_COLSUM, p->b_out_d[key], s,
                         rag ? RAG_CNT | RAG_AIDX : 0));
        }
        // (small steps, Fan: a layer's weight gradient reads what the chain has produced so far and nothing reads it --
        // it goes to a side stream while the step's stream carries the input gradients dhh -> da2 -> da1 -> da0)
        Fan fan(p, s);
        SV_TRY(gemmg(TN, H, H, rr, ldHH, ldA1, H, 0, EPI_ATOMIC, p->b_h0_w, fan.side(), rcnt));
        SV_TRY(gemmg(NT, rr, H, H, ldHH, H, ldA1, ldA1, EPI_MASK | EPI_COLSUM, p->b_h0_d, s, rcnt));
        fan.sync_sides();
        SV_TRY(gemmg(TN, H, H, rr, ldA1, ldA1, H, 0, EPI_ATOMIC, p->b_l2_w, fan.side(), rcnt));
        SV_TRY(gemmg(NT, rr, H, H, ldA1, H, ldA1, ldA1, EPI_MASK | EPI_COLSUM, p->b_l2_d, s, rcnt));
        fan.sync_sides();
        SV_TRY(gemmg(TN, H, 2 * H, rr, ldA1, ldA0, 2 * H, 0, EPI_ATOMIC, p->b_l1_w, fan.side(), rcnt));
        // Layer 0 backward, input gradient (the sum over attributes of decoders.py:127's shared input) in two forms.
        // "compact": da0 stays compact like the activations and one grouped GEMM over the observed rows of every
        // attribute ADDS its tiles into the chunk's rows: flops follow the observed fraction f, but every product is
        // an fp32 atomic (measured 80 TFLOP/s of live flops).  "dense": da0 goes back to chunk rows (missing blocks
        // cleared first) and one GEMM of K = D*2H sums over the attributes inside its K loop at 121 TFLOP/s,
        // multiplying the zeros of the missing blocks.  Measured step times (B=256, T=100): f = 0.72 (cfg3) 159.4 ms
        // compact / 157.1 dense; f = 0.11 (cfg5) 81.4 / 131.7.  auto: compact below f = 0.6.
        // (narrow activation buffers have no room for da0 in chunk rows: compact whatever the fraction)
        const bool cd = rag && (narrow || p->demb_mode == 1 || (p->demb_mode == 2 && p->obs_frac >= 0.f && p->obs_frac < 0.6f));
        if (rag && !cd) {
            if ((2 * H) % 4 == 0 && ldA0 % 4 == 0)
                SV_TRY(launch_zero_missing_blocks(p->attrs_dev, D, mask + (size_t)r0 * X, X, rc, da0, ldA0, 2 * H, s));
            else
                SV_TRY(zero_async(da0, (size_t)rc * ldA0, s));
        }
        SV_TRY(gemmg(NT, rr, 2 * H, H, ldA1, H, ldA0, ldA0, EPI_MASK | EPI_COLSUM, p->b_l1_d, s,
                     rag ? (cd ? RAG_CNT : RAG_CNT | RAG_CIDX) : 0));
        if (rag) {
            const int rl = cd ? RAG_CNT | RAG_BIDX : RAG_CNT | RAG_AIDX | RAG_BIDX, nm = p->l0_nmain;
            SV_TRY(gemmg(TN, 2 * H, nm, rr, ldA0, E4, E, 0, EPI_ATOMIC, p->b_l0_w, s, rl));
            if (nm < E) SV_TRY(gemmg(TN, 2 * H, E - nm, rr, ldA0, E4, E, 0, EPI_ATOMIC, p->b_l0_wr, s, rl));
        } else {
            SV_TRY(gemm1(TN, D * 2 * H, E, rc, da0, ldA0, embc, E4, p->grd(g[SHIVAE_P_DEC0_W]), E, nullptr, nullptr, 0,
                         EPI_ATOMIC, s));
        }
        SV_TRY(fan.join());  // (what follows overwrites the layer-1/2 buffers, or belongs to the next chunk)
        if (cd && p->l0_prod && p->l0_gather) {
            // Compact form without atomics: each attribute's product rows are stored where the dead layer-1/2 buffers
            // were (plain stores), then one pass sums, per chunk row, the rows of the attributes observed in it.
            const int nm = p->l0_nmain, ldp = D * E4;
            static const bool l0d_img = [] { const char* f = getenv("SHIVAE_L0D_IMG"); return !f || atoi(f) != 0; }();
            GemmDesc probe{NT, rr, nm, 2 * H, ldA0, 0, ldp, 0, 0, 1, 0, 1, RAG_CNT, E};
            if (split && l0d_img && p->b_l0_p.dev_img && split_nt_fits(probe))
                SV_TRY(gemmg(NT, rr, nm, 2 * H, ldA0, 0, ldp, 0, 0, p->b_l0_p, s, RAG_CNT, false, E));
            else
                SV_TRY(gemmg(NN, rr, nm, 2 * H, ldA0, E, ldp, 0, 0, p->b_l0_p, s, RAG_CNT));
            if (nm < E) SV_TRY(gemmg(NN, rr, E - nm, 2 * H, ldA0, E, ldp, 0, 0, p->b_l0_pr, s, RAG_CNT));
            SV_TRY(launch_l0_gather(rowrank, D, p->Rc, p->l0_prod, (size_t)ldp, E4, E, dembc, E4, rc, s));
        } else if (cd) {
            const int nm = p->l0_nmain;
            SV_TRY(zero_async(dembc, (size_t)rc * E4, s));
            // (split mode: as an NT launch against the attribute's K range of the transposed stack's weight image)
            static const bool l0d_img = [] { const char* f = getenv("SHIVAE_L0D_IMG"); return !f || atoi(f) != 0; }();
            GemmDesc probe{NT, rr, nm, 2 * H, ldA0, 0, E4, 0, EPI_ATOMIC, 1, 0, 1, RAG_CNT | RAG_CIDX, E};
            if (split && l0d_img && p->b_l0_d.dev_img && split_nt_fits(probe))
                SV_TRY(gemmg(NT, rr, nm, 2 * H, ldA0, 0, E4, 0, EPI_ATOMIC, p->b_l0_d, s, RAG_CNT | RAG_CIDX, true, E));
            else
                SV_TRY(gemmg(NN, rr, nm, 2 * H, ldA0, E, E4, 0, EPI_ATOMIC, p->b_l0_d, s, RAG_CNT | RAG_CIDX, true));
            if (nm < E) SV_TRY(gemmg(NN, rr, E - nm, 2 * H, ldA0, E, E4, 0, EPI_ATOMIC, p->b_l0_dr, s, RAG_CNT | RAG_CIDX, true));
        } else {
            SV_TRY(gemm1(NT, rc, E, D * 2 * H, da0, ldA0, p->wsf("wt0"), D * 2 * H, dembc, E4, nullptr, nullptr, 0, 0, s,
                         p->wsf("wi_t0"), E));
        }

        SV_TRY(launch_expand_add_emb(dembc, E4, H, K, d_emb + (size_t)r0 * Ep, Ep, Hp, Kp, rc, s));
        // (SHIVAE_SYNC_CHUNKS=1: the host waits for each row chunk.  For runs under `rocprofv3 --pmc`, whose
        // per-dispatch buffers do not survive thousands of launches enqueued ahead of the GPU -- DESIGN.md s.4)
        static const bool sync_chunks = [] { const char* f = getenv("SHIVAE_SYNC_CHUNKS"); return f && atoi(f) != 0; }();
        if (sync_chunks) SV_CHECK_HIP(hipStreamSynchronize(s));
        return SHIVAE_OK;
    };
    for (int r0 = 0; r0 < R; r0 += step_rows) {
        const int rc = std::min(step_rows, R - r0);
        if (rag && rc > p->Ra) {
            // more chunk rows than activation rows: the lists first, and the largest count checked on the host (one
            // wait per chunk -- these chunks are several times the usual size, so it is a small fraction of one)
            SV_TRY(observed_rows(r0, rc));
            SV_CHECK_HIP(hipMemcpyAsync(p->cnt_guard, rowcnt, (size_t)D * sizeof(int), hipMemcpyDeviceToHost, s));
            SV_CHECK_HIP(hipStreamSynchronize(s));
            int mx = 0;
            for (int d = 0; d < D; ++d) mx = std::max(mx, p->cnt_guard[d]);
            if (mx > p->Ra) {
                // an attribute is observed in more rows of this chunk than the buffers hold: pieces of Ra rows always fit
                p->n_overflow++;
                for (int q0 = r0; q0 < r0 + rc; q0 += p->Ra) SV_TRY(chunk(q0, std::min(p->Ra, r0 + rc - q0), false));
                continue;
            }
            SV_TRY(chunk(r0, rc, true));
        } else {
            SV_TRY(chunk(r0, rc, false));
        }
    }
    g_chunk_cnt_host = nullptr;
    return SHIVAE_OK;
}

// Phase 4: reverse-time scan over the d_emb / dmu_ext / dsd_ext / dlg_kl records (BPTT over the whole T:
// shivae.py:102-137 never detaches), then the weight gradients of the cell and the phi_x backward.
static int phase_scan_bwd(shivae_plan* p, const float* x, const float* u, const float* eps, float temp, hipStream_t s) {
    const int H = p->H, Z = p->Z, K = p->K, X = p->X, E = p->E;
    const int Hp = p->Hp, Zp = p->Zp, Kp = p->Kp, Ep = p->Ep;
    const int R = (int)p->R;
    const Tensor* g = p->glob;
    const int NN = GEMM_NN, TN = GEMM_TN;
    float *emb = p->wsf("emb"), *p1 = p->wsf("p1"), *phx = p->wsf("phx");
    ScanArgs sa = scan_args(p, u, eps, temp);
    {
        // (the reverse chain multiplies by the transposes of the forward chain's matrices: the same MFMA work.  The
        // other half of "backward = 2 x forward", the weight gradients, are the batched TN GEMMs below.)
        prof_begin(PROF_SCAN_BWD, scan_flops(p), s);
        SV_TRY(launch_scan(sa, true, s));
        prof_end(s);
        // 5. weight gradients of the cell, batched over all rows (K dim = rows, split-K + atomics).  Every launch from
        //    here to the end of the phase only reads the records and writes a gradient tensor of its own (the phi_x
        //    chain excepted, which stays on one stream): small steps deal them round the side streams.
        static const bool fan_on = [] { const char* f = getenv("SHIVAE_FANOUT"); return !f || atoi(f) != 0; }();
        const bool fan = fan_on && p->side[0] && p->R <= fanout_max_rows() && !prof_on();
        int turn = 0;
        auto S = [&]() -> hipStream_t {
            if (!fan) return s;
            const int i = turn;
            turn = (turn + 1) % (shivae_plan::NSIDE + 1);
            return i == 0 ? s : p->side[i - 1];
        };
        if (fan) {
            SV_CHECK_HIP(hipEventRecord(p->ev_fork, s));
            for (int i = 0; i < shivae_plan::NSIDE; ++i) SV_CHECK_HIP(hipStreamWaitEvent(p->side[i], p->ev_fork, 0));
        }
        const float *phiz = emb, *s_t = emb + Hp, *hprev = emb + Hp + Kp;
        float *d_gates = p->wsf("d_gates"), *d_phz = p->wsf("d_phiz_pre"), *d_mu = p->wsf("d_mu"),
              *d_sd = p->wsf("d_sd_pre"), *d_e2 = p->wsf("d_e2_pre"), *d_e1 = p->wsf("d_e1_pre"),
              *d_lg = p->wsf("d_lg_pre"), *d_as = p->wsf("d_as_pre");
        for (int gt = 0; gt < 4; ++gt) {  // gate blocks are padded H -> Hp in the records
            const float* dg = d_gates + gt * Hp;
            SV_TRY(gemm1(TN, H, H, R, dg, 4 * Hp, phiz, Ep, p->grd(g[SHIVAE_P_RNN_WIH]) + (size_t)gt * H * H, H, nullptr,
                         nullptr, 0, EPI_ATOMIC, S()));
            SV_TRY(gemm1(TN, H, H, R, dg, 4 * Hp, hprev, Ep, p->grd(g[SHIVAE_P_RNN_WHH]) + (size_t)gt * H * H, H, nullptr,
                         nullptr, 0, EPI_ATOMIC, S()));
            SV_TRY(colsum1(R, H, 4 * Hp, dg, p->grd(g[SHIVAE_P_RNN_BIH]) + gt * H, S()));
            SV_TRY(colsum1(R, H, 4 * Hp, dg, p->grd(g[SHIVAE_P_RNN_BHH]) + gt * H, S()));
        }
        SV_TRY(gemm1(TN, H, Z, R, d_phz, Hp, p->wsf("z"), Zp, p->grd(g[SHIVAE_P_PHIZ_W]), Z, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, H, Hp, d_phz, p->grd(g[SHIVAE_P_PHIZ_B]), S()));
        SV_TRY(gemm1(TN, Z, H, R, d_mu, Zp, p->wsf("e2"), Hp, p->grd(g[SHIVAE_P_ENCMEAN_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, Z, Zp, d_mu, p->grd(g[SHIVAE_P_ENCMEAN_B]), S()));
        SV_TRY(gemm1(TN, Z, H, R, d_sd, Zp, p->wsf("e2"), Hp, p->grd(g[SHIVAE_P_ENCSTD_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, Z, Zp, d_sd, p->grd(g[SHIVAE_P_ENCSTD_B]), S()));
        SV_TRY(gemm1(TN, H, H, R, d_e2, Hp, p->wsf("e1"), Hp, p->grd(g[SHIVAE_P_E2_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, H, Hp, d_e2, p->grd(g[SHIVAE_P_E2_B]), S()));
        float* gE0 = p->grd(g[SHIVAE_P_E0_W]);
        SV_TRY(gemm1(TN, H, H, R, d_e1, Hp, phx, Hp, gE0, E, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(gemm1(TN, H, K, R, d_e1, Hp, s_t, Ep, gE0 + H, E, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(gemm1(TN, H, H, R, d_e1, Hp, hprev, Ep, gE0 + H + K, E, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, H, Hp, d_e1, p->grd(g[SHIVAE_P_E0_B]), S()));
        SV_TRY(gemm1(TN, K, H, R, d_lg, Kp, p->wsf("as"), Hp, p->grd(g[SHIVAE_P_XS2_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, K, Kp, d_lg, p->grd(g[SHIVAE_P_XS2_B]), S()));
        float* gXS0 = p->grd(g[SHIVAE_P_XS0_W]);
        SV_TRY(gemm1(TN, H, X, R, d_as, Hp, x, X, gXS0, X + H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(gemm1(TN, H, H, R, d_as, Hp, hprev, Ep, gXS0 + X, X + H, nullptr, nullptr, 0, EPI_ATOMIC, S()));
        SV_TRY(colsum1(R, H, Hp, d_as, p->grd(g[SHIVAE_P_XS0_B]), S()));
        // 6. phi_x backward (encoders.py:47-51)
        float *d_phx = p->wsf("d_phx_pre"), *d_p1 = p->wsf("d_p1_pre");
        const hipStream_t sc = S();
        SV_TRY(gemm1(NN, R, H, H, d_e1, Hp, p->prm(g[SHIVAE_P_E0_W]), E, d_phx, Hp, nullptr, phx, Hp, EPI_MASK, sc));
        SV_TRY(gemm1(TN, H, H, R, d_phx, Hp, p1, Hp, p->grd(g[SHIVAE_P_PHIX2_W]), H, nullptr, nullptr, 0, EPI_ATOMIC, sc));
        SV_TRY(colsum1(R, H, Hp, d_phx, p->grd(g[SHIVAE_P_PHIX2_B]), sc));
        SV_TRY(gemm1(NN, R, H, H, d_phx, Hp, p->prm(g[SHIVAE_P_PHIX2_W]), H, d_p1, Hp, nullptr, p1, Hp, EPI_MASK, sc));
        SV_TRY(gemm1(TN, H, X, R, d_p1, Hp, x, X, p->grd(g[SHIVAE_P_PHIX0_W]), X, nullptr, nullptr, 0, EPI_ATOMIC, sc));
        SV_TRY(colsum1(R, H, Hp, d_p1, p->grd(g[SHIVAE_P_PHIX0_B]), sc));
        if (fan) {  // join: the step's stream waits for every side stream
            for (int i = 0; i < shivae_plan::NSIDE; ++i) {
                SV_CHECK_HIP(hipEventRecord(p->ev_join[i], p->side[i]));
                SV_CHECK_HIP(hipStreamWaitEvent(s, p->ev_join[i], 0));
            }
        }
    }

    return SHIVAE_OK;
}

static int step(shivae_plan* p, const float* x, const uint8_t* mask, const float* u, const float* eps, float beta,
                float temp, float inv_batch, int flags, float* losses, hipStream_t s) {
    const bool grad = !(flags & SHIVAE_STEP_FORWARD_ONLY);
    SV_TRY(phase_scan_fwd(p, x, u, eps, temp, s));
    SV_TRY(phase_prior_kl(p, beta, inv_batch, grad, s));
    SV_TRY(phase_decoder(p, x, mask, inv_batch, flags, s));
    if (grad) SV_TRY(phase_scan_bwd(p, x, u, eps, temp, s));
    // sum per-timestep terms over t (fixed order), then the batch means (base_missvae.py:211-230)
    SV_TRY(launch_loss_reduce(p->wsf("rowloss"), p->wsf("persum"), p->T, p->B, beta, inv_batch, losses, s));
    p->last_flags = flags;
    return SHIVAE_OK;
}


// ShiVAE.sample_from_prior (shivae.py:156-214): generation scan, then the decoder over all rows in chunks.
static int generate(shivae_plan* p, const float* u, const float* eps, float temp, hipStream_t s) {
    const int H = p->H, K = p->K, D = p->D, E = p->E, P = p->P;
    const int Hp = p->Hp, Kp = p->Kp, Ep = p->Ep, E4 = p->E4;
    const int R = (int)p->R;
    const Tensor* g = p->glob;
    const int NT = GEMM_NT;
    SV_TRY(launch_pack(p->wjobs_dev, p->n_wjobs, p->bjobs_dev, p->n_bjobs, p->wsf("wpack"), s));
    ScanArgs sa = scan_args(p, u, eps, temp);
    SV_TRY(launch_scan_gen(sa, s));
    float *emb = p->wsf("emb"), *embc = p->wsf("embc"), *a0 = p->wsf("a0"), *pbuf = p->wsf("pbuf");
    const int ldA0 = p->ldA0, ldA1 = p->ldA1, ldHH = p->ldHH;
    for (int r0 = 0; r0 < R; r0 += p->Ra) {  // (dense rows: no more than the activation buffers hold)
        const int rc = std::min(p->Ra, R - r0);
        SV_TRY(launch_compact_emb(emb + (size_t)r0 * Ep, Ep, Hp, Kp, H, K, embc, E4, rc, s));
        SV_TRY(gemm1(NT, rc, D * 2 * H, E, embc, E4, p->prm(g[SHIVAE_P_DEC0_W]), E, a0, ldA0, p->prm(g[SHIVAE_P_DEC0_B]),
                     nullptr, 0, EPI_BIAS | EPI_RELU, s));
        SV_TRY(gemmg(NT, rc, H, 2 * H, ldA0, 2 * H, ldA1, 0, EPI_BIAS | EPI_RELU, p->f_l1, s));
        SV_TRY(gemmg(NT, rc, H, H, ldA1, H, ldA1, 0, EPI_BIAS | EPI_RELU, p->f_l2, s));
        SV_TRY(gemmg(NT, rc, H, H, ldA1, H, ldHH, 0, EPI_BIAS | EPI_RELU, p->f_h0, s));
        for (auto& kv : p->f_out) {
            const int key = kv.first, nout = key < 0 ? -key : key, ldin = key < 0 ? ldA1 : ldHH;
            if (p->stream_heads && head_kernels_fit(H, nout, ldin, P)) {
                SV_TRY(heads(false, p->h_out[key], nout, H, ldin, P, rc, false, s));
                continue;
            }
            SV_TRY(gemmg(NT, rc, nout, H, ldin, H, P, 0, EPI_BIAS, kv.second, s));
        }
        SV_TRY(copy_async(p->wsf("pfull") + (size_t)r0 * P, pbuf, (size_t)rc * P, s));
    }
    p->last_flags = SHIVAE_STEP_FORWARD_ONLY | SHIVAE_STEP_KEEP_PARAMS;
    return SHIVAE_OK;
}

}  // namespace shivae

// =========================================================================== C ABI
extern "C" {

const char* shivae_last_error(void) { return g_err; }
const char* shivae_version(void) { return "shivae_hip gfx950 " __DATE__ " " __TIME__; }

int shivae_plan_create(const shivae_config* cfg, shivae_plan** out) {
    SV_REQUIRE(cfg && out, "plan_create: null argument");
    SV_REQUIRE(cfg->T > 0 && cfg->B > 0 && cfg->H > 0 && cfg->Z > 0 && cfg->K > 0 && cfg->n_attr > 0,
               "plan_create: T,B,H,Z,K,n_attr must be positive");
    SV_REQUIRE(cfg->attr_type && cfg->attr_nclass, "plan_create: attribute tables missing");
    SV_REQUIRE((int64_t)cfg->T * cfg->B < (1ll << 31) - 65536, "plan_create: T*B too large for one call; split the batch");
    shivae_plan* p = new shivae_plan();
    p->T = cfg->T; p->B = cfg->B; p->H = cfg->H; p->Z = cfg->Z; p->K = cfg->K; p->D = cfg->n_attr;
    p->learn_std = cfg->learn_std ? 1 : 0;
    p->R = (int64_t)p->T * p->B;
    int xoff = 0, poff = 0;
    p->nbc = 0;
    for (int d = 0; d < p->D; ++d) {
        AttrDesc a;
        a.type = cfg->attr_type[d];
        if (a.type < SHIVAE_REAL || a.type > SHIVAE_CAT) {
            set_error("plan_create: attribute %d has unknown type %d", d, a.type);
            delete p;
            return SHIVAE_ERR_INVALID;
        }
        a.dim = a.type == SHIVAE_CAT ? cfg->attr_nclass[d] : 1;
        if (a.type == SHIVAE_CAT && a.dim < 2) {
            set_error("plan_create: categorical attribute %d needs nclass >= 2", d);
            delete p;
            return SHIVAE_ERR_INVALID;
        }
        a.nout = a.type <= SHIVAE_POS ? (p->learn_std ? 2 : 1) : (a.type == SHIVAE_BIN ? 1 : a.dim - 1);
        a.xoff = xoff; a.poff = poff;
        xoff += a.dim; poff += a.nout;
        p->attrs.push_back(a);
        const bool bc = a.type >= SHIVAE_BIN;
        p->bc_index.push_back(bc ? p->nbc : -1);
        if (bc) { p->bc_attr.push_back(d); p->nbc++; }
    }
    p->X = xoff; p->P = poff; p->E = 2 * p->H + p->K;
    // likelihood kernel: largest group of attributes (multiple of 64) whose x + mask + parameter spans stay
    // within ~6 KB of LDS per wave (keeps 24-32 waves per CU resident)
    for (int ga = 64;; ga *= 2) {
        int xs = 0, ps = 0;
        for (int d0 = 0; d0 < p->D; d0 += ga) {
            const int d1 = std::min(d0 + ga, p->D);
            const int xe = d1 < p->D ? p->attrs[d1].xoff : p->X, pe = d1 < p->D ? p->attrs[d1].poff : p->P;
            xs = std::max(xs, xe - p->attrs[d0].xoff);
            ps = std::max(ps, pe - p->attrs[d0].poff);
        }
        if (ga > 64 && 5 * xs + 4 * ps > 6144) break;
        p->nll_group = ga; p->nll_xspan = xs; p->nll_pspan = ps;
        if (ga >= p->D) break;
    }
    p->Hp = round_up(p->H, 16); p->Zp = round_up(p->Z, 16); p->Kp = round_up(p->K, 16);
    p->Ep = 2 * p->Hp + p->Kp; p->E4 = round_up(p->E, 4); p->Z4 = round_up(p->Z, 4);
    if (p->Hp > 512) {
        set_error("plan_create: h_dim %d > 512 not supported by the recurrent kernel", p->H);
        delete p;
        return SHIVAE_ERR_INVALID;
    }
    p->ldA0 = p->D * 2 * p->H + 64; p->ldA1 = p->D * p->H + 64; p->ldHH = std::max(p->nbc, 1) * p->H + 64;
    build_layout(p);
    {
        std::vector<PackJob> wj, bj;
        build_pack_plan(p, wj, bj, false);
    }
    // decoder row chunk: bytes per row of chunk-local activations
    const double per_row = 4.0 * ((double)p->D * p->H * 8 + (double)std::max(p->nbc, 1) * p->H * 2 + p->P + 2.0 * p->D + 2.0 * p->E4 +
                                  2.0 * p->Hp + 2.0 * p->Zp);
    int rc = cfg->chunk_rows;
    if (rc <= 0) {
        // bytes of chunk-local activations (SHIVAE_CHUNK_BUDGET: tests of the host's chunk sizing on small models)
        const char* cb = getenv("SHIVAE_CHUNK_BUDGET");
        const double budget = cb && atof(cb) > 0 ? atof(cb) : 40e9;
        rc = (int)std::min<double>((double)p->R, std::max(256.0, budget / per_row));
        if (rc < p->R) rc = std::max(256, rc / 256 * 256);
    }
    p->Rc = (int)std::min<int64_t>(rc, p->R);
    p->Ra = p->Rc;
    {
        const char* f = getenv("SHIVAE_COMPACT");
        p->compact = f ? atoi(f) != 0 : 1;
        const char* hm = getenv("SHIVAE_HEADS");
        p->stream_heads = !(hm && strcmp(hm, "gemm") == 0);
        const char* dm = getenv("SHIVAE_DEMB");
        const char* lg = getenv("SHIVAE_L0D_GATHER");
        p->l0_gather = lg ? atoi(lg) != 0 : 1;
        p->demb_mode = !dm ? 2 : strcmp(dm, "dense") == 0 ? 0 : strcmp(dm, "compact") == 0 ? 1 : 2;
    }
    build_workspace(p);
    *out = p;
    return SHIVAE_OK;
}

static void release_bound(shivae_plan* p) {
    for (void* d : p->owned) (void)hipFree(d);
    p->owned.clear();
    if (p->cnt_host) {
        prof_forget_counts(p->cnt_host, p->cnt_host + (size_t)p->cnt_slots * p->D);
        (void)hipHostFree(p->cnt_host);
        p->cnt_host = nullptr;
    }
    if (p->cnt_guard) {
        (void)hipHostFree(p->cnt_guard);
        p->cnt_guard = nullptr;
    }
    p->cnt_slots = p->cnt_next = 0;
    for (int i = 0; i < shivae_plan::NSIDE; ++i) {
        if (p->side[i]) { (void)hipStreamDestroy(p->side[i]); p->side[i] = nullptr; }
        if (p->ev_join[i]) { (void)hipEventDestroy(p->ev_join[i]); p->ev_join[i] = nullptr; }
    }
    if (p->ev_fork) { (void)hipEventDestroy(p->ev_fork); p->ev_fork = nullptr; }
}

void shivae_plan_destroy(shivae_plan* plan) {
    if (!plan) return;
    release_bound(plan);
    delete plan;
}

int64_t shivae_plan_num_params(const shivae_plan* p) { return p ? p->n_params : 0; }
int shivae_plan_x_dim(const shivae_plan* p) { return p ? p->X : 0; }
int shivae_plan_p_dim(const shivae_plan* p) { return p ? p->P : 0; }
int shivae_plan_chunk_rows(const shivae_plan* p) { return p ? p->Rc : 0; }
float shivae_plan_observed_fraction(const shivae_plan* p) { return p ? p->obs_frac : -1.f; }
int shivae_plan_set_observed_fraction(shivae_plan* p, float f) {
    SV_REQUIRE(p, "set_observed_fraction: null plan");
    SV_REQUIRE(f <= 1.f, "set_observed_fraction: %g is not a fraction", (double)f);
    p->obs_frac = f < 0.f ? -1.f : f;
    return SHIVAE_OK;
}
int shivae_plan_activation_rows(const shivae_plan* p) { return p ? p->Ra : 0; }
long long shivae_plan_overflow_chunks(const shivae_plan* p) { return p ? p->n_overflow : 0; }
int shivae_plan_set_activation_rows(shivae_plan* p, int rows) {
    SV_REQUIRE(p, "set_activation_rows: null plan");
    SV_REQUIRE(!p->bound, "set_activation_rows: the plan is bound (the workspace layout is fixed by then); set it before shivae_plan_bind");
    p->Ra = rows <= 0 ? p->Rc : std::min(std::max(rows, 16), p->Rc);
    p->ws_off.clear();
    build_workspace(p);
    return SHIVAE_OK;
}
size_t shivae_plan_workspace_bytes(const shivae_plan* p) { return p ? p->ws_bytes : 0; }

int shivae_plan_param_info(const shivae_plan* p, int id, int attr, int64_t* offset, int* rows, int* cols) {
    SV_REQUIRE(p && id >= 0 && id < SHIVAE_P_COUNT, "param_info: bad tensor id %d", id);
    Tensor t;
    if (id < SHIVAE_P_DEC0_W) t = p->glob[id];
    else {
        SV_REQUIRE(attr >= 0 && attr < p->D, "param_info: bad attribute %d", attr);
        t = p->per_attr[id][attr];
    }
    SV_REQUIRE(t.off >= 0, "param_info: tensor %d does not exist for attribute %d", id, attr);
    if (offset) *offset = t.off;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return SHIVAE_OK;
}

int shivae_plan_bind(shivae_plan* p, void* ws, size_t bytes, float* params, float* grads) {
    SV_REQUIRE(p && ws && params && grads, "plan_bind: null argument");
    SV_REQUIRE(bytes >= p->ws_bytes, "plan_bind: workspace has %zu bytes, %zu needed", bytes, p->ws_bytes);
    SV_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "plan_bind: workspace must be 256-B aligned");
    SV_REQUIRE(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads)) & 15) == 0,
               "plan_bind: parameter/gradient buffers must be 16-B aligned");
    release_bound(p);
    p->obs_frac = -1.f;  // (measured again on the first training chunk of the new binding, unless the caller sets it)
    p->f_out.clear(); p->b_out_w.clear(); p->b_out_d.clear(); p->c_out.clear(); p->h_out.clear();
    p->ws = static_cast<char*>(ws);
    p->params = params;
    p->grads = grads;
    // Padding columns and the t=0 h_prev rows rely on the workspace starting at zero.  Cleared in pieces of 1 GiB,
    // asynchronously, with one wait at the end of bind (170 GB at cfg3 as ONE fill is a dispatch of its own kind; it
    // was a suspect for the round-1 crash under `rocprofv3 --pmc` and turned out not to be the cause -- DESIGN.md s.4).
    for (size_t off = 0; off < p->ws_bytes; off += (size_t)1 << 30)
        SV_CHECK_HIP(hipMemsetAsync(p->ws + off, 0, std::min<size_t>((size_t)1 << 30, p->ws_bytes - off), nullptr));
    {
        const int64_t nchunks = (p->R + p->Rc - 1) / p->Rc;
        p->cnt_slots = (int)std::max<int64_t>(64, std::min<int64_t>(4 * nchunks, 1 << 16));
        SV_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->cnt_host), (size_t)p->cnt_slots * p->D * sizeof(int),
                                   hipHostMallocDefault));
        p->cnt_next = 0;
        SV_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->cnt_guard), (size_t)p->D * sizeof(int), hipHostMallocDefault));
    }
    if (p->R <= fanout_max_rows()) {  // (only steps small enough to use them get side streams)
        SV_CHECK_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < shivae_plan::NSIDE; ++i) {
            SV_CHECK_HIP(hipStreamCreateWithFlags(&p->side[i], hipStreamNonBlocking));
            SV_CHECK_HIP(hipEventCreateWithFlags(&p->ev_join[i], hipEventDisableTiming));
        }
    }
    SV_TRY(init_sqnorm_noalloc());  // (per-device set-up happens here, outside any stream capture)
    SV_TRY(upload(p, p->attrs, &p->attrs_dev));
    std::vector<PackJob> wj, bj;
    build_pack_plan(p, wj, bj, true);
    p->n_wjobs = (int)wj.size(); p->n_bjobs = (int)bj.size();
    SV_TRY(upload(p, wj, &p->wjobs_dev));
    SV_TRY(upload(p, bj, &p->bjobs_dev));
    SV_TRY(build_tables(p));
    SV_CHECK_HIP(hipDeviceSynchronize());
    p->bound = true;
    return SHIVAE_OK;
}

int shivae_elbo_step(shivae_plan* p, const float* x, const uint8_t* mask, const float* u, const float* eps, float beta,
                     float temp, float inv_batch, int flags, float* losses, shivae_stream stream) {
    SV_REQUIRE(p && x && mask && losses, "elbo_step: null argument");
    SV_REQUIRE((u != nullptr) == (eps != nullptr), "elbo_step: u and eps are given together, or both null (in-kernel noise)");
    if (!p->bound) {
        set_error("elbo_step: plan has no workspace/parameters bound (call shivae_plan_bind)");
        return SHIVAE_ERR_UNBOUND;
    }
    SV_REQUIRE(temp > 0.f, "elbo_step: temperature must be > 0");
    return step(p, x, mask, u, eps, beta, temp, inv_batch, flags, losses, static_cast<hipStream_t>(stream));
}

#define SV_BOUND(name)                                                                                   \
    if (!p->bound) {                                                                                     \
        set_error(name ": plan has no workspace/parameters bound (call shivae_plan_bind)");              \
        return SHIVAE_ERR_UNBOUND;                                                                       \
    }

int shivae_scan_fwd(shivae_plan* p, const float* x, const float* u, const float* eps, float temp, shivae_stream stream) {
    SV_REQUIRE(p && x, "scan_fwd: null argument");
    SV_REQUIRE((u != nullptr) == (eps != nullptr), "scan_fwd: u and eps are given together, or both null (in-kernel noise)");
    SV_BOUND("scan_fwd");
    SV_REQUIRE(temp > 0.f, "scan_fwd: temperature must be > 0");
    p->last_flags = SHIVAE_STEP_FORWARD_ONLY;
    return phase_scan_fwd(p, x, u, eps, temp, static_cast<hipStream_t>(stream));
}

int shivae_prior_kl(shivae_plan* p, float beta, float inv_batch, int flags, shivae_stream stream) {
    SV_REQUIRE(p, "prior_kl: null argument");
    SV_BOUND("prior_kl");
    return phase_prior_kl(p, beta, inv_batch, !(flags & SHIVAE_STEP_FORWARD_ONLY), static_cast<hipStream_t>(stream));
}

int shivae_decoder_nll(shivae_plan* p, const float* x, const uint8_t* mask, float inv_batch, int flags,
                       shivae_stream stream) {
    SV_REQUIRE(p && x && mask, "decoder_nll: null argument");
    SV_BOUND("decoder_nll");
    p->last_flags = flags;
    return phase_decoder(p, x, mask, inv_batch, flags, static_cast<hipStream_t>(stream));
}

int shivae_scan_bwd(shivae_plan* p, const float* x, const float* u, const float* eps, float temp, const float* d_emb,
                    const float* dmu, const float* dstd, const float* dlogits, shivae_stream stream) {
    SV_REQUIRE(p && x, "scan_bwd: null argument");
    SV_REQUIRE((u != nullptr) == (eps != nullptr), "scan_bwd: u and eps are given together, or both null (in-kernel noise)");
    SV_BOUND("scan_bwd");
    SV_REQUIRE(temp > 0.f, "scan_bwd: temperature must be > 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = p->H, Z = p->Z, K = p->K, Hp = p->Hp, Zp = p->Zp, Kp = p->Kp, Ep = p->Ep, E = p->E;
    const int64_t R = p->R;
    // caller-supplied gradients replace the records the other phases accumulate (padding columns stay zero)
    if (d_emb) {
        float* de = p->wsf("d_emb");
        SV_TRY(zero_async(de, (size_t)R * Ep, s));
        SV_TRY(launch_copy2d(d_emb, E, 0, H, de, Ep, 0, R, s));
        SV_TRY(launch_copy2d(d_emb, E, H, K, de, Ep, Hp, R, s));
        SV_TRY(launch_copy2d(d_emb, E, H + K, H, de, Ep, Hp + Kp, R, s));
    }
    if (dmu) SV_TRY(launch_copy2d(dmu, Z, 0, Z, p->wsf("dmu_ext"), Zp, 0, R, s));
    if (dstd) SV_TRY(launch_copy2d(dstd, Z, 0, Z, p->wsf("dsd_ext"), Zp, 0, R, s));
    if (dlogits) SV_TRY(launch_copy2d(dlogits, K, 0, K, p->wsf("dlg_kl"), Kp, 0, R, s));
    return phase_scan_bwd(p, x, u, eps, temp, s);
}

int shivae_loss_reduce(shivae_plan* p, float beta, float inv_batch, float* losses, shivae_stream stream) {
    SV_REQUIRE(p && losses, "loss_reduce: null argument");
    SV_BOUND("loss_reduce");
    return launch_loss_reduce(p->wsf("rowloss"), p->wsf("persum"), p->T, p->B, beta, inv_batch, losses,
                              static_cast<hipStream_t>(stream));
}

int shivae_sample_prior(shivae_plan* p, const float* u, const float* eps, float temp, shivae_stream stream) {
    SV_REQUIRE(p, "sample_prior: null argument");
    SV_REQUIRE((u != nullptr) == (eps != nullptr), "sample_prior: u and eps are given together, or both null (in-kernel noise)");
    if (!p->bound) {
        set_error("sample_prior: plan has no workspace/parameters bound (call shivae_plan_bind)");
        return SHIVAE_ERR_UNBOUND;
    }
    SV_REQUIRE(temp > 0.f, "sample_prior: temperature must be > 0");
    return generate(p, u, eps, temp, static_cast<hipStream_t>(stream));
}

int shivae_plan_set_noise_seed(shivae_plan* p, uint64_t seed) {
    SV_REQUIRE(p, "plan_set_noise_seed: null plan");
    p->noise_seed = seed;
    return SHIVAE_OK;
}

int shivae_noise_fill(uint64_t seed, int64_t rows, int K, int Z, float* u, float* eps, shivae_stream stream) {
    SV_REQUIRE(rows >= 0 && K >= 0 && Z >= 0, "noise_fill: negative extent");
    return launch_noise_fill(seed, rows, K, Z, u, eps, static_cast<hipStream_t>(stream));
}

int shivae_output_width(const shivae_plan* p, int which) {
    if (!p) return 0;
    switch (which) {
        case SHIVAE_O_Z: case SHIVAE_O_Z_MEAN: case SHIVAE_O_Z_STD: case SHIVAE_O_ZP_MEAN: case SHIVAE_O_ZP_STD: return p->Z;
        case SHIVAE_O_S: case SHIVAE_O_S_PROBS: return p->K;
        case SHIVAE_O_DEC_PARAMS: return p->P;
        case SHIVAE_O_ROWLOSS: return 6;
        case SHIVAE_O_H_PREV: case SHIVAE_O_PHI_Z: return p->H;
        case SHIVAE_O_A_S: case SHIVAE_O_E1: case SHIVAE_O_E2: case SHIVAE_O_C_NEW: return p->H;
        case SHIVAE_O_D_PHIZ_PRE: case SHIVAE_O_D_E2_PRE: case SHIVAE_O_D_E1_PRE: case SHIVAE_O_D_AS_PRE: return p->H;
        case SHIVAE_O_LOGITS: case SHIVAE_O_D_LOGITS_PRE: return p->K;
        case SHIVAE_O_D_MU: case SHIVAE_O_D_STD_PRE: return p->Z;
        case SHIVAE_O_GATES: case SHIVAE_O_D_GATES: return 4 * p->H;
    }
    return 0;
}

int shivae_get_output(shivae_plan* p, int which, float* dst, shivae_stream stream) {
    SV_REQUIRE(p && dst, "get_output: null argument");
    if (!p->bound) { set_error("get_output: plan not bound"); return SHIVAE_ERR_UNBOUND; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t R = p->R;
    switch (which) {
        case SHIVAE_O_Z: return launch_unpad_copy(p->wsf("z"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_Z_MEAN: return launch_unpad_copy(p->wsf("mu"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_Z_STD: return launch_unpad_copy(p->wsf("sd"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_ZP_MEAN: return launch_unpad_copy(p->wsf("mup"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_ZP_STD: return launch_unpad_copy(p->wsf("sdp"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_S: return launch_unpad_copy(p->wsf("emb"), p->Ep, p->Hp, p->K, dst, R, s);
        case SHIVAE_O_S_PROBS: return launch_unpad_copy(p->wsf("probs"), p->Kp, 0, p->K, dst, R, s);
        case SHIVAE_O_PHI_Z: return launch_unpad_copy(p->wsf("emb"), p->Ep, 0, p->H, dst, R, s);
        case SHIVAE_O_H_PREV: return launch_unpad_copy(p->wsf("emb"), p->Ep, p->Hp + p->Kp, p->H, dst, R, s);
        case SHIVAE_O_ROWLOSS: return launch_unpad_copy(p->wsf("rowloss"), 6, 0, 6, dst, R, s);
        case SHIVAE_O_DEC_PARAMS:
            SV_REQUIRE(p->last_flags & SHIVAE_STEP_KEEP_PARAMS, "get_output: last step did not keep decoder parameters");
            return launch_unpad_copy(p->wsf("pfull"), p->P, 0, p->P, dst, R, s);
        // records of the recurrent scans (kernel-level parity tests)
        case SHIVAE_O_A_S: return launch_unpad_copy(p->wsf("as"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_E1: return launch_unpad_copy(p->wsf("e1"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_E2: return launch_unpad_copy(p->wsf("e2"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_C_NEW: return launch_unpad_copy(p->wsf("cnew"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_LOGITS: return launch_unpad_copy(p->wsf("logits"), p->Kp, 0, p->K, dst, R, s);
        case SHIVAE_O_D_PHIZ_PRE: return launch_unpad_copy(p->wsf("d_phiz_pre"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_D_E2_PRE: return launch_unpad_copy(p->wsf("d_e2_pre"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_D_E1_PRE: return launch_unpad_copy(p->wsf("d_e1_pre"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_D_AS_PRE: return launch_unpad_copy(p->wsf("d_as_pre"), p->Hp, 0, p->H, dst, R, s);
        case SHIVAE_O_D_LOGITS_PRE: return launch_unpad_copy(p->wsf("d_lg_pre"), p->Kp, 0, p->K, dst, R, s);
        case SHIVAE_O_D_MU: return launch_unpad_copy(p->wsf("d_mu"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_D_STD_PRE: return launch_unpad_copy(p->wsf("d_sd_pre"), p->Zp, 0, p->Z, dst, R, s);
        case SHIVAE_O_GATES: case SHIVAE_O_D_GATES: {  // four gate blocks i,f,g,o, each padded H -> Hp in the records
            const float* src = p->wsf(which == SHIVAE_O_GATES ? "gates" : "d_gates");
            for (int gt = 0; gt < 4; ++gt)
                SV_TRY(launch_copy2d(src, 4 * p->Hp, gt * p->Hp, p->H, dst, 4 * p->H, gt * p->H, R, s));
            return SHIVAE_OK;
        }
    }
    set_error("get_output: unknown output %d", which);
    return SHIVAE_ERR_INVALID;
}

int shivae_get_persum(shivae_plan* p, float* dst, shivae_stream stream) {
    SV_REQUIRE(p && dst, "get_persum: null argument");
    if (!p->bound) { set_error("get_persum: plan not bound"); return SHIVAE_ERR_UNBOUND; }
    return copy_async(dst, p->wsf("persum"), (size_t)p->B * 6, static_cast<hipStream_t>(stream));
}

int shivae_heter_nll(shivae_plan* p, const float* x, const uint8_t* mask, const float* params, int ldp, float* dparams,
                     int rows, float gs, float* rowloss, shivae_stream stream) {
    SV_REQUIRE(p && x && mask && params && rowloss, "heter_nll: null argument");
    SV_REQUIRE(ldp >= p->P && rows >= 0, "heter_nll: ldp %d < P %d", ldp, p->P);
    if (!p->attrs_dev) {
        SV_TRY(upload(p, p->attrs, &p->attrs_dev));
    }
    return launch_heter_nll(p->attrs_dev, p->D, p->learn_std, x, p->X, mask, p->X, params, ldp, dparams, ldp, rows, gs,
                            rowloss, p->X, p->P, p->nll_group, p->nll_xspan, p->nll_pspan, static_cast<hipStream_t>(stream));
}

int shivae_mc_impute(shivae_plan* p, const float* params, int ldp, int rows, int n_samples, uint64_t seed, float* xhat,
                     shivae_stream stream) {
    SV_REQUIRE(p && params && xhat, "mc_impute: null argument");
    SV_REQUIRE(ldp >= p->P && rows >= 0 && n_samples > 0, "mc_impute: bad sizes");
    if (!p->attrs_dev) {
        SV_TRY(upload(p, p->attrs, &p->attrs_dev));
    }
    return launch_mc_impute(p->attrs_dev, p->D, p->learn_std, params, ldp, rows, n_samples, seed, xhat, p->X,
                            static_cast<hipStream_t>(stream));
}

int shivae_kl_terms(const float* mu_q, const float* std_q, const float* mu_p, const float* std_p, int ldz,
                    const float* s_probs, int ldk, int rows, int Z, int K, float kl_scale, float* dmu_q, float* dstd_q,
                    float* dmu_p, float* dstd_p_pre, float* dlogits, float* rowloss, shivae_stream stream) {
    SV_REQUIRE(mu_q && std_q && mu_p && std_p && s_probs && rowloss, "kl_terms: null argument");
    SV_REQUIRE(Z > 0 && K > 0 && ldz >= Z && ldk >= K && rows >= 0, "kl_terms: bad sizes");
    const bool any = dmu_q || dstd_q || dmu_p || dstd_p_pre || dlogits;
    SV_REQUIRE(!any || (dmu_q && dstd_q && dmu_p && dstd_p_pre && dlogits), "kl_terms: pass all gradient outputs or none");
    return launch_kl(mu_q, std_q, mu_p, std_p, ldz, s_probs, ldk, dmu_q, dstd_q, dmu_p, dstd_p_pre, dlogits, rows, Z, K,
                     kl_scale, rowloss, static_cast<hipStream_t>(stream));
}

int shivae_reparam_fwd(const float* logits, const float* u, int ldk, int K, float temp, float* s, float* s_probs,
                       const float* mean, const float* sd, const float* eps, int ldz, int Z, float* z, int rows,
                       shivae_stream stream) {
    SV_REQUIRE(logits && u && s && s_probs && mean && sd && eps && z, "reparam_fwd: null argument");
    SV_REQUIRE(temp > 0.f && K > 0 && Z > 0 && ldk >= K && ldz >= Z, "reparam_fwd: bad sizes");
    return launch_reparam_fwd(logits, u, ldk, K, temp, s, s_probs, mean, sd, eps, ldz, Z, z, rows,
                              static_cast<hipStream_t>(stream));
}

int shivae_reparam_bwd(const float* s, const float* ds, int ldk, int K, float temp, float* dlogits, const float* eps,
                       const float* dz, int ldz, int Z, float* dmean, float* dstd, int rows, shivae_stream stream) {
    SV_REQUIRE(s && ds && dlogits && eps && dz && dmean && dstd, "reparam_bwd: null argument");
    SV_REQUIRE(temp > 0.f && K > 0 && Z > 0 && ldk >= K && ldz >= Z, "reparam_bwd: bad sizes");
    return launch_reparam_bwd(s, ds, ldk, K, temp, dlogits, eps, dz, ldz, Z, dmean, dstd, rows,
                              static_cast<hipStream_t>(stream));
}

int shivae_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* aux, int ldaux, int epi, int splitk, shivae_stream stream) {
    SV_REQUIRE(A && B && C, "gemm: null operand");
    SV_REQUIRE(layout >= 0 && layout <= 2, "gemm: bad layout %d", layout);
    SV_REQUIRE(!(epi & EPI_BIAS) || bias, "gemm: bias epilogue without bias");
    SV_REQUIRE(!(epi & EPI_MASK) || aux, "gemm: mask epilogue without aux");
    GemmDesc d{layout, M, N, K, lda, ldb, ldc, ldaux, epi, splitk < 1 ? 1 : splitk, 0, 0};
    GemmGroup g{A, B, C, bias, aux};
    return launch_gemm(d, nullptr, 1, g, static_cast<hipStream_t>(stream));
}

size_t shivae_presplit_bytes(int N, int K) { return (N > 0 && K > 0) ? presplit_bytes(N, K) : 0; }

int shivae_presplit_weights(const float* W, int N, int K, int ldw, void* image, shivae_stream stream) {
    SV_REQUIRE(W && image && N > 0 && K > 0 && ldw >= K, "presplit_weights: bad argument");
    return launch_presplit(W, N, K, ldw, 1, 0, image, static_cast<hipStream_t>(stream));
}

int shivae_gemm_nt_presplit(int M, int N, int K, const float* A, int lda, const void* image, float* C, int ldc,
                            const float* bias, const float* aux, int ldaux, int epi, int splitk, shivae_stream stream) {
    SV_REQUIRE(A && image && C, "gemm_nt_presplit: null operand");
    SV_REQUIRE(!(epi & EPI_BIAS) || bias, "gemm_nt_presplit: bias epilogue without bias");
    SV_REQUIRE(!(epi & EPI_MASK) || aux, "gemm_nt_presplit: mask epilogue without aux");
    GemmDesc d{GEMM_NT, M, N, K, lda, K, ldc, ldaux, epi, splitk < 1 ? 1 : splitk, 0, 0, 0, N};
    SV_REQUIRE(split_nt_fits(d), "gemm_nt_presplit: needs N %% 256 == 0, K %% 4 == 0, K >= 32, M > 64, lda / ldc / ldaux %% 4 == 0 "
               "(M=%d N=%d K=%d)", M, N, K);
    GemmGroup g{A, static_cast<const float*>(image), C, bias, aux};
    return launch_gemm_split_nt(d, nullptr, 1, g, static_cast<hipStream_t>(stream));
}

int shivae_profile_active(void) { return g_prof_on ? 1 : 0; }

int shivae_profile_start(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = true;
    return SHIVAE_OK;
}

int shivae_profile_stop(double* out, int max_entries) {
    g_prof_on = false;
    SV_CHECK_HIP(hipDeviceSynchronize());
    double calls[PROF_COUNT] = {0}, ms[PROF_COUNT] = {0}, work[PROF_COUNT] = {0};
    for (auto& r : g_prof) {
        if (!r.rag_attr || !r.rag_cnt || r.rag_attr->empty() || r.rag_ext <= 0) continue;
        double live = 0.0;
        for (int a : *r.rag_attr) live += r.rag_cnt[a];
        r.work *= live / ((double)r.rag_attr->size() * r.rag_ext);
    }
    for (auto& r : g_prof) {
        float t = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.id >= 0 && r.id < PROF_COUNT) {
            calls[r.id] += 1; ms[r.id] += t; work[r.id] += r.work;
        }
    }
    if (getenv("SHIVAE_PROF_DETAIL")) {  // per-shape table on stderr
        std::map<std::vector<int>, std::pair<double, double>> tab;
        std::map<std::vector<int>, int> cnt;
        for (auto& r : g_prof) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
            std::vector<int> key{r.id, r.M, r.N, r.K, r.G};
            tab[key].first += t; tab[key].second += r.work; cnt[key]++;
        }
        for (auto& kv : tab)
            fprintf(stderr, "prof id=%2d M=%7d N=%7d K=%7d G=%4d launches=%4d ms=%9.3f rate=%8.2f T/s\n", kv.first[0],
                    kv.first[1], kv.first[2], kv.first[3], kv.first[4], cnt[kv.first], kv.second.first,
                    kv.second.second / (kv.second.first * 1e-3) / 1e12);
    }
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    int n = 0;
    for (int i = 0; i < PROF_COUNT && n < max_entries; ++i)
        if (calls[i] > 0) { out[4 * n] = i; out[4 * n + 1] = calls[i]; out[4 * n + 2] = ms[i]; out[4 * n + 3] = work[i]; ++n; }
    return n;
}

/* diagnostics (tools/gemm_stamps.py): 4 s_memtime stamps per workgroup of the following GEMM launches, or NULL */
int shivae_debug_gemm_stamps(unsigned long long* buf_dev) {
    shivae::g_gemm_stamps = buf_dev;
    return SHIVAE_OK;
}

int shivae_set_gemm_split(int on) {
    shivae::g_gemm_split = on ? 1 : 0;
    return SHIVAE_OK;
}

int shivae_zero(float* buf, int64_t n, shivae_stream stream) {
    SV_REQUIRE(buf && n >= 0, "zero: bad argument");
    SV_CHECK_HIP(hipMemsetAsync(buf, 0, (size_t)n * sizeof(float), static_cast<hipStream_t>(stream)));
    return SHIVAE_OK;
}

int shivae_sqnorm(const float* buf, int64_t n, float* scratch, float* out, shivae_stream stream) {
    SV_REQUIRE(buf && scratch && out && n >= 0, "sqnorm: bad argument");
    return launch_sqnorm(buf, n, scratch, out, static_cast<hipStream_t>(stream));
}

int shivae_init(void) { return init_sqnorm_noalloc(); }

int shivae_sqnorm_noalloc(const float* buf, int64_t n, float* out, shivae_stream stream) {
    SV_REQUIRE(buf && out && n >= 0, "sqnorm_noalloc: bad argument");
    SV_REQUIRE(n <= SHIVAE_SQNORM_NOALLOC_MAX, "sqnorm_noalloc: %lld elements exceed the supported maximum of %lld",
               (long long)n, (long long)SHIVAE_SQNORM_NOALLOC_MAX);
    return launch_sqnorm_noalloc(buf, n, out, static_cast<hipStream_t>(stream));
}

// ---- gradient all-reduce over an RCCL communicator.  librccl is not a link-time dependency: it is
// opened on first use, so that single-GPU users never load it and a process that already holds an RCCL (torch's)
// shares that copy.
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static int rccl_api(const RcclApi** out) {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.handle, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.handle, "ncclGroupEnd"));
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        }
    });
    if (!api.handle || !api.AllReduce || !api.GroupStart || !api.GroupEnd || !api.GetUniqueId || !api.CommInitRank ||
        !api.CommDestroy || !api.GetErrorString) {
        set_error("RCCL not available: %s", api.handle ? "librccl lacks a required symbol" : dlerror());
        return SHIVAE_ERR_HIP;
    }
    *out = &api;
    return SHIVAE_OK;
}
#define SV_CHECK_RCCL(api, expr)                                                                  \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess) {                                                                 \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(r__));   \
            return SHIVAE_ERR_HIP;                                                                \
        }                                                                                         \
    } while (0)

int shivae_comm_unique_id(void* id_out) {
    SV_REQUIRE(id_out, "comm_unique_id: null argument");
    const RcclApi* api;
    SV_TRY(rccl_api(&api));
    ncclUniqueId id;
    SV_CHECK_RCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return SHIVAE_OK;
}

int shivae_comm_init(void** comm_out, int world, int rank, const void* id_in) {
    SV_REQUIRE(comm_out && id_in && world >= 1 && rank >= 0 && rank < world, "comm_init: bad argument");
    const RcclApi* api;
    SV_TRY(rccl_api(&api));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclComm_t c = nullptr;
    SV_CHECK_RCCL(api, api->CommInitRank(&c, world, id, rank));
    *comm_out = c;
    return SHIVAE_OK;
}

int shivae_comm_destroy(void* comm) {
    if (!comm) return SHIVAE_OK;
    const RcclApi* api;
    SV_TRY(rccl_api(&api));
    SV_CHECK_RCCL(api, api->CommDestroy(static_cast<ncclComm_t>(comm)));
    return SHIVAE_OK;
}

int shivae_allreduce_grads(float* grads, int64_t n, void* comm, int64_t bucket_floats, shivae_stream stream) {
    SV_REQUIRE(grads && comm && n >= 0, "allreduce_grads: bad argument");
    const RcclApi* api;
    SV_TRY(rccl_api(&api));
    const int64_t b = bucket_floats > 0 ? bucket_floats : n;
    // one group of in-place sums, bucket by bucket: ring collectives over xGMI are bound per link, and buckets of
    // ~100 MB keep every link streaming while staying small enough to pipeline
    SV_CHECK_RCCL(api, api->GroupStart());
    for (int64_t a = 0; a < n; a += b) {
        const ncclResult_t r = api->AllReduce(grads + a, grads + a, (size_t)std::min<int64_t>(b, n - a), ncclFloat, ncclSum,
                                              static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
        if (r != ncclSuccess) {
            // close the group before reporting: an open group would swallow every later collective of this thread
            // (torch's included)
            (void)api->GroupEnd();
            set_error("allreduce_grads: ncclAllReduce of bucket at %lld -> %s", (long long)a, api->GetErrorString(r));
            return SHIVAE_ERR_HIP;
        }
    }
    SV_CHECK_RCCL(api, api->GroupEnd());
    return SHIVAE_OK;
}

int shivae_clip_adam(float* params, const float* grads, float* m, float* v, int64_t n, const float* sqnorm, float clip,
                     float lr, float b1, float b2, float eps, int step, shivae_stream stream) {
    SV_REQUIRE(params && grads && m && v && sqnorm, "clip_adam: null argument");
    return launch_clip_adam(params, grads, m, v, n, sqnorm, clip, lr, b1, b2, eps, step, static_cast<hipStream_t>(stream));
}

}  // extern "C"
