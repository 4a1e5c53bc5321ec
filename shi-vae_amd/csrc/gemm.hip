// Grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, fmaf chain).
//
// This is the workhorse of the Shi-VAE step: every Linear of the reference that is NOT on the
// serial recurrence (encoders.py:47-51 phi_x; shivae.py:56-61 prior; decoders.py:60-102 the
// per-attribute decoder trunks and heads = 99 % of the FLOPs) and all of their backward passes
// (input gradients and weight gradients) are batched over the T*B rows and run through here.
//
// Tile: (WM*TM*32) x (WN*TN*32) x 32 per workgroup of WM*WN waves (64 lanes each); each wave owns
// TM x TN accumulators of 32x32.  Operands are staged global -> registers -> LDS with the next
// K-tile's global loads in flight during the MFMAs of the current one; the 128x128 TN / NN tile stages
// by LDS-DMA instead (global_load_lds_dwordx4, no register round trip).  The K loops are straight-line:
// the last K-tile is peeled rather than guarded inside the loop (a branch per K-tile cost 5 %).
//
// Ragged groups (the mask-aware decoder, GemmDesc::ragged): a group's live extent can be a device-side
// count and its rows can go through index lists -- see GemmGroup in common.h and the kernel head below.
//
// LDS images and bank behaviour (MI355X_MICROARCH.md, LDS):
//   K-contiguous operand  : [rows][32+4] floats.  A lane reads the 4 consecutive k it needs for
//                           four MFMAs with one ds_read_b128; row stride 36 dwords puts the 16
//                           lanes of a b128 lane group on 16 distinct 4-bank slots (conflict-free).
//   row-contiguous operand: [32][rows] floats.  Lanes 0..31 read consecutive floats (ds_read_b32,
//                           conflict-free).
// MFMA operand map (cdna_hip_programming.md s.3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; with per-lane float4 v = X[.][8s+4h .. 8s+4h+3] (h = l>>5), MFMA number c uses
// v[c] on both sides, i.e. the k pair {8s+c, 8s+4+c}.  D: col = l&31, row = (r&3)+8*(r>>2)+4*h.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace shivae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// Operand pointers come out of the group table, so the compiler only knows them as generic pointers and would
// emit flat_load: flat operations count on BOTH vmcnt and lgkmcnt, and the first `s_waitcnt lgkmcnt(0)` for an
// LDS fragment read then also waits for the global loads of the next K-tile issued just before it -- the
// prefetch stops overlapping the MFMAs.  Loads through these global-address-space views are global_load.
typedef const __attribute__((address_space(1))) float* gbl_f32_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gbl_f32x4_ptr;
typedef const __attribute__((address_space(1))) int* gbl_i32_ptr;
// (the same for the epilogue: a flat_store also counts on lgkmcnt, so the LDS read of the next piece of the
// staged tile would wait for the previous piece's stores to complete)
__device__ __forceinline__ float4 gload4(const float* p) {
    const f32x4 t = *(gbl_f32x4_ptr)p;
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float gload(const float* p) { return *(gbl_f32_ptr)p; }
__device__ __forceinline__ void gstore4(float* p, float a, float b, float c, float d) {
    f32x4 t;
    t.x = a; t.y = b; t.z = c; t.w = d;
    *(__attribute__((address_space(1))) f32x4*)p = t;
}
__device__ __forceinline__ void gstore(float* p, float v) { *(__attribute__((address_space(1))) float*)p = v; }
__device__ __forceinline__ void gatomic_add(float* p, float v) {
    __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fp32 products on the bf16 matrix cores ("split" mode, SHIVAE_GEMM_SPLIT / shivae_set_gemm_split) ----------
// An fp32 value is the exact sum of three bf16 values: hi = bf16_rn(x), mid = bf16_rn(x - hi), lo = x - hi - mid
// (every subtraction exact, lo fits 8 bits).  a*b is then accumulated in fp32 as the six bf16 products whose magnitude
// is >= 2^-16 of the leading one -- hi*hi into the main accumulator, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi into a second
// one that is added at the end; the dropped ones (mid*lo, lo*mid, lo*lo) are <= 2^-26 |a b|.  v_mfma_f32_32x32x16_bf16
// runs at 16x the rate of the fp32-input MFMA, so six of them per product still give 2.67x its throughput.
// LDS images of a split tile (bf16, built when the fp32 registers are parked in LDS):
//   K-contiguous operand  : [rows][104] halves = hi[32] | mid[32] | lo[32] | pad[8]; a lane's fragment of one plane
//                           (8 consecutive k) is one ds_read_b128; 208-B rows are conflict-free for them
//   row-contiguous operand: [plane][8 k-quads][row & 3][row >> 2 (+8 pad)][2 dwords]: the four k of a k-quad of one
//                           row are 8 contiguous bytes, so a fragment (8 k) is 2 ds_read_b64 (256 B/clk; the
//                           first form, [k-pair][rows] dwords and 4 ds_read_b32 at 128 B/clk, made the TN / NN
//                           kernels LDS-bound).  A thread holds 4 consecutive k of 4 consecutive rows and stores
//                           one 8-B piece per row: lanes = consecutive row quads = contiguous LDS.  The padded
//                           stride of the (row & 3) blocks puts the four blocks a read touches on disjoint banks.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SPLIT_RS = 104;  // halves per row of the K-contiguous split image
__host__ __device__ constexpr int split_qs(int rows) { return rows / 4 + 8; }        // row-quad stride of a (row & 3) block
__host__ __device__ constexpr int split_rc_plane(int rows) { return 64 * split_qs(rows); }  // dwords per plane

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {  // {bf16 trunc(a), bf16 trunc(b)} as one dword
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302);
}
__device__ __forceinline__ float bf16_rn(float x) {  // x rounded to the nearest bf16 (ties to even), as a float
    return (float)(__bf16)x;
}
__device__ __forceinline__ void split3(float x, float& hi, float& mid, float& lo) {
    hi = bf16_rn(x);
    const float r = x - hi;   // exact: at most 16 significant bits
    mid = bf16_rn(r);
    lo = r - mid;             // exact: at most 8 significant bits, so its upper half is all of it
}

template <bool KC, int ROWS, int NT, bool VEC, bool SPLIT = false>
struct TileLoader {
    // KC : global X[row*ld + k], tile rows = ROWS along the M/N extent, 32 along K
    // !KC: global X[k*ld + row]
    // Loads are UNCONDITIONAL (out-of-range lanes read X[0] and are zeroed by a select): a branch
    // around a load makes hipcc wait vmcnt(0) per element and serialises the whole tile fetch.
    //
    // Row lists (ragged launches): the operand's rows -- tile rows for KC, k-rows for !KC -- may come through an
    // index list.  src[i] is the source row of this thread's piece i: fixed over the K loop for KC; for !KC it is
    // the k-row of the NEXT load() and is refreshed one K-tile ahead, so the index fetch never sits between a
    // load() and its data.
    static constexpr int NV = ROWS * 8 / NT;  // float4 per thread
    static constexpr int TPR = ROWS / 4;      // !KC: threads per k-row
    float4 v[NV > 0 ? NV : 1];
    int src[NV > 0 ? NV : 1];

    // piece i of thread tid -> (tile row, tile k).  Split mode gives a thread of a row-contiguous operand NV
    // CONSECUTIVE k of the same four rows, so that it can pack (k, k+1) bf16 pairs.
    __device__ __forceinline__ static int prow(int i, int tid) {
        return KC ? (tid + i * NT) >> 3 : ((SPLIT ? tid : tid + i * NT) % TPR) * 4;
    }
    __device__ __forceinline__ static int pk(int i, int tid) {
        return KC ? ((tid + i * NT) & 7) * 4 : (SPLIT ? NV * (tid / TPR) + i : (tid + i * NT) / TPR);
    }

    // (the list is read unconditionally -- without a list, a dummy word of the operand itself -- and selected
    // afterwards: a branch around the load would put a full vmcnt(0) wait into the K loop)
    __device__ __forceinline__ static int pick(const int* __restrict__ idx, const float* __restrict__ X, int i, int lim) {
        const gbl_i32_ptr q = idx ? (gbl_i32_ptr)idx : (gbl_i32_ptr)X;
        const int t = q[(idx && i < lim) ? i : 0];
        return idx ? t : i;
    }

    __device__ __forceinline__ void init(const int* __restrict__ idx, const float* __restrict__ X, int row0, int nrows,
                                         int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            src[i] = KC ? pick(idx, X, row0 + prow(i, tid), nrows) : pick(idx, X, k0 + pk(i, tid), kend);
    }

    __device__ __forceinline__ void load(const float* __restrict__ X, int ld, int row0, int nrows, int k0,
                                         int kend, int tid, const int* __restrict__ idx) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = prow(i, tid), k = pk(i, tid);
            const int gr = row0 + r, gk = k0 + k;
            const size_t off = KC ? (size_t)src[i] * ld + gk : (size_t)src[i] * ld + gr;
            const gbl_f32_ptr Xg = (gbl_f32_ptr)X;
            if (VEC) {  // contiguous extent is a multiple of 4: a float4 is all-valid or all-invalid
                const bool ok = gr < nrows && gk < kend;
                const f32x4 x = *(gbl_f32x4_ptr)(Xg + (ok ? off : 0));
                v[i] = ok ? make_float4(x.x, x.y, x.z, x.w) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool ok = KC ? (gr < nrows && gk + c < kend) : (gk < kend && gr + c < nrows);
                    const float t = Xg[ok ? off + c : 0];
                    e[c] = ok ? t : 0.f;
                }
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        if (!KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) src[i] = pick(idx, X, k0 + 32 + pk(i, tid), kend);
        }
    }

    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = prow(i, tid), k = pk(i, tid);
            const int off = KC ? r * 36 + k : k * ROWS + r;
            *reinterpret_cast<float4*>(lds + off) = v[i];
        }
    }

    // split mode: park the tile as three bf16 planes (layouts above)
    __device__ __forceinline__ void store_split(float* lds, int tid) const {
        unsigned* l32 = reinterpret_cast<unsigned*>(lds);
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = prow(i, tid), k = pk(i, tid);
                float h[4], m[4], l[4];
                split3(v[i].x, h[0], m[0], l[0]); split3(v[i].y, h[1], m[1], l[1]);
                split3(v[i].z, h[2], m[2], l[2]); split3(v[i].w, h[3], m[3], l[3]);
                unsigned* d = l32 + (r * SPLIT_RS + k) / 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
                *reinterpret_cast<uint2*>(d + 16) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
                *reinterpret_cast<uint2*>(d + 32) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
            }
        } else {
            static_assert(KC || NV == 4, "row-contiguous split tiles: a thread holds one k-quad of four rows");
            const int q = prow(0, tid) >> 2, kq = pk(0, tid) >> 2;
            float hh[4][4], mm[4][4], ll[4][4];  // [k][row]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) split3(e[c], hh[i][c], mm[i][c], ll[i][c]);
            }
            constexpr int QS = split_qs(ROWS), PL = split_rc_plane(ROWS);
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // row 4 q + c: its k-quad as two bf16 pairs
                unsigned* d = l32 + ((kq * 4 + c) * QS + q) * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(hh[0][c], hh[1][c]), pack_hi16(hh[2][c], hh[3][c]));
                *reinterpret_cast<uint2*>(d + PL) = make_uint2(pack_hi16(mm[0][c], mm[1][c]), pack_hi16(mm[2][c], mm[3][c]));
                *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(pack_hi16(ll[0][c], ll[1][c]), pack_hi16(ll[2][c], ll[3][c]));
            }
        }
    }
};

// fragment of plane p for k-step s (16 k) of a split tile: the 8 bf16 k = 16 s + 8 h .. + 7 of one row
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 read_frag_split(const float* lds, int row, int p, int s, int h) {
    const unsigned* l32 = reinterpret_cast<const unsigned*>(lds);
    uint4 u;
    if (KC) u = *reinterpret_cast<const uint4*>(l32 + (row * SPLIT_RS + p * 32 + 16 * s + 8 * h) / 2);
    else {
        constexpr int QS = split_qs(ROWS), PL = split_rc_plane(ROWS);
        const unsigned* q = l32 + p * PL + (((4 * s + 2 * h) * 4 + (row & 3)) * QS + (row >> 2)) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(q), hi = *reinterpret_cast<const uint2*>(q + 4 * QS * 2);
        u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    return __builtin_bit_cast(bf16x8, u);
}

// LDS-DMA staging (global_load_lds_dwordx4): a wave-instruction writes 1 KiB of LDS linearly (wave-uniform base +
// lane * 16 B) from per-lane global addresses, with no VGPR round trip and no ds_write (the VGPR->LDS transfer
// path delivers only ~80 B/clk/CU, MI355X_MICROARCH.md LDS table: for an fp32 tile that write phase was ~15 % of
// a K-step).  The LDS image must therefore be unpadded; the K-contiguous image [rows][32] avoids bank conflicts
// with an XOR swizzle of the 16-B chunk index by (row >> 1) & 7, applied to the SOURCE address here and to the
// fragment read (cdna_hip_programming.md rule 21: both sides or neither).  Out-of-range rows are clamped (their
// products land in rows / columns that are never stored); K must be a multiple of 32 (no zero fill).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

// Row-contiguous (!KC) tiles take the source k-row of each of this lane's pieces from krow[] (DmaRows below): the
// K range may end inside the tile (rows past it are clamped duplicates; the caller zeroes the A fragment there) and
// may run through a row list.
template <bool KC, int ROWS, int NT>
struct DmaRows {
    static constexpr int PIECES = ROWS * 32 / 256;               // 1-KiB pieces of the [ROWS x 32] tile
    static constexpr int PER = (PIECES + NT / 64 - 1) / (NT / 64);  // pieces per wave
    int krow[PER];
    // rows of the K-tile at k0, clamped to the last live k-row (the DMA loop waits for all of a wave's loads at its
    // barrier anyway, so the list may sit behind a uniform branch here)
    __device__ __forceinline__ void fetch(const int* __restrict__ idx, int k0, int kend, int tid) {
        if (KC) return;
        constexpr int LPR = ROWS / 4;  // lanes per k-row
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int pc = wave + j * (NT / 64);
            krow[j] = max(min(k0 + pc * (64 / LPR) + lane / LPR, kend - 1), 0);
        }
        if (idx) {
#pragma unroll
            // (a generic pointer on purpose: the flat_load counts on lgkmcnt and leaves the vmcnt queue to the
            // DMAs; through the global address space this loop measured 6 % slower)
            for (int j = 0; j < PER; ++j) krow[j] = idx[krow[j]];
        }
    }
};

template <bool KC, int ROWS, int NT>
__device__ __forceinline__ void dma_tile(const float* __restrict__ X, int ld, int row0, int nrows, int k0, float* stage,
                                         int tid, const DmaRows<KC, ROWS, NT>& rows) {
    constexpr int PIECES = ROWS * 32 / 256;
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < DmaRows<KC, ROWS, NT>::PER; ++j) {
        const int pc = wave + j * (NT / 64);
        if constexpr (PIECES % (NT / 64) != 0) {  // (otherwise every wave has exactly PER pieces: no branch in the K loop)
            if (pc >= PIECES) break;
        }
        const float* src;
        if (KC) {
            const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            src = X + (size_t)min(row0 + r, nrows - 1) * ld + k0 + c * 4;
        } else {
            constexpr int LPR = ROWS / 4;
            const int cc = (lane % LPR) * 4;
            src = X + (size_t)rows.krow[j] * ld + min(row0 + cc, nrows - 4);
        }
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(stage + pc * 256), 16, 0, 0);
    }
}

template <bool KC, int ROWS>
__device__ __forceinline__ float4 read_frag_swz(const float* lds, int row, int s, int h) {
    if (KC) return *reinterpret_cast<const float4*>(lds + row * 32 + (((s * 2 + h) ^ ((row >> 1) & 7)) << 2));
    const float* p = lds + (s * 8 + h * 4) * ROWS + row;
    return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
}

template <bool KC, int ROWS>
__device__ __forceinline__ float4 read_frag(const float* lds, int row, int s, int h) {
    if (KC) return *reinterpret_cast<const float4*>(lds + row * 36 + s * 8 + h * 4);
    const float* p = lds + (s * 8 + h * 4) * ROWS + row;
    return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
}

struct GemmParams {
    int M, N, K, lda, ldb, ldc, ldaux, epi, splitk, kchunk, vecA, vecB, vecC, ragged;
    // tile -> workgroup map (see tile_coords): tile counts, super-tile shape (SX*SY*SZ = 64), super-tile counts
    int ntx, nty, ntz, SX, SY, SZ, cx, cy, cz, rowslow, nsuper, nlin;
    unsigned long long* stamps;  // diagnostics: 4 s_memtime stamps per workgroup (tools/gemm_stamps.py), else null
    const GemmGroup* groups;
    GemmGroup g0;
};

// XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id and each XCD has its own
// 4 MiB L2, so with the plain (x fastest) order the column tiles that share an A tile sit on different
// XCDs and every operand tile is fetched over the fabric once per consumer (measured with FETCH_SIZE: 3x
// the algorithmic read bytes).  Here the tile grid is cut into super-tiles of 64 tiles (SX x SY x SZ,
// neighbours in n, m and group/split); super-tile s belongs to XCD s % 8, and the 64 workgroups that
// XCD receives back to back are exactly that super-tile, so shared operand tiles meet in one L2.
__device__ __forceinline__ bool tile_coords(const GemmParams& p, int lin, int& bx, int& by, int& bz) {
    if (p.nsuper == 0) {  // plain order
        bx = lin % p.ntx; by = (lin / p.ntx) % p.nty; bz = lin / (p.ntx * p.nty);
        return true;
    }
    const int xcd = lin & 7, k = lin >> 3;
    const int s = (k >> 6) * 8 + xcd, within = k & 63;
    if (s >= p.nsuper) return false;
    int sx, sy, sz;
    if (p.rowslow) {
        // ragged rows: row super-tiles slowest, so the dead row tiles of every group (those past its live count)
        // come last in the launch and all XCDs run out of live work together
        sz = s % p.cz; sx = (s / p.cz) % p.cx; sy = s / (p.cz * p.cx);
    } else {
        sx = s % p.cx; sy = (s / p.cx) % p.cy; sz = s / (p.cx * p.cy);
    }
    bx = sx * p.SX + within % p.SX;
    by = sy * p.SY + (within / p.SX) % p.SY;
    bz = sz * p.SZ + within / (p.SX * p.SY);
    return bx < p.ntx && by < p.nty && bz < p.ntz;
}

// The group table is read-only for the kernel: reading it through the constant address space lets the compiler
// use scalar loads (SGPR pointers) wherever it is read, also after stores to C.
__device__ __forceinline__ GemmGroup load_group(const GemmParams& p, int gi) {
    if (!p.groups) return p.g0;
    static_assert(sizeof(GemmGroup) == 10 * sizeof(void*), "GemmGroup is ten pointers");
    typedef const unsigned long long __attribute__((address_space(4))) * const_u64_ptr;
    const_u64_ptr q = (const_u64_ptr)(uintptr_t)(p.groups + __builtin_amdgcn_readfirstlane(gi));
    GemmGroup g;
    g.A = reinterpret_cast<const float*>(q[0]);
    g.B = reinterpret_cast<const float*>(q[1]);
    g.C = reinterpret_cast<float*>(q[2]);
    g.bias = reinterpret_cast<const float*>(q[3]);
    g.aux = reinterpret_cast<const float*>(q[4]);
    g.csum = reinterpret_cast<float*>(q[5]);
    g.cnt = reinterpret_cast<const int*>(q[6]);
    g.aidx = reinterpret_cast<const int*>(q[7]);
    g.bidx = reinterpret_cast<const int*>(q[8]);
    g.cidx = reinterpret_cast<const int*>(q[9]);
    return g;
}

template <bool A_KC, bool B_KC, bool VEC, int WM, int WN, int TM, int TN, int WPE, bool DB, bool DMA, bool SPLIT = false>
__global__ __launch_bounds__(WM * WN * 64, WPE) void gemm_kernel(GemmParams p) {
    static_assert(!DMA || (DB && VEC), "the LDS-DMA path is the two-stage, 16-B aligned one");
    static_assert(!SPLIT || !DMA, "split mode converts in registers");
    static_assert(WM * WN <= 8, "at most 8 waves");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int LDS_A = SPLIT ? (A_KC ? BM * SPLIT_RS / 2 : 3 * split_rc_plane(BM)) : (A_KC ? BM * (DMA ? 32 : 36) : 32 * BM);
    constexpr int LDS_B = SPLIT ? (B_KC ? BN * SPLIT_RS / 2 : 3 * split_rc_plane(BN)) : (B_KC ? BN * (DMA ? 32 : 36) : 32 * BN);
    constexpr int STAGE = LDS_A + LDS_B;
    // (the epilogue stages 32-row bands of the C tile in the same array: keep room for four of them)
    constexpr int LDS_FLOATS = (DB ? 2 : 1) * STAGE > 4 * 32 * (BN + 4) || !DMA ? (DB ? 2 : 1) * STAGE : 4 * 32 * (BN + 4);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* ldsA = lds;
    float* ldsB = lds + LDS_A;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    int bx, by, bz;
    if (!tile_coords(p, blockIdx.x, bx, by, bz)) return;  // padding of the last super-tiles (uniform per workgroup)
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();
    const int gidx = bz / p.splitk, ks = bz % p.splitk;
    const GemmGroup g = load_group(p, gidx);
    const int m0 = by * BM, n0 = bx * BN;
    int kchunk = p.kchunk;
    // Ragged launch: the group's live extent is a device-side count (rows of C for NT / NN, K for TN) and rows may
    // go through index lists.  Tiles wholly past the count leave at once (uniform per workgroup).
    int Mrows = p.M, Kext = p.K;
    const int *aidx = nullptr, *bidx = nullptr, *cidx = nullptr;
    if (p.ragged) {
        if (p.ragged & RAG_CNT) {
            typedef const int __attribute__((address_space(4))) * const_int_ptr;
            const int c = *(const_int_ptr)(uintptr_t)g.cnt;
            if (A_KC) Mrows = min(Mrows, c);
            else {  // split the LIVE K evenly over the K-splits (whole K-tiles)
                Kext = min(Kext, c);
                kchunk = ((Kext + p.splitk - 1) / p.splitk + 31) & ~31;
            }
        }
        if (p.ragged & RAG_AIDX) aidx = g.aidx;
        if (p.ragged & RAG_BIDX) bidx = g.bidx;
        if (p.ragged & RAG_CIDX) cidx = g.cidx;
    }
    const int kbeg = ks * kchunk;
    if (p.ragged && (m0 >= Mrows || kbeg >= Kext)) return;  // (TN ragged launches are atomic: nothing to add)
    const int kend = min(Kext, kbeg + kchunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    TileLoader<A_KC, BM, NT, VEC, SPLIT> la;
    TileLoader<B_KC, BN, NT, VEC, SPLIT> lb;
    if constexpr (!DMA) {
        la.init(aidx, g.A, m0, Mrows, kbeg, kend, tid);
        lb.init(bidx, g.B, n0, p.N, kbeg, kend, tid);
    }
    auto compute = [&](const float* sa, const float* sb, int klim = 32) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int x = 0; x < TM; ++x) {
                a[x] = DMA ? read_frag_swz<A_KC, BM>(sa, (wm * TM + x) * 32 + li, s, lh)
                           : read_frag<A_KC, BM>(sa, (wm * TM + x) * 32 + li, s, lh);
                if (DMA && !A_KC) {  // k-rows past the live K hold clamped duplicates: they must not count
                    const int kf = s * 8 + lh * 4;
                    a[x].x = kf < klim ? a[x].x : 0.f; a[x].y = kf + 1 < klim ? a[x].y : 0.f;
                    a[x].z = kf + 2 < klim ? a[x].z : 0.f; a[x].w = kf + 3 < klim ? a[x].w : 0.f;
                }
            }
#pragma unroll
            for (int y = 0; y < TN; ++y)
                b[y] = DMA ? read_frag_swz<B_KC, BN>(sb, (wn * TN + y) * 32 + li, s, lh)
                           : read_frag<B_KC, BN>(sb, (wn * TN + y) * 32 + li, s, lh);
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].x, b[y].x, acc[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].y, b[y].y, acc[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].z, b[y].z, acc[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].w, b[y].w, acc[x][y], 0, 0, 0);
                }
        }
    };
    f32x16 acc2[SPLIT ? TM : 1][SPLIT ? TN : 1];  // split mode: the five correction products, kept apart from hi*hi
    if constexpr (SPLIT) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
    }
    auto compute_split = [&](const float* sa, const float* sb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[x][pl] = read_frag_split<A_KC, BM>(sa, (wm * TM + x) * 32 + li, pl, s, lh);
#pragma unroll
            for (int y = 0; y < TN; ++y)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[y][pl] = read_frag_split<B_KC, BN>(sb, (wn * TN + y) * 32 + li, pl, s, lh);
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y) {  // smallest terms first
                    acc2[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][2], b[y][0], acc2[x][y], 0, 0, 0);
                    acc2[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][2], acc2[x][y], 0, 0, 0);
                    acc2[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][1], acc2[x][y], 0, 0, 0);
                    acc2[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][0], acc2[x][y], 0, 0, 0);
                    acc2[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][1], acc2[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][0], acc[x][y], 0, 0, 0);
                }
        }
    };
    if constexpr (SPLIT && DB) {
        // two LDS stages, one barrier per K-tile: the next tile's registers are split and parked in the other stage
        // in the shadow of this tile's MFMAs
        if (kbeg < kend) {
            la.load(g.A, p.lda, m0, Mrows, kbeg, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, kbeg, kend, tid, bidx);
        }
        la.store_split(ldsA, tid);
        lb.store_split(ldsB, tid);
        __syncthreads();
        int cur = 0, k0 = kbeg;
        for (; k0 + 32 < kend; k0 += 32) {
            la.load(g.A, p.lda, m0, Mrows, k0 + 32, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, k0 + 32, kend, tid, bidx);
            compute_split(ldsA + cur * STAGE, ldsB + cur * STAGE);
            la.store_split(ldsA + (cur ^ 1) * STAGE, tid);
            lb.store_split(ldsB + (cur ^ 1) * STAGE, tid);
            __syncthreads();
            cur ^= 1;
        }
        if (k0 < kend) {
            compute_split(ldsA + cur * STAGE, ldsB + cur * STAGE);
            __syncthreads();
        }
    } else if constexpr (SPLIT) {
        if (kbeg < kend) {
            la.load(g.A, p.lda, m0, Mrows, kbeg, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, kbeg, kend, tid, bidx);
        }
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            la.store_split(ldsA, tid);
            lb.store_split(ldsB, tid);
            __syncthreads();
            if (k0 + 32 < kend) {
                la.load(g.A, p.lda, m0, Mrows, k0 + 32, kend, tid, aidx);
                lb.load(g.B, p.ldb, n0, p.N, k0 + 32, kend, tid, bidx);
            }
            compute_split(ldsA, ldsB);
            __syncthreads();
        }
    } else if constexpr (DMA) {
        // two LDS stages filled by LDS-DMA: the next K-tile streams into the other stage while this one is
        // multiplied; __syncthreads() waits for this wave's DMAs (vmcnt) before the barrier.  The source rows of
        // the tile after next are fetched one step ahead (ragged launches: through the row lists).
        DmaRows<A_KC, BM, NT> ra;
        DmaRows<B_KC, BN, NT> rb;
        ra.fetch(aidx, kbeg, kend, tid);
        rb.fetch(bidx, kbeg, kend, tid);
        if (kbeg < kend) {
            dma_tile<A_KC, BM, NT>(g.A, p.lda, m0, Mrows, kbeg, ldsA, tid, ra);
            dma_tile<B_KC, BN, NT>(g.B, p.ldb, n0, p.N, kbeg, ldsB, tid, rb);
        }
        ra.fetch(aidx, kbeg + 32, kend, tid);
        rb.fetch(bidx, kbeg + 32, kend, tid);
        __syncthreads();
        if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();
        int cur = 0, k0 = kbeg;
        for (; k0 + 32 <= kend; k0 += 32) {  // whole K-tiles: no masking code in this loop
            if (k0 + 32 < kend) {
                dma_tile<A_KC, BM, NT>(g.A, p.lda, m0, Mrows, k0 + 32, ldsA + (cur ^ 1) * STAGE, tid, ra);
                dma_tile<B_KC, BN, NT>(g.B, p.ldb, n0, p.N, k0 + 32, ldsB + (cur ^ 1) * STAGE, tid, rb);
                ra.fetch(aidx, k0 + 64, kend, tid);
                rb.fetch(bidx, k0 + 64, kend, tid);
            }
            compute(ldsA + cur * STAGE, ldsB + cur * STAGE);
            __syncthreads();
            cur ^= 1;
        }
        if (k0 < kend) {  // the K range ends inside this tile
            compute(ldsA + cur * STAGE, ldsB + cur * STAGE, kend - k0);
            __syncthreads();
        }
    } else if constexpr (DB) {
        if (kbeg < kend) {
            la.load(g.A, p.lda, m0, Mrows, kbeg, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, kbeg, kend, tid, bidx);
        }
        // two LDS stages, one barrier per K-tile: the next tile's registers are parked in the other
        // stage after this tile's MFMAs, so the global-load latency hides behind them
        la.store(ldsA, tid);
        lb.store(ldsB, tid);
        __syncthreads();
        if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();
        int cur = 0, k0 = kbeg;
        for (; k0 + 32 < kend; k0 += 32) {  // straight-line body (a branch per K-tile cost 5 % in the DMA loop)
            la.load(g.A, p.lda, m0, Mrows, k0 + 32, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, k0 + 32, kend, tid, bidx);
            compute(ldsA + cur * STAGE, ldsB + cur * STAGE);
            la.store(ldsA + (cur ^ 1) * STAGE, tid);
            lb.store(ldsB + (cur ^ 1) * STAGE, tid);
            __syncthreads();
            cur ^= 1;
        }
        if (k0 < kend) {  // the last K-tile
            compute(ldsA + cur * STAGE, ldsB + cur * STAGE);
            __syncthreads();
        }
    } else {
        if (kbeg < kend) {
            la.load(g.A, p.lda, m0, Mrows, kbeg, kend, tid, aidx);
            lb.load(g.B, p.ldb, n0, p.N, kbeg, kend, tid, bidx);
        }
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            la.store(ldsA, tid);
            lb.store(ldsB, tid);
            __syncthreads();
            if (k0 + 32 < kend) {
                la.load(g.A, p.lda, m0, Mrows, k0 + 32, kend, tid, aidx);
                lb.load(g.B, p.ldb, n0, p.N, k0 + 32, kend, tid, bidx);
            }
            compute(ldsA, ldsB);
            __syncthreads();
        }
    }

    if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SPLIT) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += acc2[a][b][r];
    }
    const int epi = p.epi;
    constexpr int LDE = BN + 4;                              // padded row length of the staged tile
    constexpr int BANDS = WM * TM;                           // 32-row bands of the block tile
    constexpr int GMAX = LDS_FLOATS / (32 * LDE);            // bands that fit in LDS at once
    constexpr int G = GMAX >= BANDS ? BANDS : (GMAX >= 4 ? 4 : (GMAX >= 2 ? 2 : (GMAX >= 1 ? 1 : 0)));
    if constexpr (G > 0 && BANDS % (G > 0 ? G : 1) == 0 && (G * 32 * (BN / 4)) % NT == 0) {
        if (p.vecC) {
            // Vectorised epilogue: stage G 32-row bands through LDS, then each thread handles float4
            // pieces of whole rows (16 B per lane, 512 B contiguous per 32 lanes).  A thread's pieces all
            // sit in the same 4 columns, so its bias is loaded once; the ReLU-gate activations of a whole
            // pass are requested BEFORE the staging barrier (the stores to C may alias them as far as the
            // compiler knows, so it cannot hoist these loads itself and would pay one memory round trip
            // per piece).
            constexpr int PER = G * 32 * (BN / 4) / NT;
            constexpr int CH = PER < 2 ? PER : 2;  // float4 pieces of old C in flight per thread
            constexpr bool PF = TM * TN <= 2;      // registers for the whole pass's gate loads (else per round)
            static_assert(NT % (BN / 4) == 0, "a thread keeps its columns across pieces");
            const int c4 = tid % (BN / 4), col = n0 + c4 * 4;
            const bool colok = col < p.N;
            float cs4[4] = {0.f, 0.f, 0.f, 0.f};  // EPI_COLSUM: this thread's share of the column sums
            float bias4[4] = {0.f, 0.f, 0.f, 0.f};
            if ((epi & EPI_BIAS) && colok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bias4[c] = gload(g.bias + col + c);
            }
#pragma unroll
            for (int pi = 0; pi < BANDS / G; ++pi) {
                float4 ax[PER];
                if (PF && (epi & EPI_MASK)) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int lr = (tid + i * NT) / (BN / 4);
                        const int row = m0 + (pi * G + (lr >> 5)) * 32 + (lr & 31);
                        ax[i] = gload4(g.aux + ((colok && row < Mrows) ? (size_t)row * p.ldaux + col : 0));
                    }
                }
#pragma unroll
                for (int x = 0; x < TM; ++x) {
                    const int band = wm * TM + x;
                    if (band / G == pi) {
#pragma unroll
                        for (int y = 0; y < TN; ++y)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                lds[((band % G) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDE + (wn * TN + y) * 32 + li] =
                                    acc[x][y][r];
                    }
                }
                __syncthreads();
                if (PF && (epi & EPI_MASK)) {
                    // Keep the gate values as raw registers until here: without this the compiler turns them into
                    // lane masks right after the loads, i.e. waits for the loads BEFORE the staging above and the
                    // barrier, and the memory round trip they were meant to hide is back on the critical path.
#pragma unroll
                    for (int i = 0; i < PER; ++i)
                        asm volatile("" : "+v"(ax[i].x), "+v"(ax[i].y), "+v"(ax[i].z), "+v"(ax[i].w));
                }
#pragma unroll
                for (int i0 = 0; i0 < PER; i0 += CH) {
                    float4 v[CH], cold[CH];
                    bool ok[CH];
                    size_t off[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int lr = (tid + (i0 + i) * NT) / (BN / 4);
                        const int row = m0 + (pi * G + (lr >> 5)) * 32 + (lr & 31);
                        ok[i] = colok && row < Mrows;
                        const int crow = cidx ? ((gbl_i32_ptr)cidx)[ok[i] ? row : 0] : row;
                        off[i] = ok[i] ? (size_t)crow * p.ldc + col : 0;
                        v[i] = *reinterpret_cast<const float4*>(lds + lr * LDE + c4 * 4);
                        if (epi & EPI_ADDC) cold[i] = gload4(g.C + off[i]);
                        if (!PF && (epi & EPI_MASK))
                            ax[i0 + i] = gload4(g.aux + (ok[i] ? (size_t)row * p.ldaux + col : 0));
                    }
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        if (!ok[i]) continue;
                        float* cp = g.C + off[i];
                        float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                        if (epi & EPI_ATOMIC) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) gatomic_add(cp + c, e[c]);
                            continue;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] += bias4[c];
                        if (epi & EPI_ADDC) { e[0] += cold[i].x; e[1] += cold[i].y; e[2] += cold[i].z; e[3] += cold[i].w; }
                        if (epi & EPI_RELU) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) e[c] = fmaxf(e[c], 0.f);
                        }
                        if (epi & EPI_SOFTPLUS) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) e[c] = softplus_f(e[c]);
                        }
                        if (epi & EPI_MASK) {
                            const float4 a4 = ax[i0 + i];
                            e[0] = a4.x > 0.f ? e[0] : 0.f; e[1] = a4.y > 0.f ? e[1] : 0.f;
                            e[2] = a4.z > 0.f ? e[2] : 0.f; e[3] = a4.w > 0.f ? e[3] : 0.f;
                        }
                        gstore4(cp, e[0], e[1], e[2], e[3]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) cs4[c] += e[c];
                    }
                }
                if (pi + 1 < BANDS / G) __syncthreads();
            }
            if (epi & EPI_COLSUM) {
                // the NT / (BN/4) threads that share 4 columns meet in LDS; one atomic per column and tile
                static_assert(NT * 4 <= LDS_FLOATS, "column-sum staging fits the tile's LDS");
                __syncthreads();
                *reinterpret_cast<float4*>(lds + (tid / (BN / 4)) * BN + c4 * 4) = make_float4(cs4[0], cs4[1], cs4[2], cs4[3]);
                __syncthreads();
                if (tid < BN && n0 + tid < p.N) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < NT / (BN / 4); ++k) t += lds[k * BN + tid];
                    gatomic_add(g.csum + n0 + tid, t);
                }
            }
            if (p.stamps && tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are out
                unsigned long long* o = p.stamps + 4ull * blockIdx.x;
                o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
            }
            return;
        }
    }
    // scalar epilogue (unaligned C / ragged N): lanes 0..31 of a half hold 32 consecutive columns of one row
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + (wn * TN + y) * 32 + li;
            if (col >= p.N) continue;
            const float bias = (epi & EPI_BIAS) ? gload(g.bias + col) : 0.f;
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= Mrows) continue;
                float* cp = g.C + (size_t)(cidx ? ((gbl_i32_ptr)cidx)[row] : row) * p.ldc + col;
                float v = acc[x][y][r];
                if (epi & EPI_ATOMIC) {
                    gatomic_add(cp, v);
                    continue;
                }
                v += bias;
                if (epi & EPI_ADDC) v += gload(cp);
                if (epi & EPI_RELU) v = fmaxf(v, 0.f);
                if (epi & EPI_SOFTPLUS) v = softplus_f(v);
                if (epi & EPI_MASK) v = gload(g.aux + (size_t)row * p.ldaux + col) > 0.f ? v : 0.f;
                gstore(cp, v);
                cs += v;
            }
            if (epi & EPI_COLSUM) gatomic_add(g.csum + col, cs);
        }
}

unsigned long long* g_gemm_stamps = nullptr;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int g_gemm_split = -1;  // -1: not set (SHIVAE_GEMM_SPLIT decides at the first launch), 0 / 1: shivae_set_gemm_split
bool gemm_split_on() {
    if (g_gemm_split < 0) { const char* f = getenv("SHIVAE_GEMM_SPLIT"); g_gemm_split = f ? atoi(f) != 0 : 0; }
    return g_gemm_split > 0;
}

template <bool A_KC, bool B_KC, bool VEC, int WM, int WN, int TM, int TN, int MINB = (TM * TN <= 2 ? 4 : (TM * TN <= 4 ? 3 : 2)), bool DB = false,
          bool DMA = false, bool SPLIT = false>
static int launch_cfg(const GemmParams& p, int ngroups, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int layout = A_KC ? (B_KC ? GEMM_NT : GEMM_NN) : GEMM_TN;
    constexpr int tile = SPLIT ? 5 : (BM == 256 && BN == 32) ? 0 : (BM == 32 && BN == 256) ? 1 : (BM == 256 && BN == 64) ? 2
                       : (BM == 64 && BN == 256) ? 3 : 4;
    (void)tile;
    GemmParams q = p;
    q.stamps = g_gemm_stamps;
    q.ntx = ceil_div(p.N, BN); q.nty = ceil_div(p.M, BM); q.ntz = ngroups * p.splitk;
    const long long tiles = (long long)q.ntx * q.nty * q.ntz;
    static const bool swizzle = [] { const char* f = getenv("SHIVAE_GEMM_SWIZZLE"); return !f || atoi(f) != 0; }();
    long long nblocks = tiles;
    q.nsuper = 0;
    if (swizzle && tiles >= 512) {
        auto pow2_le = [](int v, int cap) { int r = 1; while (r * 2 <= v && r * 2 <= cap) r *= 2; return r; };
        q.SX = pow2_le(q.ntx, 8);
        q.SY = pow2_le(q.nty, 64 / q.SX);
        q.SZ = 64 / (q.SX * q.SY);
        q.cx = ceil_div(q.ntx, q.SX); q.cy = ceil_div(q.nty, q.SY);
        q.cz = ceil_div(q.ntz, q.SZ);
        q.rowslow = (A_KC && (p.ragged & RAG_CNT)) ? 1 : 0;
        const long long ns = (long long)q.cx * q.cy * q.cz;
        SV_REQUIRE(ns < (1ll << 24), "gemm: too many tiles");
        q.nsuper = (int)ns;
        nblocks = (ns + 7) / 8 * 8 * 64;
    }
    SV_REQUIRE(nblocks < (1ll << 31), "gemm grid too large (M=%d groups=%d splitk=%d)", p.M, ngroups, p.splitk);
    q.nlin = (int)nblocks;
    dim3 grid((unsigned)nblocks);
    prof_shape(p.M, p.N, p.K, ngroups);
    prof_begin(layout * 6 + tile, 2.0 * p.M * p.N * (double)p.K * ngroups, stream);
    hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, VEC, WM, WN, TM, TN, MINB, DB, DMA, SPLIT>), grid, dim3(WM * WN * 64), 0, stream, q);
    prof_end(stream);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

template <bool A_KC, bool B_KC, bool VEC>
static int launch_tiles(const GemmParams& p, int ngroups, hipStream_t stream) {
    // tile choice: 128x128 by default; skinny shapes get a 32-wide tile on the skinny side
    if (p.N <= 32) return launch_cfg<A_KC, B_KC, VEC, 4, 1, 2, 1>(p, ngroups, stream);   // 256 x 32
    if (p.M <= 32) return launch_cfg<A_KC, B_KC, VEC, 1, 4, 1, 2>(p, ngroups, stream);   // 32 x 256
    if (p.N <= 64) return launch_cfg<A_KC, B_KC, VEC, 4, 1, 2, 2>(p, ngroups, stream);   // 256 x 64
    if (p.M <= 64) return launch_cfg<A_KC, B_KC, VEC, 1, 4, 2, 2>(p, ngroups, stream);   // 64 x 256
    // default: 128 x 128 on 8 waves (32 x 64 per wave: no spills at 4 waves/SIMD) with two LDS stages.
    // SHIVAE_GEMM_TILE selects the other measured variants (tools/gemm_bench.py).
    static const int mode = [] { const char* f = getenv("SHIVAE_GEMM_TILE"); return f ? atoi(f) : 0; }();
    if (mode == 2) return launch_cfg<A_KC, B_KC, VEC, 4, 2, 1, 2, 4, false>(p, ngroups, stream);  // 8 waves, 1 stage
    if constexpr (VEC) {
        (void)gemm_split_on();
        // split mode: fp32 products as six bf16 MFMAs (see TileLoader); 128 x 128, 8 waves, one LDS stage
        if (g_gemm_split > 0) {
            static const int st = [] { const char* f = getenv("SHIVAE_SPLIT_TILE"); return f ? atoi(f) : 0; }();
            if (st == 1) return launch_cfg<A_KC, B_KC, VEC, 2, 2, 2, 2, 2, false, false, true>(p, ngroups, stream);  // 4 waves of 64x64
            if constexpr (A_KC && B_KC) {
                if (st == 2) return launch_cfg<A_KC, B_KC, VEC, 4, 2, 1, 2, 2, true, false, true>(p, ngroups, stream);   // two LDS stages, 1 WG/CU
            }
            if constexpr (A_KC) {
                // 64 x 256 tiles for N a multiple of 256 (activation operand fetched once per 256 output columns):
                // measured 7 % slower than 128 x 128 in the training step (144.7 against 155.9 TFLOP/s), kept for A/B runs
                if (st == 4 && p.N % 256 == 0) return launch_cfg<A_KC, B_KC, VEC, 2, 4, 1, 2, 4, false, false, true>(p, ngroups, stream);
            }
            // NT: 8 waves of 32 x 64 (both accumulator sets fit 128 registers, two workgroups per CU).  NN / TN: the
            // row-contiguous fragments need more registers and spilled there (64-80 B per lane, TN 150 -> 90 TFLOP/s),
            // so they run 4 waves of 64 x 64 at 256 registers, still two workgroups per CU
            if constexpr (!B_KC) return launch_cfg<A_KC, B_KC, VEC, 2, 2, 2, 2, 2, false, false, true>(p, ngroups, stream);
            else return launch_cfg<A_KC, B_KC, VEC, 4, 2, 1, 2, 4, false, false, true>(p, ngroups, stream);
        }
        // LDS-DMA staging.  TN masks the tail of the last K-tile, so any K and ragged K go; K-contiguous operands
        // need whole K-tiles, and dense rows (no row lists on this path for them).  Measured in the training step
        // (B=256, T=100, straight-line K loops): TN 127 -> 135 TFLOP/s, NN +3 %; NT (both operands K-contiguous)
        // -9 % at K=256 and equal at long K, so NT keeps the register path (SHIVAE_GEMM_DMA bit 2 turns it on).
        static const int dma = [] { const char* f = getenv("SHIVAE_GEMM_DMA"); return f ? atoi(f) : 1; }();
        const bool want = B_KC ? (dma & 2) != 0 : (dma & 1) != 0;
        if (want && mode != 3 && p.M >= 4 && p.N >= 4 && (A_KC ? (!p.ragged && p.K % 32 == 0) : true))
            return launch_cfg<A_KC, B_KC, VEC, 4, 2, 1, 2, 4, true, true>(p, ngroups, stream);    // 128 x 128, 8 waves, LDS-DMA
    }
    return launch_cfg<A_KC, B_KC, VEC, 4, 2, 1, 2, 4, true>(p, ngroups, stream);                  // 128 x 128, 8 waves
}

template <bool A_KC, bool B_KC>
static int launch_layout(const GemmParams& p, int ngroups, hipStream_t stream) {
    if (p.vecA && p.vecB) return launch_tiles<A_KC, B_KC, true>(p, ngroups, stream);
    return launch_tiles<A_KC, B_KC, false>(p, ngroups, stream);
}

int launch_gemm(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                hipStream_t stream) {
    if (d.M <= 0 || d.N <= 0 || ngroups <= 0) return SHIVAE_OK;
    SV_REQUIRE(d.K >= 0 && d.splitk >= 1, "gemm: bad K/splitk");
    SV_REQUIRE(groups_dev != nullptr || ngroups == 1, "gemm: %d groups need a device group table", ngroups);
    SV_REQUIRE(d.splitk == 1 || (d.epi == EPI_ATOMIC), "gemm: split-K needs the atomic epilogue");
    SV_REQUIRE(!(d.epi & EPI_ATOMIC) || d.epi == EPI_ATOMIC, "gemm: atomic epilogue excludes the others");
    SV_REQUIRE(!(d.epi & EPI_COLSUM) || groups_dev != nullptr || g0.csum != nullptr, "gemm: EPI_COLSUM needs csum");
    GemmParams p;
    p.M = d.M; p.N = d.N; p.K = d.K;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.ldaux = d.ldaux;
    p.epi = d.epi; p.splitk = d.splitk; p.ragged = d.ragged;
    SV_REQUIRE(!d.ragged || groups_dev != nullptr, "gemm: ragged launches take a device group table");
    SV_REQUIRE(!d.ragged || d.layout != GEMM_TN || d.epi == EPI_ATOMIC || d.epi == EPI_ADDC, "gemm: a ragged TN launch accumulates into C");
    SV_REQUIRE(!(d.ragged & RAG_BIDX) || d.layout == GEMM_TN, "gemm: a B row list is for TN");
    SV_REQUIRE(!(d.ragged & RAG_CIDX) || d.layout != GEMM_TN, "gemm: a C row list is for NT / NN");
    p.kchunk = round_up(ceil_div(d.K > 0 ? d.K : 1, d.splitk), 32);
    p.groups = groups_dev; p.g0 = g0;
    const bool a_kc = d.layout != GEMM_TN, b_kc = d.layout == GEMM_NT;
    // vector (16-B) global loads need aligned bases/strides and a contiguous extent that is a
    // multiple of 4; group tables are built by the plan, which only marks them vectorisable when
    // every group base is 16-B aligned (checked here for the inline group only).
    const int extA = a_kc ? d.K : d.M, extB = b_kc ? d.K : d.N;
    p.vecA = (d.lda % 4 == 0) && (extA % 4 == 0) && (groups_dev || aligned16(g0.A));
    p.vecB = (d.ldb % 4 == 0) && (extB % 4 == 0) && (groups_dev || aligned16(g0.B));
    if (d.unaligned) p.vecA = p.vecB = 0;
    // vector epilogue: whole float4 pieces of C (and aux) rows must be addressable and 16-B aligned
    p.vecC = (d.ldc % 4 == 0) && (d.N % 4 == 0) && (d.c_aligned || (!groups_dev && aligned16(g0.C))) &&
             (!(d.epi & EPI_MASK) || (d.ldaux % 4 == 0 && (d.c_aligned || (!groups_dev && aligned16(g0.aux)))));
    if (d.layout == GEMM_TN && gemm_split_on() && split_tn_fits(d) &&
        (groups_dev ? d.c_aligned != 0 : (aligned16(g0.A) && aligned16(g0.B) && aligned16(g0.C)))) {
        // split mode, weight-gradient shapes: the 128 x 256 x 16 two-stage kernel of gemm_split.hip
        static const bool on = [] { const char* f = getenv("SHIVAE_SPLIT_TN"); return !f || atoi(f) != 0; }();
        if (on) return launch_gemm_split_tn(d, groups_dev, ngroups, g0, stream);
    }
    switch (d.layout) {
        case GEMM_NT: return launch_layout<true, true>(p, ngroups, stream);
        case GEMM_NN: return launch_layout<true, false>(p, ngroups, stream);
        case GEMM_TN: return launch_layout<false, false>(p, ngroups, stream);
    }
    set_error("gemm: unknown layout %d", d.layout);
    return SHIVAE_ERR_INVALID;
}

// ---------------------------------------------------------------- column sums (bias gradients)
struct ColsumParams {
    int M, N, ld, rows_per_block;
    const ColsumGroup* groups;
    ColsumGroup g0;
};

__global__ __launch_bounds__(256) void colsum_kernel(ColsumParams p) {
    __shared__ float part[4][64];
    const ColsumGroup g = p.groups ? p.groups[blockIdx.y] : p.g0;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
    const int r0 = blockIdx.z * p.rows_per_block, r1 = min(p.M, r0 + p.rows_per_block);
    float s = 0.f;
    if (c < p.N) {
        // (four independent partial sums: the loads of a block are then requested four deep instead of one by one)
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            s += g.Y[(size_t)r * p.ld + c];
            s1 += g.Y[(size_t)(r + 4) * p.ld + c];
            s2 += g.Y[(size_t)(r + 8) * p.ld + c];
            s3 += g.Y[(size_t)(r + 12) * p.ld + c];
        }
        for (; r < r1; r += 4) s += g.Y[(size_t)r * p.ld + c];
        s += s1 + s2 + s3;
    }
    part[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < p.N) atomicAdd(g.out + c, part[0][threadIdx.x] + part[1][threadIdx.x] +
                                                     part[2][threadIdx.x] + part[3][threadIdx.x]);
}

int launch_colsum(int M, int N, int ld, const ColsumGroup* groups_dev, int ngroups, const ColsumGroup& g0,
                  hipStream_t stream) {
    if (M <= 0 || N <= 0 || ngroups <= 0) return SHIVAE_OK;
    SV_REQUIRE(groups_dev != nullptr || ngroups == 1, "colsum: groups need a device table");
    ColsumParams p;
    p.M = M; p.N = N; p.ld = ld;
    // enough blocks to fill the chip, at least 256 rows each -- 64 for a matrix of a few thousand rows, whose launch
    // is a latency chain of one block's row loop (cfg2: 21 launches of 15 us per step with 256-row blocks)
    int zs = ceil_div(2048, ceil_div(N, 64) * ngroups);
    zs = std::max(1, std::min(zs, ceil_div(M, M <= 65536 ? 64 : 256)));
    p.rows_per_block = ceil_div(M, zs);
    zs = ceil_div(M, p.rows_per_block);
    p.groups = groups_dev; p.g0 = g0;
    SV_REQUIRE(ngroups <= 65535 && zs <= 65535, "colsum grid too large");
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), ngroups, zs), dim3(256), 0, stream, p);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
