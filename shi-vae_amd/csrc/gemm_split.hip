// Split-mode NT GEMM against PRE-SPLIT weights: C[M,N] = epilogue(A[M,K] * W[N,K]^T), fp32 in / fp32 out, products on
// the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
//
// The decoder's Linear layers (decoders.py:60-66,127-171) and their input gradients all multiply a tall activation
// matrix by a weight matrix that is a PARAMETER: constant for the whole step and re-used by every row tile.  The
// generic split kernel of gemm.hip re-splits both operands in every workgroup; here the weights are split once per
// step into their three bf16 planes (presplit_kernel), stored in the order the tiles consume them, and then stream
// into LDS by LDS-DMA with no register round trip and no arithmetic.  Only the activation tile is split in the kernel.
//
// Arithmetic (same contract as gemm.hip's split mode, see include/shivae_hip.h shivae_set_gemm_split):
//   x = hi + mid + lo exactly, each term x rounded to the NEAREST bf16 of what is left (v_cvt_pk_bf16_f32);
//   a*b ~ ah*bh  [main accumulator]  +  (ah*bm + am*bh + am*bm + ah*bl + al*bh)  [correction accumulator];
//   the dropped terms (am*bl, al*bm, al*bl) are <= 2^-26 |a b|.  The corrections are <= 2^-8 of the product, so the
//   roundings of THEIR running sum are invisible at fp32 level and the main sum takes one rounding per 16 products:
//   measured error at or below the fp32-input MFMA's (tests/test_gpu_split_mode.py).
//
// Tile: 128 x 256 x 32 per workgroup of 8 waves (2 x 4, 64 x 64 per wave = 2 x 2 MFMA tiles, two accumulator sets =
// 128 registers), one workgroup per CU, two LDS stages of 72 KiB (A 3 x 128 x 64 B + W 3 x 256 x 64 B), one barrier
// per K-tile.  A 256-wide tile covers ALL output columns of the trunk's H = 256 layers, so their activation operand
// is fetched exactly once.
//
// Weight image (global): [KT][3 planes][N][32 bf16], K zero-padded to a multiple of 32: the 256 rows of a column
// tile are 16 KiB contiguous per (K-tile, plane), i.e. every 1-KiB LDS-DMA piece is one contiguous KiB of global memory.
// LDS images (both operands): [plane][row][64 B], the four 16-B chunks of a row XOR-swizzled by (row >> 2) & 3 --
// applied to the DMA's SOURCE address and to the fragment reads (cdna_hip_programming.md rule 21) -- so that the 16
// lanes of a ds_read_b128 lane group hit 16 distinct 4-bank slots.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace shivae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gbl_f32_ptr;
typedef const __attribute__((address_space(1))) f32x4* gbl_f32x4_ptr;
typedef const __attribute__((address_space(1))) int* gbl_i32_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

#ifndef SW_DBG
#define SW_DBG 0  // diagnostic builds (tools/build_variant.sh): 4 no MFMA, 8 no fragment reads (NT), 16 no split / park (TN), 32 no C stores (NT),
                  // 128 one K-tile per tile (NT).  There is NO switch
                  // that drops the DMAs or the activation loads: the loop's hand-counted vmcnt waits count on them; with
                  // fewer operations in flight `vmcnt(8)` stops waiting, the registers of the consumed activation set
                  // are handed out as address temporaries, and the late load lands on an address (GPU fault, seen once)
#endif

constexpr int SW_BM = 128, SW_BN = 256, SW_NT = 512;
constexpr int SW_A_BYTES = 3 * SW_BM * 64;
constexpr int SW_W_BYTES = 3 * SW_BN * 64;
constexpr int SW_STAGE = SW_A_BYTES + SW_W_BYTES;
constexpr int SW_LDS = 2 * SW_STAGE;  // 147 456 B
constexpr int SW_LDE = SW_BN + 4;     // floats per row of the staged C tile (128 x 260 x 4 B = 133 120 B)
static_assert(SW_BM * SW_LDE * 4 <= SW_LDS, "the staged C tile fits the K loop's LDS");

size_t presplit_bytes(int N, int K) { return (size_t)ceil_div(K, 32) * 3 * (size_t)N * 64; }

// ---------------------------------------------------------------- weights -> three bf16 planes, once per step
// One thread per (row n, 8 consecutive k): coalesced fp32 reads along the row, one 16-B store per plane.
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ W, int N, int K, int ld, size_t in_stride,
                                                       unsigned char* __restrict__ img, size_t out_stride) {
    const int KT = (K + 31) >> 5, k8n = KT * 4;
    const float* src = W + blockIdx.y * in_stride;
    unsigned char* dst = img + blockIdx.y * out_stride;
    const size_t total = (size_t)N * k8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / k8n), k8 = (int)(i % k8n), k0 = k8 * 8;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (k0 + c < K) ? src[(size_t)n * ld + k0 + c] : 0.f;
        bf16x8 h, m, l;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const __bf16 hb = (__bf16)v[c];
            const float r = v[c] - (float)hb;
            const __bf16 mb = (__bf16)r;
            h[c] = hb; m[c] = mb; l[c] = (__bf16)(r - (float)mb);
        }
        const int kt = k8 >> 2, c4 = k8 & 3;
        unsigned char* o = dst + ((size_t)kt * 3 * N + n) * 64 + c4 * 16;
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + (size_t)N * 64) = m;
        *reinterpret_cast<bf16x8*>(o + (size_t)2 * N * 64) = l;
    }
}

int launch_presplit(const float* W, int N, int K, int ld, int batch, size_t in_stride, void* img, hipStream_t stream) {
    if (N <= 0 || K <= 0 || batch <= 0) return SHIVAE_OK;
    SV_REQUIRE(batch <= 65535, "presplit: too many matrices");
    SV_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "presplit: image must be 16-B aligned");
    const size_t total = (size_t)N * ceil_div(K, 32) * 4;
    const int gx = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(presplit_kernel, dim3(gx, batch), dim3(256), 0, stream, W, N, K, ld, in_stride,
                       static_cast<unsigned char*>(img), presplit_bytes(N, K));
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// ---------------------------------------------------------------- the GEMM
struct SplitNtParams {
    int M, N, K, lda, ldc, ldaux, epi, ragged, splitk, kt_per_split, KT, Nimg;
    int ntx, nty, nz;
    const GemmGroup* groups;  // GemmGroup::B = the group's weight IMAGE (presplit_kernel), not fp32 weights
    GemmGroup g0;
};

__device__ __forceinline__ GemmGroup sw_load_group(const GemmGroup* groups, const GemmGroup& g0, int gi) {
    if (!groups) return g0;
    typedef const unsigned long long __attribute__((address_space(4))) * const_u64_ptr;
    const_u64_ptr q = (const_u64_ptr)(uintptr_t)(groups + __builtin_amdgcn_readfirstlane(gi));
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

// A 16-B global load the COMPILER DOES NOT TRACK: hipcc answers any register dependency on a load with
// `s_waitcnt vmcnt(0)` while LDS-DMAs are in flight beside it (it takes the two kinds to complete out of order), which
// would make every use of a prefetched activation wait for the DMAs and loads issued after it.  The K loop therefore
// issues these loads by hand and waits for them by hand (sw_wait_loads), counting in issue order.
__device__ __forceinline__ f32x4 sw_load16_untracked(const float* ptr) {
    f32x4 v;
    // (s_nop: the address registers were just written by a VALU instruction; the compiler pads that hazard for its own
    // loads -- `s_nop 0` between v_lshl_add_u64 and global_load -- but does not look inside inline assembly)
    asm volatile("s_nop 1\n\tglobal_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// wait until at most N vector-memory operations of this wave are outstanding; the registers pass through so that no
// use of them can move above the wait
template <int N>
__device__ __forceinline__ void sw_wait_loads(f32x4 (&v)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(N) : "memory");
}

// 16-B LDS read / waits the compiler does not track, same reason (see the K loop).  OFF: the instruction's offset field
template <int OFF>
__device__ __forceinline__ bf16x8 sw_ds_read16(int addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    bf16x8 v;
#if SW_DBG & 8
    asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr) : "memory");
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
#endif
    return v;
}
template <int N>
__device__ __forceinline__ void sw_wait_lds(bf16x8 (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void sw_wait_lds(bf16x8 (&a)[3], bf16x8 (&b)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]),
                   "+v"(b[1][1]), "+v"(b[1][2])
                 : "n"(N)
                 : "memory");
}
// before the barrier: at most the 2 youngest vector-memory operations (the activation loads) and no LDS operation
// outstanding
__device__ __forceinline__ void sw_wait_all_lds(bf16x8 (&a)[2][3], bf16x8 (&b)[2][3]) {
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(b[0][0]),
                   "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])
                 :
                 : "memory");
}

// the four bf16 of a packed vector back as floats, by bit operations on the two packed dwords (left to itself hipcc
// CONVERTS every element a second time, unpacked, to get these: 7.5 instead of 5.5 VALU instructions per split element)
__device__ __forceinline__ f32x4 sw_unpack_bf16x4(bf16x4 h) {
    const uint2 u = __builtin_bit_cast(uint2, h);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}

__device__ __forceinline__ void sw_gstore4(float* p, float a, float b, float c, float d) {
    f32x4 t;
    t.x = a; t.y = b; t.z = c; t.w = d;
    // (nontemporal: an activation tile is next read a whole launch -- gigabytes -- later; kept out of the L2 it leaves
    // the weight image and the activation tiles the column tiles share in place.  Measured +2-4 % on K = 256 .. 528)
    __builtin_nontemporal_store(t, (__attribute__((address_space(1))) f32x4*)p);
}
__device__ __forceinline__ void sw_atomic_add(float* p, float v) {
    __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(SW_NT, 2) void gemm_split_nt_kernel(SplitNtParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[SW_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 31, lh = lane >> 5;

    // XCD-aware order: workgroups go to the 8 XCDs round-robin by linear id; the column tiles that share an
    // activation tile get consecutive slots of ONE XCD, and consecutive row tiles of a group are dealt round the XCDs
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = slot % p.ntx, u = (slot / p.ntx) * 8 + xcd;
    if (u >= p.nty * p.nz) return;
    const int by = u % p.nty, bz = u / p.nty;
    const int gidx = bz / p.splitk, ks = bz % p.splitk;
    const GemmGroup g = sw_load_group(p.groups, p.g0, gidx);
    const int n0 = bx * SW_BN;
    int Mrows = p.M;
    const int *aidx = nullptr, *cidx = nullptr;
    if (p.ragged) {
        if (p.ragged & RAG_CNT) {
            typedef const int __attribute__((address_space(4))) * const_int_ptr;
            Mrows = min(Mrows, *(const_int_ptr)(uintptr_t)g.cnt);
        }
        if (p.ragged & RAG_AIDX) aidx = g.aidx;
        if (p.ragged & RAG_CIDX) cidx = g.cidx;
    }
#if SW_DBG & 128
    const int kt0 = ks * p.kt_per_split, kt1 = min(min(p.KT, kt0 + p.kt_per_split), kt0 + 1);  // one K-tile: the fixed cost of a tile
#else
    const int kt0 = ks * p.kt_per_split, kt1 = min(p.KT, kt0 + p.kt_per_split);
#endif
    if (kt0 >= kt1) return;  // (split-K launches are atomic: nothing to add)

    if (by * SW_BM >= Mrows) return;
    const int m0 = by * SW_BM;
    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[x][y][r] = 0.f; cor[x][y][r] = 0.f; }

    // LDS: [A stage 0 | A stage 1 | W stage 0 | W stage 1]; every address below is ONE lane-dependent register plus a
    // compile-time constant (stage, plane and row-block offsets sit in the instructions' offset fields: the loop keeps
    // 128 accumulator and 72 fragment registers live, there is no room for a register per address)
    constexpr int A_ST = SW_A_BYTES, W_0 = 2 * SW_A_BYTES, W_ST = SW_W_BYTES;

    // ---- activation tile: thread -> rows ar, ar + 64, four consecutive k at 4 * akq
    const int ar = tid >> 3, akq = tid & 7;
    const float* pa[2];
    bool aok[2];
    {
        int rsel[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = m0 + ar + 64 * i;
            aok[i] = r < Mrows;
            rsel[i] = aok[i] ? r : 0;
        }
        if (aidx) {
            // both list entries are requested before either is used (entry 0 for a dead row): written plainly, hipcc
            // sinks each load into its row's `r < Mrows` branch with a vmcnt(0) behind it -- two memory round trips
            // in a row at the head of every tile
            int t0 = ((gbl_i32_ptr)aidx)[rsel[0]], t1 = ((gbl_i32_ptr)aidx)[rsel[1]];
            asm volatile("" : "+v"(t0), "+v"(t1));
            rsel[0] = aok[0] ? t0 : 0;
            rsel[1] = aok[1] ? t1 : 0;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = g.A + (size_t)rsel[i] * p.lda + 4 * akq;
    }
    auto loadA = [&](int kt, f32x4 (&v)[2]) {
        const bool kok = kt * 32 + 4 * akq < p.K;  // K % 4 == 0: a float4 is all-valid or all-invalid
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (always issued -- the K loop's waits count on exactly two loads per call -- and kept RAW: storeA zeroes
            // what is not live)
            v[i] = sw_load16_untracked((aok[i] && kok) ? pa[i] + kt * 32 : g.A);
        }
    };
    // byte offset of this thread's 8-B piece inside one plane of the A image ((r >> 2) & 3 is the same for r, r + 64)
    const int aoff = ar * 64 + (((akq >> 1) ^ ((ar >> 2) & 3)) << 4) + (akq & 1) * 8;
    auto storeA = [&](int stage, const f32x4 (&vr)[2], int kt) {
        const bool kok = kt * 32 + 4 * akq < p.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 t = vr[i];
            const bool ok = aok[i] && kok;
            f32x4 v1;
            v1.x = ok ? t.x : 0.f; v1.y = ok ? t.y : 0.f; v1.z = ok ? t.z : 0.f; v1.w = ok ? t.w : 0.f;
            const bf16x4 h = __builtin_convertvector(v1, bf16x4);
            const f32x4 r = v1 - sw_unpack_bf16x4(h);
            const bf16x4 m = __builtin_convertvector(r, bf16x4);
            const bf16x4 l = __builtin_convertvector(r - sw_unpack_bf16x4(m), bf16x4);
            unsigned char* q = lds + aoff;
            *reinterpret_cast<bf16x4*>(q + (stage * A_ST + i * 4096)) = h;
            *reinterpret_cast<bf16x4*>(q + (stage * A_ST + i * 4096 + SW_BM * 64)) = m;
            *reinterpret_cast<bf16x4*>(q + (stage * A_ST + i * 4096 + 2 * SW_BM * 64)) = l;
        }
    };

    // ---- weight tile by LDS-DMA: 48 pieces of 1 KiB (3 planes x 16 blocks of 16 rows), 6 per wave.  Source address =
    //      wave-uniform piece base + one lane offset (row of the piece, swizzled chunk)
    const unsigned char* Wimg = reinterpret_cast<const unsigned char*>(g.B) + (size_t)n0 * 64;
    const unsigned wlane = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    auto dmaW = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int pc = wid + 8 * j, pl = pc >> 4, blk = pc & 15;
            const unsigned char* src = Wimg + ((size_t)(kt * 3 + pl) * p.Nimg + blk * 16) * 64;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + wlane), (lds_void_ptr)(lds + W_0 + stage * W_ST + pc * 1024),
                                             16, 0, 0);
        }
    };

    // ---- fragments: 8 consecutive k (16 B) of one row and plane.  Chunk c = 2 h + lh of row r sits at
    //      ((c ^ ((r >> 2) & 3)) << 4); (r >> 2) & 3 = (li >> 2) & 3 for every row block, and the second half of a
    //      K-tile (h = 1) is the first half's address with bit 5 flipped
    const int sw = (li >> 2) & 3;
    const int lds0 = (int)(uintptr_t)(lds_void_ptr)lds;
    int fra_at[2], frb_at[2][2];  // [h], [h][stage] (the weight stages are more than one offset field apart)
    fra_at[0] = lds0 + (wm * 64 + li) * 64 + ((lh ^ sw) << 4);
    fra_at[1] = fra_at[0] ^ 32;
    frb_at[0][0] = lds0 + W_0 + (wn * 64 + li) * 64 + ((lh ^ sw) << 4);
    frb_at[1][0] = frb_at[0][0] ^ 32;
    frb_at[0][1] = frb_at[0][0] + W_ST;
    frb_at[1][1] = frb_at[1][0] + W_ST;
    // Fragment registers: ONE set for the activations (24 registers), TWO for the weights (48).  A K-tile is multiplied
    // in two halves of 16 k; the fragments of the next half are requested behind the current half's MFMAs -- the
    // weights into the other set, the activations of row block x once the MFMAs that read block x have been issued.
    // The fragment reads are issued and awaited BY HAND (sw_ds_read16 / sw_wait_lds): while an LDS-DMA is in flight
    // hipcc turns every wait for an LDS read into lgkmcnt(0) -- all reads, the ones just issued for the next half
    // included.  LDS operations complete in issue order, so `lgkmcnt(N)` = "all but my N youngest are done".
    // Measured (B=256, T=100 step, same box): first form (12 reads, lgkmcnt(0), 24 MFMAs, twice per K-tile, barrier at
    // the end of the K-tile) 137 / 162 / 157 / 205 TFLOP/s at K = 256 / 512 / 528 / 131072; this form 143 / 168 /
    // 163 / 219.  Pinning the order written here with sched_barrier (reads strictly ahead of each MFMA block) measured
    // SLOWER than letting the scheduler place the MFMAs around the hand-issued reads (207 at long K): what is left is
    // not LDS latency -- a diagnostic build without any fragment read runs at the same speed -- but the weight DMA's
    // latency (one K-tile in flight is all 2 x 72 KiB allow) beside the per-K-tile barrier.
    bf16x8 fra[2][3], frb[2][2][3];
    auto readA = [&](auto xc, auto stc, auto hc) {
        constexpr int x = decltype(xc)::value, st = decltype(stc)::value, h = decltype(hc)::value;
        fra[x][0] = sw_ds_read16<st * A_ST + x * 2048>(fra_at[h]);
        fra[x][1] = sw_ds_read16<st * A_ST + x * 2048 + SW_BM * 64>(fra_at[h]);
        fra[x][2] = sw_ds_read16<st * A_ST + x * 2048 + 2 * SW_BM * 64>(fra_at[h]);
    };
    auto readB = [&](auto bufc, auto stc, auto hc) {
        constexpr int buf = decltype(bufc)::value, st = decltype(stc)::value, h = decltype(hc)::value;
        frb[buf][0][0] = sw_ds_read16<0>(frb_at[h][st]);
        frb[buf][0][1] = sw_ds_read16<SW_BN * 64>(frb_at[h][st]);
        frb[buf][0][2] = sw_ds_read16<2 * SW_BN * 64>(frb_at[h][st]);
        frb[buf][1][0] = sw_ds_read16<2048>(frb_at[h][st]);
        frb[buf][1][1] = sw_ds_read16<2048 + SW_BN * 64>(frb_at[h][st]);
        frb[buf][1][2] = sw_ds_read16<2048 + 2 * SW_BN * 64>(frb_at[h][st]);
    };
    auto block = [&](int x, int buf) {
#if SW_DBG & 4
        return;
#endif
#pragma unroll
        for (int y = 0; y < 2; ++y) {  // corrections smallest first; the leading product on its own
            cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][2], frb[buf][y][0], cor[x][y], 0, 0, 0);
            cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][0], frb[buf][y][2], cor[x][y], 0, 0, 0);
            cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][1], frb[buf][y][1], cor[x][y], 0, 0, 0);
            cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][1], frb[buf][y][0], cor[x][y], 0, 0, 0);
            cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][0], frb[buf][y][1], cor[x][y], 0, 0, 0);
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[x][0], frb[buf][y][0], acc[x][y], 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> c0;
    typedef std::integral_constant<int, 1> c1;

    // ---- K loop: two LDS stages, ONE barrier per K-tile, placed BETWEEN the two halves of the tile's MFMA work:
    //   first half of K-tile k  : MFMAs on (k, first 16) | reads fragments (k, second 16) | splits and stores the
    //                             activations of K-tile k+1 into the other stage
    //   wait (my DMAs of k+1 landed, my LDS traffic done), barrier: stage k+1 is complete, stage k is dead
    //   second half             : requests K-tile k+2's weights (DMA into the dead stage) and K-tile k+3's activations,
    //                             MFMAs on (k, second 16) | reads fragments (k+1, first 16) from the other stage
    // Activations travel through registers, requested two K-tiles before they are split (two register sets that swap
    // roles); in every request group the 6 DMAs are issued before the 2 loads, so `vmcnt(8)` means "the loads of the
    // group before the last have landed" and `vmcnt(2)` "the DMAs of the last group have landed".  All waits on
    // vector memory inside the loop are written by hand (see sw_load16_untracked).  The stage is a compile-time
    // constant of each copy of the body (K-tiles alternate between the two copies).
    // Fragment reads in issue order per K-tile: [B h1: 6][A0 h1: 3][A1 h1: 3] barrier [B' h0: 6][A0' h0: 3][A1' h0: 3]
    f32x4 va[2], vb[2];
    dmaW(kt0, 0);
    loadA(kt0, vb);
    loadA(kt0 + 1, va);
    sw_wait_loads<2>(vb);  // (vb is younger than the DMAs: they have landed too; va's two loads fly on)
    storeA(0, vb, kt0);
    if (kt0 + 1 < kt1) dmaW(kt0 + 1, 1);
    loadA(kt0 + 2, vb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (bare: __syncthreads() carries a fence that waits for ALL vector-memory operations)
    asm volatile("" ::: "memory");
    readB(c0(), c0(), c0());
    readA(c0(), c0(), c0());
    readA(c1(), c0(), c0());
    // vs: the activations of K-tile k+1; leaves with those of k+3 requested.  ST: the stage K-tile k lives in
    auto body = [&](int k, f32x4 (&vs)[2], auto stc) {
        constexpr int ST = decltype(stc)::value;
        typedef std::integral_constant<int, ST ^ 1> nxc;
        readB(c1(), stc, c1());
        sw_wait_lds<9>(fra[0], frb[0]);  // (the 9 youngest: A1 of this half, B of the next)
        block(0, 0);
        readA(c0(), stc, c1());
        sw_wait_lds<9>(fra[1]);          // (the 9 youngest: B and A0 of the next half)
        sw_wait_loads<8>(vs);
        storeA(ST ^ 1, vs, k + 1);       // (VALU work the scheduler may spread between these MFMAs)
        block(1, 0);
        readA(c1(), stc, c1());
        // my DMAs of K-tile k+1 have landed; all my LDS reads and writes are done: fragments of the second half in
        // registers (no further wait before its MFMAs), nobody still reads the stage that is overwritten next
        sw_wait_all_lds(fra, frb[1]);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (k + 2 < kt1) dmaW(k + 2, ST);  // (skipped only in the last body: no counted wait follows it)
        loadA(k + 3, vs);
        readB(c0(), nxc(), c0());
        block(0, 1);
        readA(c0(), nxc(), c0());
        block(1, 1);
        readA(c1(), nxc(), c0());
    };
    int kt = kt0, cur = 0;
    for (; kt + 2 < kt1; kt += 2) {
        body(kt, va, c0());
        body(kt + 1, vb, c1());
    }
    if (kt + 1 < kt1) {
        body(kt, va, c0());
        cur = 1;
    }
    {   // last K-tile of the range: nothing left to stage (ONE copy for both stages, so that the accumulators of the
        // two ways into it meet in the same registers; its addresses take the stage at run time)
        const int la = fra_at[1] + cur * A_ST, lw = frb_at[1][0] + cur * W_ST;
        frb[1][0][0] = sw_ds_read16<0>(lw);
        frb[1][0][1] = sw_ds_read16<SW_BN * 64>(lw);
        frb[1][0][2] = sw_ds_read16<2 * SW_BN * 64>(lw);
        frb[1][1][0] = sw_ds_read16<2048>(lw);
        frb[1][1][1] = sw_ds_read16<2048 + SW_BN * 64>(lw);
        frb[1][1][2] = sw_ds_read16<2048 + 2 * SW_BN * 64>(lw);
        sw_wait_lds<9>(fra[0], frb[0]);
        block(0, 0);
        fra[0][0] = sw_ds_read16<0>(la);
        fra[0][1] = sw_ds_read16<SW_BM * 64>(la);
        fra[0][2] = sw_ds_read16<2 * SW_BM * 64>(la);
        sw_wait_lds<9>(fra[1]);
        block(1, 0);
        fra[1][0] = sw_ds_read16<2048>(la);
        fra[1][1] = sw_ds_read16<2048 + SW_BM * 64>(la);
        fra[1][2] = sw_ds_read16<2048 + 2 * SW_BM * 64>(la);
        sw_wait_lds<3>(fra[0], frb[1]);
        block(0, 1);
        sw_wait_lds<0>(fra[1]);
        block(1, 1);
    }
    // Untracked loads whose data is never used (K-tiles past the range) still write their registers when they land:
    // the registers must stay reserved until then, or the allocator hands them to a ds_read / MFMA operand that the
    // late write then clobbers.  Hence: wait for everything, then a (void) use of both sets.
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]) : "memory");
    __syncthreads();

    // ---- epilogue: the whole C tile is staged in LDS once, then every thread handles 16-B pieces of whole rows
    //      (thread -> column quad c4, rows rw + 8 i).  What the pieces need from global memory -- the ReLU gate's
    //      activations, the row list of a scattered store -- is requested for ALL 16 pieces before the staging, into
    //      the registers the correction accumulators give back: one memory round trip per tile instead of one per round.
    const int epi = p.epi;
    const int c4 = tid & 63, col = n0 + c4 * 4, rw = tid >> 6;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] += cor[x][y];
    f32x4 ax[16];
    int crow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + rw + 8 * i;
        const bool ok = row < Mrows;
        crow[i] = row;
        if (cidx) crow[i] = ((gbl_i32_ptr)cidx)[ok ? row : 0];
        if (epi & EPI_MASK) ax[i] = *(gbl_f32x4_ptr)((gbl_f32_ptr)g.aux + (ok ? (size_t)row * p.ldaux + col : 0));
    }
    float bias4[4] = {0.f, 0.f, 0.f, 0.f}, cs4[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi & EPI_BIAS) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bias4[c] = *((gbl_f32_ptr)g.bias + col + c);
    }
    float* cst = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cst[(wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * SW_LDE + wn * 64 + y * 32 + li] = acc[x][y][r];
    __syncthreads();
    // Everything the pieces need from global memory has been requested above; ONE wait the compiler can see, here.
    // Without it every round below starts with `s_waitcnt vmcnt(0)` -- the first use of that round's row number, which
    // hipcc cannot prove landed because the previous rounds' waits sit inside their `row < Mrows` branches -- and
    // that wait also covers the STORE of the round before: sixteen store round trips in a row, 8 of a K = 256 tile's
    // 31 us.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), the other counters at their maxima (gfx9 encoding)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int lr = rw + 8 * i, row = m0 + lr;
        if (row >= Mrows) continue;
        float* cp = g.C + (size_t)crow[i] * p.ldc + col;
        const f32x4 v = *reinterpret_cast<const f32x4*>(cst + lr * SW_LDE + c4 * 4);
        float e[4] = {v.x, v.y, v.z, v.w};
        if (epi & EPI_ATOMIC) {
#pragma unroll
            for (int c = 0; c < 4; ++c) sw_atomic_add(cp + c, e[c]);
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] += bias4[c];
        if (epi & EPI_RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = fmaxf(e[c], 0.f);
        }
        if (epi & EPI_MASK) {
            e[0] = ax[i].x > 0.f ? e[0] : 0.f; e[1] = ax[i].y > 0.f ? e[1] : 0.f;
            e[2] = ax[i].z > 0.f ? e[2] : 0.f; e[3] = ax[i].w > 0.f ? e[3] : 0.f;
        }
#if SW_DBG & 32
        if (e[0] == 1.2345e-30f)  // (no stores)
#endif
        sw_gstore4(cp, e[0], e[1], e[2], e[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) cs4[c] += e[c];
    }
    if (epi & EPI_COLSUM) {
        // the 8 waves hold partial sums of the same 256 columns: meet in LDS, one atomic per column and tile
        __syncthreads();
        *reinterpret_cast<f32x4*>(cst + rw * SW_BN + c4 * 4) = (f32x4){cs4[0], cs4[1], cs4[2], cs4[3]};
        __syncthreads();
        if (tid < SW_BN) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += cst[k * SW_BN + tid];
            sw_atomic_add(g.csum + n0 + tid, t);
        }
    }
}

static bool sw_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Shapes this kernel takes (everything else stays on gemm.hip's kernels)
bool split_nt_fits(const GemmDesc& d) {
    return d.layout == GEMM_NT && d.N >= SW_BN && d.N % SW_BN == 0 && d.K % 4 == 0 && d.K >= 32 && d.lda % 4 == 0 &&
           d.ldc % 4 == 0 && (!(d.epi & EPI_MASK) || d.ldaux % 4 == 0) && !(d.epi & (EPI_SOFTPLUS | EPI_ADDC)) && !d.unaligned &&
           !(d.ragged & RAG_BIDX) && d.M > 64;
}

int launch_gemm_split_nt(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                         hipStream_t stream) {
    if (d.M <= 0 || d.N <= 0 || ngroups <= 0) return SHIVAE_OK;
    SV_REQUIRE(split_nt_fits(d), "gemm_split_nt: shape M=%d N=%d K=%d not supported", d.M, d.N, d.K);
    SV_REQUIRE(groups_dev != nullptr || ngroups == 1, "gemm_split_nt: %d groups need a device group table", ngroups);
    SV_REQUIRE(d.splitk >= 1 && (d.splitk == 1 || d.epi == EPI_ATOMIC), "gemm_split_nt: split-K needs the atomic epilogue");
    SV_REQUIRE(!(d.epi & EPI_ATOMIC) || d.epi == EPI_ATOMIC, "gemm_split_nt: atomic epilogue excludes the others");
    SV_REQUIRE(!d.ragged || groups_dev != nullptr, "gemm_split_nt: ragged launches take a device group table");
    if (!groups_dev)
        SV_REQUIRE(sw_aligned16(g0.A) && sw_aligned16(g0.B) && sw_aligned16(g0.C) && (!(d.epi & EPI_MASK) || sw_aligned16(g0.aux)),
                   "gemm_split_nt: operands must be 16-B aligned");
    SplitNtParams p;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldc = d.ldc; p.ldaux = d.ldaux; p.epi = d.epi; p.ragged = d.ragged;
    p.splitk = d.splitk;
    p.KT = ceil_div(d.K, 32);
    p.Nimg = d.nimg > 0 ? d.nimg : d.N;
    SV_REQUIRE(p.Nimg >= d.N, "gemm_split_nt: image has %d rows, N = %d", p.Nimg, d.N);
    p.kt_per_split = ceil_div(p.KT, d.splitk);
    p.ntx = d.N / SW_BN; p.nty = ceil_div(d.M, SW_BM); p.nz = ngroups * d.splitk;
    p.groups = groups_dev; p.g0 = g0;
    const long long units = (long long)p.nty * p.nz;
    const long long nlin = (units + 7) / 8 * 8 * p.ntx;
    SV_REQUIRE(nlin < (1ll << 31), "gemm_split_nt: grid too large");
    // (Sizing the grid of a launch with device-side row counts by the EXPECTED live row tiles -- at cfg5 seven of
    // eight tiles are dead -- measured 1.3 % of the cfg5 step: a dead workgroup leaves after two scalar loads.  Not
    // worth a loop around this kernel's body, which cost 101 spilled registers when tried.)
    // (Persistent workgroups -- one per CU striding through the tiles, the next tile's first K-tile requested while
    // the last one of the current tile is multiplied -- were built and measured: same results, K = 256 tiles 109
    // against 137 TFLOP/s, long K equal.  With one workgroup per tile the dispatcher starts the next workgroup while
    // the finished one's stores still drain; a persistent workgroup serialises its own epilogue.  Dropped.)
    prof_shape(d.M, d.N, d.K, ngroups);
    prof_begin(PROF_SPLITW_NT, 2.0 * d.M * d.N * (double)d.K * ngroups, stream);
    hipLaunchKernelGGL(gemm_split_nt_kernel, dim3((unsigned)nlin), dim3(SW_NT), 0, stream, p);
    prof_end(stream);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

// =====================================================================================================================
// Split-mode TN GEMM: C[M,N] (+)= A[K,M]^T B[K,N], the decoder's weight gradients (dY^T x activations, K = the rows of
// a chunk; decoders.py:60-66,127-171 under autograd).  Both operands are activations, so both are split in the
// kernel -- but ONCE per 128 x 256 output tile and with the split of the next K-tile running behind the current
// K-tile's MFMAs:
//   tile 128 x 256 x 16 on 8 waves (2 x 4, 64 x 64 per wave, two accumulator sets), one workgroup per CU;
//   two LDS stages of 42 KiB, ONE barrier per K-tile; a thread owns a 4(k) x 4(m) block of one operand: four 16-B
//   global loads, requested two K-tiles ahead into one of two register sets, split in registers, stored as one 8-B
//   piece per row and plane;
//   LDS image (gemm.hip, row-contiguous operands): [plane][k-quad][row & 3][row >> 2 (+8)][4 bf16]: a fragment (8 k of
//   one row) is two ds_read_b64, stores and reads are conflict-free.
// The generic split kernel (gemm.hip: 128 x 128 x 32 on 4 waves, ONE LDS stage, two barriers per K-tile, two
// workgroups per CU) splits every operand element once per 128 output columns and overlaps its split phase only with
// the other workgroup of the CU: 143-156 TFLOP/s on the weight-gradient shapes of cfg3.
// Takes: M % 128 == 0, N % 256 == 0, 16-B aligned rows; EPI_ATOMIC (split-K), EPI_ADDC or plain store; ragged K
// (device-side count) and row lists for both operands (the mask-aware decoder).
constexpr int ST_BM = 128, ST_BN = 256, ST_KT = 16, ST_NT = 512;
// LDS image of one operand: [plane][k-quad][row & 3][row >> 2 (+8 pad)][4 bf16 = 8 B]: a thread's four rows of a k-quad
// go to the four (row & 3) blocks (one ds_write_b64 each, lanes = consecutive row quads = contiguous LDS), a fragment
// (8 k of one row) is two ds_read_b64; the padded block stride puts the four blocks a read touches on disjoint banks.
// (A row-linear image with two ds_write_b128 per plane and block -- half the LDS store instructions -- measured 3-4 %
// slower on the shapes without row lists: 184 / 172 / 176 against 191 / 180 / 183 TFLOP/s, three alternating runs.)
constexpr int ST_QA = ST_BM / 4 + 8, ST_QB = ST_BN / 4 + 8;          // row-quad strides of the (row & 3) blocks
constexpr int ST_PA = 4 * 4 * ST_QA * 8, ST_PB = 4 * 4 * ST_QB * 8;  // bytes per plane (4 k-quads x 4 x Q x 8 B)
constexpr int ST_A_BYTES = 3 * ST_PA, ST_B_BYTES = 3 * ST_PB;
constexpr int ST_STAGE = ST_A_BYTES + ST_B_BYTES;                    // 43 008 B
constexpr int ST_LDE = ST_BN + 4;
constexpr int ST_LDS = ST_BM * ST_LDE * 4 > 2 * ST_STAGE ? ST_BM * ST_LDE * 4 : 2 * ST_STAGE;  // 133 120 B (C staging)

struct SplitTnParams {
    int M, N, K, lda, ldb, ldc, epi, ragged, splitk, kchunk;
    int ntx, nty, nz;
    const GemmGroup* groups;
    GemmGroup g0;
};


// LISTS: 0 no row lists; 1 both operands' K rows come through row lists (layer 0 of the mask-aware decoder, dense
// input-gradient form); 2 only B's (compact form: the gradient operand is compact, the embedding is gathered)
template <int LISTS>
__global__ __launch_bounds__(ST_NT, 2) void gemm_split_tn_kernel(SplitTnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[ST_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, li = lane & 31, lh = lane >> 5;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = slot % p.ntx, u = (slot / p.ntx) * 8 + xcd;
    if (u >= p.nty * p.nz) return;
    const int by = u % p.nty, bz = u / p.nty;
    const int gidx = bz / p.splitk, ks = bz % p.splitk;
    const GemmGroup g = sw_load_group(p.groups, p.g0, gidx);
    const int m0 = by * ST_BM, n0 = bx * ST_BN;
    int Kext = p.K, kchunk = p.kchunk;
    const int *aidx = nullptr, *bidx = nullptr;
    if (p.ragged) {
        if (p.ragged & RAG_CNT) {  // the LIVE K, split evenly over the K-splits (whole K-tiles)
            typedef const int __attribute__((address_space(4))) * const_int_ptr;
            Kext = min(Kext, *(const_int_ptr)(uintptr_t)g.cnt);
            kchunk = ((Kext + p.splitk - 1) / p.splitk + ST_KT - 1) & ~(ST_KT - 1);
        }
        if (p.ragged & RAG_AIDX) aidx = g.aidx;
        if (p.ragged & RAG_BIDX) bidx = g.bidx;
    }
    const int kbeg = ks * kchunk;
    if (kbeg >= Kext) return;  // (such launches accumulate: nothing to add)
    const int kend = min(Kext, kbeg + kchunk);
    const int nkt = (kend - kbeg + ST_KT - 1) / ST_KT;

    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[x][y][r] = 0.f; cor[x][y][r] = 0.f; }

    // ---- loader role of this thread: waves 0-1 the A tile (32 row quads x 4 k-quads), waves 2-5 the B tile (64 x 4),
    //      waves 6-7 none: they run the same code on zeros and park them in the spare LDS behind the two stages (no
    //      branch in the loop: the split arithmetic of the loaders is scheduled between the MFMAs of the same block).
    //      Block = k-quad kq (4 k-rows) x row quad rq (4 consecutive m / n)
    const bool isA = wid < 2, loader = wid < 6;
    const int lt = isA ? tid : tid - 128;
    const int rq = isA ? (lt & 31) : (lt & 63), kq = isA ? (lt >> 5) : ((lt >> 6) & 3);
    const float* X = isA ? g.A : g.B;
    const int ldx = isA ? p.lda : p.ldb;
    // (LISTS == 2: the A loaders fetch B's list entries too and ignore them -- every wave issues the same sequence of
    // vector memory operations, there is no branch round a load)
    constexpr bool LIST = LISTS != 0;
    const int* xidx = (LISTS == 2 || !isA) ? bidx : aidx;
    const bool use_list = LISTS == 1 || (LISTS == 2 && !isA);
    const int col = (isA ? m0 : n0) + 4 * rq;
    // LDS byte offset of this block's piece for row c = 0: ((kq * 4 + c) * Q + rq) * 8; planes ST_PA / ST_PB apart
    const int Q = isA ? ST_QA : ST_QB;
    const int soff = loader ? (isA ? 0 : ST_A_BYTES) + (kq * 4 * Q + rq) * 8 : 2 * ST_STAGE + (tid - 384) * 8;
    const int sstage = loader ? ST_STAGE : 0;                      // (the spare area is not staged)
    const int cstep = loader ? Q * 8 : 1024, pstep = loader ? (isA ? ST_PA : ST_PB) : 4096;
    static_assert(2 * ST_STAGE + 128 * 8 + 3 * 1024 + 2 * 4096 <= ST_LDS, "spare LDS for the idle waves' stores");

    // Row lists: the list entries of the NEXT request are fetched one request ahead and BEFORE this request's data
    // loads (vector memory completes in order: waiting for the entries then leaves the data loads in flight); they stay
    // raw registers until they are used -- a select right behind the load would wait for everything just requested.
    int raw[4] = {0, 0, 0, 0};
    auto fetch_rows = [&](int kt, int (&t)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kbeg + kt * ST_KT + 4 * kq + r;
            t[r] = LIST ? ((gbl_i32_ptr)xidx)[k < kend ? k : 0] : 0;
        }
    };
    // K-tile kt of this thread's block, four 16-B loads, always issued and kept RAW (park() zeroes what is not live:
    // a select right behind a load would wait for it, i.e. for the whole memory latency, in every K-tile).
    // Without row lists: a wave-uniform slab base (scalar arithmetic) + four loop-invariant byte offsets -- no vector
    // arithmetic at all (this kernel is bound by instruction issue).  The slab of the last K-tile is moved back so
    // that it ends with the operand (slot (kq, r) then holds row slab_row + 4 kq + r, live iff it lies in
    // [tile start, kend)); requests past the last K-tile repeat it.  No branch anywhere near a load.
    unsigned voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) voff[r] = loader ? (unsigned)(((4 * kq + r) * ldx + col) * 4) : 0u;
    auto slab_row = [&](int kt) { return min(kbeg + min(kt, nkt - 1) * ST_KT, p.K - ST_KT); };
    auto request = [&](int kt, f32x4 (&v)[4]) {
        if constexpr (!LIST) {
            const unsigned char* slab = reinterpret_cast<const unsigned char*>(X) + (size_t)slab_row(kt) * ldx * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = *(gbl_f32x4_ptr)(gbl_void_ptr)(slab + voff[r]);
        } else {
            int nxt[4];
            fetch_rows(kt + 1, nxt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kbeg + kt * ST_KT + 4 * kq + r;
                int row = raw[r];
                asm volatile("" : "+v"(row));
                row = use_list ? row : k;
                v[r] = *(gbl_f32x4_ptr)((gbl_f32_ptr)X + ((loader && k < kend) ? (size_t)row * ldx + col : 0));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[r] = nxt[r];
        }
    };
    // split the 4 x 4 block of K-tile kt, one 8-B piece per row and plane
    auto park = [&](int stage, const f32x4 (&v)[4], int kt) {
#if SW_DBG & 16
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));  // (diagnostic: loads kept, no split, no LDS write)
        return;
#endif
        unsigned char* d = lds + stage * sstage + soff;
        // slot r of this thread holds row first + r; live iff tile start <= row < kend
        const int t0 = kbeg + kt * ST_KT;
        const int first = (LIST ? t0 : slab_row(kt)) + 4 * kq;
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) live[r] = (unsigned)(first + r - t0) < (unsigned)(kend - t0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // the four k of row c as one vector: v_cvt_pk_bf16_f32 rounds a PAIR and leaves it packed the way the
            // image wants it
            f32x4 t;
            t.x = live[0] ? v[0][c] : 0.f; t.y = live[1] ? v[1][c] : 0.f; t.z = live[2] ? v[2][c] : 0.f; t.w = live[3] ? v[3][c] : 0.f;
            const bf16x4 h = __builtin_convertvector(t, bf16x4);
            const f32x4 e = t - sw_unpack_bf16x4(h);
            const bf16x4 m = __builtin_convertvector(e, bf16x4);
            const bf16x4 l = __builtin_convertvector(e - sw_unpack_bf16x4(m), bf16x4);
            *reinterpret_cast<bf16x4*>(d + c * cstep) = h;
            *reinterpret_cast<bf16x4*>(d + c * cstep + pstep) = m;
            *reinterpret_cast<bf16x4*>(d + c * cstep + 2 * pstep) = l;
        }
    };

    // ---- fragments: row = 32-row block + li, k = 8 lh .. 8 lh + 7 = k-quads 2 lh, 2 lh + 1
    int fa[2], fb[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int ra = wm * 64 + x * 32 + li, rb = wn * 64 + x * 32 + li;
        fa[x] = ((2 * lh * 4 + (ra & 3)) * ST_QA + (ra >> 2)) * 8;
        fb[x] = ST_A_BYTES + ((2 * lh * 4 + (rb & 3)) * ST_QB + (rb >> 2)) * 8;
    }
    auto frag = [&](const unsigned char* q, int kqstep) {
        const uint2 lo = *reinterpret_cast<const uint2*>(q), hi = *reinterpret_cast<const uint2*>(q + kqstep);
        return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
    };
    auto compute = [&](int stage) {
#if SW_DBG & 4
        return;
#endif
        const unsigned char* st = lds + stage * ST_STAGE;
        bf16x8 a[3], b[2][3];  // (one row block of A at a time: 36 fragment registers, not 48)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[y][pl] = frag(st + fb[y] + pl * ST_PB, 4 * ST_QB * 8);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = frag(st + fa[x] + pl * ST_PA, 4 * ST_QA * 8);
#pragma unroll
            for (int y = 0; y < 2; ++y) {  // corrections smallest first; the leading product on its own
                cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[y][0], cor[x][y], 0, 0, 0);
                cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][2], cor[x][y], 0, 0, 0);
                cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][1], cor[x][y], 0, 0, 0);
                cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][0], cor[x][y], 0, 0, 0);
                cor[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][1], cor[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][0], acc[x][y], 0, 0, 0);
            }
        }
    };

    // ---- K loop: K-tile t lives in register set t & 1 from two iterations before it is parked, in LDS stage t & 1
    //      from the iteration before it is multiplied
    f32x4 v0[4], v1[4];
    fetch_rows(0, raw);
    request(0, v0);
    request(1, v1);
    park(0, v0, 0);
    request(2, v0);
    __syncthreads();
    // (Measured and dropped: the two waves of a SIMD -- w and w + 4 -- taking "multiply" and "split + park" in
    // opposite order, so that one issues MFMAs while the other does vector work: 164-171 against 177-184 TFLOP/s.  The
    // scheduler's fine-grained mix of the split's VALU instructions between the MFMAs of the same wave is better.)
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
        compute(0);
        park(1, v1, kt + 1);
        request(kt + 3, v1);
        __syncthreads();
        compute(1);
        park(0, v0, kt + 2);
        request(kt + 4, v0);
        __syncthreads();
    }
    int cur = 0;
    if (kt + 1 < nkt) {
        compute(0);
        park(1, v1, kt + 1);
        __syncthreads();
        cur = 1;
    }
    compute(cur);  // (one copy for both stages: the accumulators of the two ways into it meet in the same registers)
    __syncthreads();

    // ---- epilogue: the C tile staged in LDS, then 16-B pieces of whole rows (thread -> column quad c4, rows rw + 8 i)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] += cor[x][y];
    float* cst = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cst[(wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ST_LDE + wn * 64 + y * 32 + li] = acc[x][y][r];
    __syncthreads();
    const int epi = p.epi;
    const int c4 = tid & 63, rw = tid >> 6;
    float* cbase = g.C + (size_t)m0 * p.ldc + n0 + c4 * 4;
    if (epi & EPI_ATOMIC) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = rw + 8 * i;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cst + lr * ST_LDE + c4 * 4);
            float* cp = cbase + (size_t)lr * p.ldc;
            sw_atomic_add(cp, v.x); sw_atomic_add(cp + 1, v.y); sw_atomic_add(cp + 2, v.z); sw_atomic_add(cp + 3, v.w);
        }
        return;
    }
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {
        f32x4 old[4];
        if (epi & EPI_ADDC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) old[i] = *(gbl_f32x4_ptr)(gbl_f32_ptr)(cbase + (size_t)(rw + 8 * (i0 + i)) * p.ldc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = rw + 8 * (i0 + i);
            f32x4 v = *reinterpret_cast<const f32x4*>(cst + lr * ST_LDE + c4 * 4);
            if (epi & EPI_ADDC) v += old[i];
            sw_gstore4(cbase + (size_t)lr * p.ldc, v.x, v.y, v.z, v.w);
        }
    }
}

bool split_tn_fits(const GemmDesc& d) {
    return d.layout == GEMM_TN && d.M % ST_BM == 0 && d.N % ST_BN == 0 && d.K >= ST_KT && d.lda % 4 == 0 && d.ldb % 4 == 0 &&
           d.ldc % 4 == 0 && (d.epi == EPI_ATOMIC || d.epi == EPI_ADDC || d.epi == 0) && !d.unaligned &&
           !(d.ragged & RAG_CIDX) && (!(d.ragged & RAG_AIDX) || (d.ragged & RAG_BIDX));
}

int launch_gemm_split_tn(const GemmDesc& d, const GemmGroup* groups_dev, int ngroups, const GemmGroup& g0,
                         hipStream_t stream) {
    if (d.M <= 0 || d.N <= 0 || ngroups <= 0) return SHIVAE_OK;
    SV_REQUIRE(split_tn_fits(d), "gemm_split_tn: shape M=%d N=%d K=%d not supported", d.M, d.N, d.K);
    SV_REQUIRE(groups_dev != nullptr || ngroups == 1, "gemm_split_tn: %d groups need a device group table", ngroups);
    SV_REQUIRE(d.splitk >= 1 && (d.splitk == 1 || d.epi == EPI_ATOMIC), "gemm_split_tn: split-K needs the atomic epilogue");
    SV_REQUIRE(!d.ragged || groups_dev != nullptr, "gemm_split_tn: ragged launches take a device group table");
    SV_REQUIRE(!d.ragged || d.epi == EPI_ATOMIC || d.epi == EPI_ADDC, "gemm_split_tn: a ragged launch accumulates into C");
    if (!groups_dev)
        SV_REQUIRE(sw_aligned16(g0.A) && sw_aligned16(g0.B) && sw_aligned16(g0.C), "gemm_split_tn: operands must be 16-B aligned");
    else
        SV_REQUIRE(d.c_aligned, "gemm_split_tn: group tables must be 16-B aligned");
    SplitTnParams p;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.epi = d.epi; p.ragged = d.ragged;
    p.splitk = d.splitk;
    if (d.epi == EPI_ATOMIC) {
        // the caller sized the K-splits for 128 x 128 tiles; these tiles are twice as large, so a launch of few groups
        // would fill half the chip (atomic accumulation: any number of splits gives the same sum up to its order)
        const long long tiles = (long long)(d.M / ST_BM) * (d.N / ST_BN) * ngroups;
        while (tiles * p.splitk < 512 && d.K / (2 * p.splitk) >= 512) p.splitk *= 2;
    }
    p.kchunk = round_up(ceil_div(d.K, p.splitk), ST_KT);
    p.ntx = d.N / ST_BN; p.nty = d.M / ST_BM; p.nz = ngroups * p.splitk;
    p.groups = groups_dev; p.g0 = g0;
    const long long units = (long long)p.nty * p.nz;
    const long long nlin = (units + 7) / 8 * 8 * p.ntx;
    SV_REQUIRE(nlin < (1ll << 31), "gemm_split_tn: grid too large");
    prof_shape(d.M, d.N, d.K, ngroups);
    prof_begin(PROF_SPLIT_TN, 2.0 * d.M * d.N * (double)d.K * ngroups, stream);
    if (d.ragged & RAG_AIDX) hipLaunchKernelGGL(gemm_split_tn_kernel<1>, dim3((unsigned)nlin), dim3(ST_NT), 0, stream, p);
    else if (d.ragged & RAG_BIDX) hipLaunchKernelGGL(gemm_split_tn_kernel<2>, dim3((unsigned)nlin), dim3(ST_NT), 0, stream, p);
    else hipLaunchKernelGGL(gemm_split_tn_kernel<0>, dim3((unsigned)nlin), dim3(ST_NT), 0, stream, p);
    prof_end(stream);
    SV_CHECK_HIP(hipGetLastError());
    return SHIVAE_OK;
}

}  // namespace shivae
