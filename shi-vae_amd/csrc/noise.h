// Counter-based noise for the recurrent scans: the Gumbel uniforms u[t][b][k] and the Gaussian eps[t][b][n] of the
// reparameterisation (samplers.py:23-30 torch.rand, samplers.py:57-62 normal_()) as pure functions of
// (seed, row = t*B + b, index), so that the forward scan, the reverse scan and a filling kernel produce the SAME
// numbers without the 2.6 GB per cfg3 step of materialised noise going through HBM twice.
//
// Generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
// published constants and known-answer vectors are checked in tests/test_gpu_noise.py).
//   key     = (seed low 32, seed high 32)
//   counter = (row low 32, row high 32, group g, stream)   stream 0: eps, 1: u; group g = index / 4
// One call gives the four values index 4g .. 4g+3 of a row.
//   uniform : ((x >> 9) + 0.5) * 2^-23   in (0, 1), every step exact in fp32
//   normal  : Box-Muller on two uniforms -> two normals (sqrtf(-2 logf(u0)) * {cos, sin}(2 pi u1))
// Only single multiplications: nothing for the compiler to contract differently in different kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace shivae {

struct SvU4 { unsigned v[4]; };

__host__ __device__ inline SvU4 sv_philox4x32_10(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return SvU4{{c0, c1, c2, c3}};
}

__host__ __device__ inline float sv_u01(unsigned x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

struct SvF4 { float v[4]; };

// u[row][4g .. 4g+3]
__device__ __forceinline__ SvF4 sv_noise_u4(unsigned long long seed, unsigned long long row, int g) {
    const SvU4 x = sv_philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)row, (unsigned)(row >> 32), (unsigned)g, 1u);
    return SvF4{{sv_u01(x.v[0]), sv_u01(x.v[1]), sv_u01(x.v[2]), sv_u01(x.v[3])}};
}

// eps[row][4g .. 4g+3]
__device__ __forceinline__ SvF4 sv_noise_eps4(unsigned long long seed, unsigned long long row, int g) {
    const SvU4 x = sv_philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)row, (unsigned)(row >> 32), (unsigned)g, 0u);
    const float r0 = sqrtf(-2.f * logf(sv_u01(x.v[0]))), r1 = sqrtf(-2.f * logf(sv_u01(x.v[2])));
    float s0, c0, s1, c1;
    sincospif(2.f * sv_u01(x.v[1]), &s0, &c0);
    sincospif(2.f * sv_u01(x.v[3]), &s1, &c1);
    return SvF4{{r0 * c0, r0 * s0, r1 * c1, r1 * s1}};
}

}  // namespace shivae
