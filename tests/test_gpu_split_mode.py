"""Split mode of the 128x128-tile GEMMs (shivae_set_gemm_split: every fp32 operand as three bf16 terms, six bf16 MFMA
products per fp32 product, fp32 accumulation) against the fp32-input MFMA mode on adversarial operands, both measured
against fp64 through the C ABI (shivae_gemm).  The claim under test: on the SAME inputs the split mode's error is no
larger than the fp32-MFMA mode's, so it is an fp32-level way of computing the same products, not a reduced precision.

Layers it carries: decoders.py:60-66,127-171 (every trunk / head Linear of the reference) and their backward passes.

Cases: K = 131 072 (the K of the layer-0 input gradient at cfg3: D*2H), magnitudes spread over 2^+-40 inside one
row, catastrophic cancellation, operands in the fp32 denormal range, and +-inf / NaN propagation."""
import numpy as np
import pytest
import torch

from shivae_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda"
LAYOUTS = {"NT": 0, "NN": 1, "TN": 2, "NT_presplit": 3}  # 3: the 128x256-tile kernel against a pre-split weight image


def _gemm(layout, A, B, split):
    """C[M,N] = A[M,K] B[N,K]^T in the given storage layout (A, B given K-contiguous) and arithmetic mode."""
    M, K = A.shape
    Nn = B.shape[0]
    L = N.lib()
    if layout == 3:
        if not split:
            return _gemm(0, A, B, 0)
        a, b = A.contiguous(), B.contiguous()
        img = torch.empty(L.shivae_presplit_bytes(Nn, K), dtype=torch.uint8, device=DEV)
        C = torch.full((M, Nn), float("nan"), device=DEV)
        N.check(L.shivae_presplit_weights(N.ptr(b), Nn, K, K, N.ptr(img), N.stream()))
        N.check(L.shivae_gemm_nt_presplit(M, Nn, K, N.ptr(a), K, N.ptr(img), N.ptr(C), Nn, None, None, 0, 0, 1, N.stream()))
        torch.cuda.synchronize()
        return C
    if layout == 0:
        a, b = A.contiguous(), B.contiguous()
    elif layout == 1:
        a, b = A.contiguous(), B.t().contiguous()
    else:
        a, b = A.t().contiguous(), B.t().contiguous()
    C = torch.full((M, Nn), float("nan"), device=DEV)
    L.shivae_set_gemm_split(split)
    try:
        N.check(L.shivae_gemm(layout, M, Nn, K, N.ptr(a), a.shape[1], N.ptr(b), b.shape[1], N.ptr(C), Nn, None, None, 0,
                              0, 1, N.stream()))
        torch.cuda.synchronize()
    finally:
        L.shivae_set_gemm_split(0)
    return C


def _errors(A, B, layout):
    """Per mode: (rms, max) over the outputs of |C - C64| / sum_k |a||b| (the normwise scale of one dot product)."""
    ref = A.double() @ B.double().t()
    nat = A.double().abs() @ B.double().abs().t()
    nat = torch.where(nat > 0, nat, torch.ones_like(nat))
    out = []
    for split in (0, 1):
        C = _gemm(layout, A, B, split)
        assert bool(torch.isfinite(C).all())
        e = (C.double() - ref).abs() / nat
        out.append((float(e.pow(2).mean().sqrt()), float(e.max())))
    return out


def _spread(rng, shape, lo=-40, hi=40):
    v = rng.standard_normal(shape).astype(np.float32)
    e = rng.integers(lo, hi + 1, size=shape)
    return np.ldexp(v, e).astype(np.float32)


def _case(name, rng):
    M, Nn = 128, 256
    if name == "long_k":           # K of the layer-0 input gradient (D*2H at cfg3), dense random operands
        K = 131072
        A, B = rng.standard_normal((M, K)), rng.standard_normal((Nn, K))
    elif name == "long_k_positive":  # no cancellation at all: every rounding and every dropped term has one sign
        K = 131072
        A, B = np.abs(rng.standard_normal((M, K))) + 0.5, np.abs(rng.standard_normal((Nn, K))) + 0.5
    elif name == "spread":         # 2^-40 .. 2^40 inside every row of both operands
        K = 4096
        A, B = _spread(rng, (M, K)), _spread(rng, (Nn, K))
    elif name == "spread_one_side":  # activations of ordinary size against weights spread over 80 binades
        K = 528
        A, B = rng.standard_normal((M, K)), _spread(rng, (Nn, K))
    elif name == "cancellation":   # sum_k a_k b_k = sum_k r_k d_k with |d| = 2^-12 |s|: 12 bits cancel
        K = 8192
        r, s = rng.standard_normal((M, K // 2)), rng.standard_normal((Nn, K // 2))
        A = np.concatenate([r, r], 1)
        B = np.concatenate([s, -s * (1 - 2.0 ** -12)], 1)
    elif name == "denormal_products":  # tiny x huge: operands down to 2^-100, partners up to 2^100
        K = 2048
        A = _spread(rng, (M, K), -100, -60)
        B = _spread(rng, (Nn, K), 60, 100)
    else:
        raise KeyError(name)
    return torch.from_numpy(np.asarray(A, dtype=np.float32)).to(DEV), torch.from_numpy(np.asarray(B, dtype=np.float32)).to(DEV)


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("name", ["long_k", "long_k_positive", "spread", "spread_one_side", "cancellation",
                                  "denormal_products"])
def test_split_mode_error_not_above_fp32_mfma(name, layout):
    rng = np.random.default_rng(len(name) * 131 + LAYOUTS[layout])
    A, B = _case(name, rng)
    (r_f32, m_f32), (r_split, m_split) = _errors(A, B, LAYOUTS[layout])
    print("%s %s: err / sum|a||b|  fp32-MFMA rms %.3e max %.3e | split rms %.3e max %.3e" % (name, layout, r_f32, m_f32, r_split, m_split))
    # no worse than the fp32-input MFMA on the same operands: the rms over the outputs outright, the single
    # worst of the 32 768 outputs (an extreme-value statistic of two different rounding sequences) within a quarter
    assert r_split <= r_f32, (r_f32, r_split)
    assert m_split <= 1.25 * m_f32, (m_f32, m_split)


def test_split_mode_fp32_denormal_operands():
    """Operands that ARE fp32 denormals (|x| < 2^-126) are outside the split mode's contract (include/shivae_hip.h,
    shivae_set_gemm_split): the bf16 matrix cores flush denormal inputs, so such operands count as zero or as their
    leading term only.  What must still hold: finite results that never exceed the true scale of the dot product."""
    rng = np.random.default_rng(5)
    M = Nn = 128
    K = 512
    A = _spread(rng, (M, K), -140, -128)    # denormal fp32 values
    B = _spread(rng, (Nn, K), 100, 120)
    A, B = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    ref = A.double() @ B.double().t()
    nat = A.double().abs() @ B.double().abs().t()
    C0, C1 = _gemm(0, A, B, 0), _gemm(0, A, B, 1)
    assert bool(torch.isfinite(C0).all()) and bool(torch.isfinite(C1).all())
    e0, e1 = float(((C0.double() - ref).abs() / nat).max()), float(((C1.double() - ref).abs() / nat).max())
    print("fp32 denormal operands: fp32-MFMA %.3e split %.3e" % (e0, e1))
    assert e0 < 1e-5 and e1 <= 1.0, (e0, e1)


@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_split_mode_nonfinite_propagation(layout):
    """+-inf and NaN operands (include/shivae_hip.h, shivae_set_gemm_split): every output whose dot product touches a
    non-finite operand is non-finite in BOTH modes and every other output is unaffected; the split mode turns an
    infinite operand into NaN (x - hi(x) = inf - inf) where the fp32-input MFMA gives +-inf or NaN."""
    rng = np.random.default_rng(17)
    M, Nn = 128, 256
    K = 256
    A, B = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((Nn, K)).astype(np.float32)
    A[3, 7] = np.inf
    A[64, 100] = -np.inf
    A[90, 255] = np.nan
    B[5, 0] = np.nan
    B[77, 31] = np.inf
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    bad = np.zeros((M, Nn), dtype=bool)
    bad[[3, 64, 90], :] = True
    bad[:, [5, 77]] = True
    Af, Bf = np.where(np.isfinite(A), A, 0).astype(np.float64), np.where(np.isfinite(B), B, 0).astype(np.float64)
    ref = Af @ Bf.T
    for split in (0, 1):
        C = _gemm(LAYOUTS[layout], At, Bt, split).cpu().numpy()
        assert not np.isfinite(C[bad]).any(), split
        assert np.isfinite(C[~bad]).all(), split
        np.testing.assert_allclose(C[~bad], ref[~bad], rtol=0, atol=2e-4)
        if split:
            assert np.isnan(C[bad]).all()   # the documented difference: inf operands give NaN, not inf
