"""GPU parity of the individual HIP kernels (called through the C ABI) against numpy / the oracle."""
import numpy as np
import pytest
import torch

import _golden
from oracle import shivae_oracle as so
from shivae_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)


def _gemm(layout, A, B, C0, bias=None, aux=None, epi=0, splitk=1, M=None, N_=None, K=None, lda=None, ldb=None):
    At, Bt, Ct = _t(A), _t(B), _t(C0)
    bt = None if bias is None else _t(bias)
    xt = None if aux is None else _t(aux)
    N.check(N.lib().shivae_gemm(layout, M, N_, K, N.ptr(At), lda or A.shape[1], N.ptr(Bt), ldb or B.shape[1], N.ptr(Ct),
                                C0.shape[1], N.ptr(bt), N.ptr(xt), 0 if aux is None else aux.shape[1], epi, splitk,
                                N.stream()))
    torch.cuda.synchronize()
    return Ct.cpu().numpy()


SHAPES = [(1, 1, 1), (5, 3, 7), (33, 65, 31), (128, 128, 32), (130, 20, 23), (257, 129, 100), (64, 300, 36),
          (1000, 17, 260), (40, 513, 64), (300, 2, 256), (2, 300, 256),
          # 16-B aligned 128x128-tile cases: LDS-DMA staging (TN / NN) with a K range that ends inside a K-tile,
          # a K shorter than one tile, whole tiles only
          (132, 260, 100), (128, 132, 4), (200, 136, 333), (256, 256, 96)]


@pytest.mark.parametrize("M,Nn,K", SHAPES)
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_layouts(layout, M, Nn, K):
    rng = np.random.default_rng(M * 1000 + Nn * 10 + K + layout)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((Nn, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    if layout == 0:
        out = _gemm(0, A, B, np.zeros((M, Nn), np.float32), M=M, N_=Nn, K=K)
    elif layout == 1:
        out = _gemm(1, A, np.ascontiguousarray(B.T), np.zeros((M, Nn), np.float32), M=M, N_=Nn, K=K)
    else:
        out = _gemm(2, np.ascontiguousarray(A.T), np.ascontiguousarray(B.T), np.zeros((M, Nn), np.float32), M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-5 * np.sqrt(K))


def test_gemm_epilogues_and_splitk():
    rng = np.random.default_rng(0)
    M, Nn, K = 200, 70, 90
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((Nn, K)).astype(np.float32)
    bias = rng.standard_normal(Nn).astype(np.float32)
    C0 = rng.standard_normal((M, Nn)).astype(np.float32)
    aux = rng.standard_normal((M, Nn)).astype(np.float32)
    base = A.astype(np.float64) @ B.astype(np.float64).T
    out = _gemm(0, A, B, C0, bias=bias, epi=N.EPI_BIAS | N.EPI_RELU, M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, np.maximum(base + bias, 0), rtol=1e-5, atol=1e-4)
    out = _gemm(0, A, B, C0, bias=bias, epi=N.EPI_BIAS | N.EPI_SOFTPLUS, M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, so.softplus(base + bias), rtol=1e-5, atol=1e-4)
    out = _gemm(0, A, B, C0, bias=bias, aux=aux, epi=N.EPI_BIAS | N.EPI_ADDC | N.EPI_RELU | N.EPI_MASK, M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, np.maximum(base + bias + C0, 0) * (aux > 0), rtol=1e-5, atol=1e-4)
    out = _gemm(2, np.ascontiguousarray(A.T), np.ascontiguousarray(B.T), C0, epi=N.EPI_ATOMIC, splitk=3, M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, base + C0, rtol=1e-5, atol=1e-4)
    # strided views: leading dims larger than the extents
    Ap = np.zeros((M, K + 5), np.float32); Ap[:, :K] = A
    Bp = np.zeros((Nn, K + 3), np.float32); Bp[:, :K] = B
    out = _gemm(0, Ap, Bp, np.zeros((M, Nn), np.float32), M=M, N_=Nn, K=K)
    np.testing.assert_allclose(out, base, rtol=1e-5, atol=1e-4)


def _plan_for(cfg, T, B):
    tl = cfg["types_list"]
    return N.Plan(T, B, cfg["H"], cfg["Z"], cfg["K"], [N.TYPE_CODE[t["type"]] for t in tl],
                  [int(t["dim"]) if t["type"] == "cat" else 1 for t in tl], cfg["learn_std"])


def _raw_params(cfg, dec_params, learn_std):
    cols = []
    for t, p in zip(cfg["types_list"], dec_params):
        if t["type"] in ("real", "pos"):
            cols.append(p[0][..., None])
            if learn_std:
                sd = p[1].astype(np.float64)
                cols.append(np.where(sd > 20, sd, np.log(np.expm1(np.minimum(sd, 20))))[..., None])
        elif t["type"] == "bin":
            cols.append(p[..., None])
        else:
            cols.append(p[..., 1:])
    return np.concatenate(cols, -1).astype(np.float32)


@pytest.mark.parametrize("name", _golden.CASES)
def test_heter_nll_kernel_matches_reference(name):
    g, cfg, sd = _golden.load(name)
    T, B = cfg["T"], cfg["B"]
    plan = _plan_for(cfg, T, B)
    gp = _golden.golden_dec_params(g, cfg["types_list"])
    P = _raw_params(cfg, gp, cfg["learn_std"])
    assert P.shape[-1] == plan.p_dim
    x, mask = _t(g["x"]), torch.as_tensor(g["mask"], device=DEV)
    Pt = _t(P)
    dP = torch.zeros_like(Pt)
    rl = torch.zeros(T * B, 6, device=DEV)
    gs = 1.0 / B
    N.check(N.lib().shivae_heter_nll(plan.h, N.ptr(x), N.ptr(mask.view(torch.uint8)), N.ptr(Pt), plan.p_dim, N.ptr(dP),
                                     T * B, gs, N.ptr(rl), N.stream()))
    torch.cuda.synchronize()
    rl = rl.cpu().numpy().reshape(T, B, 6)
    for i, ty in enumerate(("real", "pos", "bin", "cat")):
        np.testing.assert_allclose(rl[..., i].sum(0), g["nll_b/nll_" + ty], rtol=2e-5, atol=2e-4)
    # per-timestep terms against the reference's per-attribute nll
    per_t = {ty: np.zeros((T, B)) for ty in ("real", "pos", "bin", "cat")}
    for i, t in enumerate(cfg["types_list"]):
        per_t[t["type"]] += g["nll_attr/%d" % i]
    for i, ty in enumerate(("real", "pos", "bin", "cat")):
        np.testing.assert_allclose(rl[..., i], per_t[ty], rtol=2e-5, atol=2e-4)
    # gradient wrt the raw head outputs against the oracle's formulas
    tidx = so.transform_idx(cfg["types_list"])
    dPn = dP.cpu().numpy().reshape(T, B, -1)
    off = 0
    for i, t in enumerate(cfg["types_list"]):
        lo, dim, ty = tidx[i], int(t["dim"]), t["type"]
        obs = (~g["mask"][:, :, lo + dim - 1]) * gs
        par = gp[i] if ty not in ("real", "pos") else (gp[i][0].astype(np.float64), gp[i][1].astype(np.float64))
        gr = so.attr_nll_grad(ty, g["x"][:, :, lo:lo + dim].astype(np.float64), par if ty in ("real", "pos") else par.astype(np.float64))
        if ty in ("real", "pos"):
            np.testing.assert_allclose(dPn[..., off], gr[0] * obs, rtol=1e-4, atol=1e-5); off += 1
            if cfg["learn_std"]:
                np.testing.assert_allclose(dPn[..., off], gr[1] * obs * so.softplus_grad_from_out(par[1]), rtol=1e-4, atol=1e-5); off += 1
        elif ty == "bin":
            np.testing.assert_allclose(dPn[..., off], gr * obs, rtol=1e-4, atol=1e-6); off += 1
        else:
            np.testing.assert_allclose(dPn[..., off:off + dim - 1], (gr * obs[..., None])[..., 1:], rtol=1e-4, atol=1e-6); off += dim - 1


@pytest.mark.parametrize("name", _golden.CASES)
def test_kl_kernel_matches_reference(name):
    g, cfg, sd = _golden.load(name)
    T, B, Z, K = cfg["T"], cfg["B"], cfg["Z"], cfg["K"]
    mu, sdq, mup, sdp, pr = (_t(g[k]) for k in ("z_mean", "z_std", "zp_mean", "zp_std", "s_probs"))
    outs = [torch.zeros_like(mu) for _ in range(4)] + [torch.zeros_like(pr)]
    rl = torch.zeros(T * B, 6, device=DEV)
    kz = 0.37
    N.check(N.lib().shivae_kl_terms(N.ptr(mu), N.ptr(sdq), N.ptr(mup), N.ptr(sdp), Z, N.ptr(pr), K, T * B, Z, K, kz,
                                    *[N.ptr(o) for o in outs], N.ptr(rl), N.stream()))
    torch.cuda.synchronize()
    rl = rl.cpu().numpy().reshape(T, B, 6)
    np.testing.assert_allclose(rl[..., 4].sum(0), g["klz_b"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(rl[..., 5].sum(0), g["kls_b"], rtol=2e-5, atol=1e-4)
    a = [g[k].astype(np.float64) for k in ("z_mean", "z_std", "zp_mean", "zp_std")]
    dmq, dsq, dmp, dsp = so.kl_q_z_grad(*a)
    ref = [dmq * kz, dsq * kz, dmp * kz, dsp * kz * so.softplus_grad_from_out(a[3])]
    for o, r in zip(outs[:4], ref):
        np.testing.assert_allclose(o.cpu().numpy(), r, rtol=2e-4, atol=1e-5)
    pb = g["s_probs"].astype(np.float64)
    dpi = so.kl_q_s_grad(pb) * kz
    dlg = pb * (dpi - (pb * dpi).sum(-1, keepdims=True))
    np.testing.assert_allclose(outs[4].cpu().numpy(), dlg, rtol=2e-4, atol=1e-6)


def test_reparam_kernels():
    rng = np.random.default_rng(3)
    R, K, Z, temp = 333, 5, 7, 0.7
    lg = rng.standard_normal((R, K)).astype(np.float32)
    u = rng.random((R, K)).astype(np.float32)
    mean, sd = rng.standard_normal((R, Z)).astype(np.float32), (rng.random((R, Z)) + 0.1).astype(np.float32)
    eps = rng.standard_normal((R, Z)).astype(np.float32)
    t = [_t(a) for a in (lg, u, mean, sd, eps)]
    s, pr, z = torch.zeros(R, K, device=DEV), torch.zeros(R, K, device=DEV), torch.zeros(R, Z, device=DEV)
    N.check(N.lib().shivae_reparam_fwd(N.ptr(t[0]), N.ptr(t[1]), K, K, temp, N.ptr(s), N.ptr(pr), N.ptr(t[2]), N.ptr(t[3]),
                                       N.ptr(t[4]), Z, Z, N.ptr(z), R, N.stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pr.cpu().numpy(), so.softmax(lg.astype(np.float64)), rtol=1e-5, atol=1e-7)
    sref = so.gumbel_softmax(lg.astype(np.float64), u, np.float64(temp))
    np.testing.assert_allclose(s.cpu().numpy(), sref, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(z.cpu().numpy(), so.reparam(mean, sd, eps), rtol=1e-6, atol=1e-6)
    ds, dz = rng.standard_normal((R, K)).astype(np.float32), rng.standard_normal((R, Z)).astype(np.float32)
    dl, dm, dsd = torch.zeros(R, K, device=DEV), torch.zeros(R, Z, device=DEV), torch.zeros(R, Z, device=DEV)
    st, dst, dzt = _t(sref), _t(ds), _t(dz)
    N.check(N.lib().shivae_reparam_bwd(N.ptr(st), N.ptr(dst), K, K, temp, N.ptr(dl), N.ptr(t[4]), N.ptr(dzt), Z, Z, N.ptr(dm),
                                       N.ptr(dsd), R, N.stream()))
    torch.cuda.synchronize()
    ref = sref * (ds - (sref * ds).sum(-1, keepdims=True)) / temp
    np.testing.assert_allclose(dl.cpu().numpy(), ref, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dsd.cpu().numpy(), dz * eps, rtol=1e-6)
    np.testing.assert_allclose(dm.cpu().numpy(), dz)


def test_clip_adam_kernel():
    rng = np.random.default_rng(5)
    n = 100003
    p0, g0 = rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 3).astype(np.float32)
    P = {"w": p0.astype(np.float64).copy()}
    G = {"w": g0.astype(np.float64)}
    m, v = {"w": np.zeros(n)}, {"w": np.zeros(n)}
    pt, gt = _t(p0), _t(g0)
    mt, vt = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    scratch, sq = torch.zeros(1024, device=DEV), torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        tot = so.clip_adam_step(P, G, m, v, step, lr=5e-3, clip=0.5)
        N.check(N.lib().shivae_sqnorm(N.ptr(gt), n, N.ptr(scratch), N.ptr(sq), N.stream()))
        N.check(N.lib().shivae_clip_adam(N.ptr(pt), N.ptr(gt), N.ptr(mt), N.ptr(vt), n, N.ptr(sq), 0.5, 5e-3, 0.9, 0.999,
                                         1e-8, step, N.stream()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(sq.sqrt()), tot, rtol=1e-5)
        np.testing.assert_allclose(pt.cpu().numpy(), P["w"], rtol=0, atol=2e-6)
