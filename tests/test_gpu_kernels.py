"""GPU parity of the individual HIP kernels (called through the C ABI) against numpy / the oracle."""
import numpy as np
import pytest
import torch

import _golden
from oracle import shivae_oracle as so
from shivae_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["f32_mfma", "split_bf16"], autouse=True)
def gemm_mode(request):
    """Every test of this module runs in both arithmetic modes of the 128x128-tile GEMMs (shivae_set_gemm_split): the
    fp32-input MFMA and the exact 3-way bf16 split with six bf16 MFMA products -- same tolerances."""
    N.lib().shivae_set_gemm_split(1 if request.param == "split_bf16" else 0)
    yield request.param
    N.lib().shivae_set_gemm_split(0)


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
          (132, 260, 100), (128, 132, 4), (200, 136, 333), (256, 256, 96),
          # M % 128 == 0, N % 256 == 0: in split mode the TN layout takes gemm_split_tn_kernel (128 x 256 x 16 tiles) --
          # one K-tile exactly, one row more, K ranges that end inside a K-tile (the last slab is moved back), an odd
          # number of K-tiles, several row / column tiles
          (128, 256, 16), (128, 256, 17), (128, 256, 33), (256, 512, 1000), (128, 512, 47), (384, 256, 2049)]


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
    # the split-mode TN kernel's epilogues (M % 128 == 0, N % 256 == 0): split-K atomics, read-modify-write, plain store
    M2, N2, K2 = 256, 256, 531
    A2 = rng.standard_normal((K2, M2)).astype(np.float32)
    B2 = rng.standard_normal((K2, N2)).astype(np.float32)
    C2 = rng.standard_normal((M2, N2)).astype(np.float32)
    base2 = A2.astype(np.float64).T @ B2.astype(np.float64)
    for epi, sk, want in ((N.EPI_ATOMIC, 3, base2 + C2), (N.EPI_ATOMIC, 1, base2 + C2), (N.EPI_ADDC, 1, base2 + C2), (0, 1, base2)):
        out = _gemm(2, A2, B2, C2, epi=epi, splitk=sk, M=M2, N_=N2, K=K2)
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=2e-4, err_msg="epi %d splitk %d" % (epi, sk))
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


# ---------------------------------------------------------------- the phases of the step (SURVEY.md 8b minimum set)
def _bound_model(tl, T, B, H, Z, K, P, learn_std=True):
    import shivae_amd
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=learn_std)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.zero_grad()
    return m, m._activate(T, B)


def _out(plan, which, T, B):
    o = torch.empty(T, B, plan.output_width(N.O[which]), device=DEV)
    N.check(N.lib().shivae_get_output(plan.h, N.O[which], N.ptr(o), N.stream()))
    torch.cuda.synchronize()
    return o.cpu().numpy()


SCAN_CASES = {
    # name: (n_real, n_pos, n_bin, cats, T, B, H, Z, K)
    "h16": (1, 1, 1, [3], 9, 21, 16, 4, 3),
    "h80_multiwave": (2, 1, 1, [5], 6, 18, 80, 24, 6),
    "k20_two_ktiles": (2, 1, 1, [3], 5, 17, 32, 6, 20),
    # scan_fwd_kernel<2,512> / scan_bwd_kernel<2,512>: the instantiation of the cfg3 benchmark (8 waves, two hidden
    # tiles per wave), T = 64 steady-state steps of its weight / record pipelines, a full and a ragged sequence tile
    "h256_T64": (1, 1, 1, [4], 64, 20, 256, 64, 16),
}


@pytest.mark.parametrize("name", list(SCAN_CASES))
def test_scan_fwd_and_bwd_records_match_oracle(name):
    """shivae_scan_fwd / shivae_scan_bwd on their own: every record of the forward chain (shivae.py:102-143) and of
    the reverse-time chain against the oracle, the backward driven by the oracle's own incoming gradients."""
    import shivae_amd
    nr, npos, nb, cats, T, B, H, Z, K = SCAN_CASES[name]
    tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
    rng = np.random.default_rng(len(name) * 7 + H)
    P = so.init_params(tl, H, Z, K, True, seed=3)
    for k in P:
        if "bias" in k:
            P[k] = (rng.standard_normal(P[k].shape) * 0.1).astype(np.float32)
    X = so.x_dim(tl)
    x = rng.standard_normal((T, B, X)).astype(np.float32)
    mask = rng.random((T, B, X)) < 0.3
    x = np.where(mask, 0, x).astype(np.float32)
    u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
    temp, beta = 1.3, 0.7
    om = so.OracleShiVAE(P, tl, H, Z, K, True)
    c = om.forward(x, mask, u, eps, beta, temp)
    G = om.backward()
    m, plan = _bound_model(tl, T, B, H, Z, K, P)
    xt, ut, et = _t(x), _t(u), _t(eps)
    L = N.lib()
    N.check(L.shivae_scan_fwd(plan.h, N.ptr(xt), N.ptr(ut), N.ptr(et), temp, N.stream()))
    fwd = {"Z": c["z"], "Z_MEAN": c["mu"], "Z_STD": c["std"], "S": c["s"], "S_PROBS": c["probs"], "PHI_Z": c["phi_z"],
           "H_PREV": c["h_prev"], "A_S": c["a_s"], "LOGITS": c["logits"], "E1": c["e1"], "E2": c["e2"],
           "C_NEW": c["c_new"], "GATES": np.concatenate([c["gi"], c["gf"], c["gg"], c["go"]], -1)}
    for k, ref in fwd.items():
        np.testing.assert_allclose(_out(plan, k, T, B), ref, rtol=1e-4, atol=2e-5, err_msg=k)
    # the gradients that reach the chain from the decoder, the prior and the KL terms, as the oracle has them
    sb = c["scan_bwd"]
    d_emb = np.concatenate([sb["dphi_z_ext"], sb["ds_ext"], sb["dh_ext"]], -1)
    pb, dpr = c["probs"], sb["dprobs"]
    dlg_ext = pb * (dpr - (pb * dpr).sum(-1, keepdims=True))
    ext = [_t(a) for a in (d_emb, sb["dmu_ext"], sb["dstd_ext"], dlg_ext)]
    N.check(L.shivae_scan_bwd(plan.h, N.ptr(xt), N.ptr(ut), N.ptr(et), temp, *[N.ptr(a) for a in ext], N.stream()))
    bwd = {"D_GATES": sb["d_gates"], "D_PHIZ_PRE": sb["d_phiz_pre"], "D_MU": sb["d_mu"], "D_STD_PRE": sb["d_std_pre"],
           "D_E2_PRE": sb["d_e2_pre"], "D_E1_PRE": sb["d_e1_pre"], "D_LOGITS_PRE": sb["d_lg_pre"], "D_AS_PRE": sb["d_as_pre"]}
    for k, ref in bwd.items():
        got = _out(plan, k, T, B)
        assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), (k, np.abs(got - ref).max(), np.abs(ref).max())
    # ... and the parameter gradients this phase owns (the recurrent cell, the encoder, phi_x)
    m._publish_grads(None)
    import _golden
    bad = _golden.bad_grads([(k, p.grad.cpu().numpy()) for k, p in m.named_parameters()
                             if k.startswith(("encoder.", "phi_z.", "rnn."))], G)
    assert not bad, bad[:8]


def test_phases_compose_to_the_step():
    """shivae_elbo_step == scan_fwd; prior_kl; decoder_nll; scan_bwd; loss_reduce (same kernels, same order)."""
    import shivae_amd
    tl = shivae_amd.utils.make_types_list(2, 2, 1, [3, 4])
    T, B, H, Z, K = 7, 26, 32, 8, 5
    rng = np.random.default_rng(21)
    P = so.init_params(tl, H, Z, K, True, seed=4)
    X = so.x_dim(tl)
    x = rng.standard_normal((T, B, X)).astype(np.float32)
    mask = rng.random((T, B, X)) < 0.4
    mask[..., 5:8] = mask[..., 5:6]; mask[..., 8:12] = mask[..., 8:9]
    x = np.where(mask, 0, x).astype(np.float32)
    u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
    xt, mt, ut, et = _t(x), torch.as_tensor(mask, device=DEV).view(torch.uint8), _t(u), _t(eps)
    beta, temp, ib = 0.6, 2.0, 1.0 / B
    L = N.lib()
    res = []
    for composed in (False, True):
        m, plan = _bound_model(tl, T, B, H, Z, K, P)
        losses = torch.zeros(16, device=DEV)
        s = N.stream()
        if not composed:
            N.check(L.shivae_elbo_step(plan.h, N.ptr(xt), N.ptr(mt), N.ptr(ut), N.ptr(et), beta, temp, ib, 0, N.ptr(losses), s))
        else:
            N.check(L.shivae_scan_fwd(plan.h, N.ptr(xt), N.ptr(ut), N.ptr(et), temp, s))
            N.check(L.shivae_prior_kl(plan.h, beta, ib, 0, s))
            N.check(L.shivae_decoder_nll(plan.h, N.ptr(xt), N.ptr(mt), ib, 0, s))
            N.check(L.shivae_scan_bwd(plan.h, N.ptr(xt), N.ptr(ut), N.ptr(et), temp, None, None, None, None, s))
            N.check(L.shivae_loss_reduce(plan.h, beta, ib, N.ptr(losses), s))
        torch.cuda.synchronize()
        res.append((losses[:9].cpu().numpy(), m.flat_grads.cpu().numpy().copy()))
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-6)
    assert np.abs(res[1][1] - res[0][1]).max() <= 1e-5 * np.abs(res[0][1]).max()  # (atomic summation order)
    assert np.abs(res[0][1]).max() > 0
