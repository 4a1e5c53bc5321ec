"""GPU parity at the attribute mixes BASELINE.json names (cfg3: D=256, 64 real/64 pos/64 bin/64 cat with C=4,
H=256, Z=64, K=16; cfg5: D=512, 52/51/51 + 358 cat with C=8, 90 % missing) against the fp64 oracle, at a T*B the
oracle finishes in seconds.  What these pin that the small-D cases cannot: 256- and 512-group ragged GEMM tables,
lanes of the likelihood kernel that own several attributes, its multi-group staging loop (cfg5: 8 groups of 64),
its non-LDS fallback kernel, two micro-batches of several row chunks each.

Reference: decoders.py:109-176 (per-attribute networks), base_missvae.py:233-285 (masked likelihood)."""
import zlib

import numpy as np
import pytest
import torch

import _golden
import shivae_amd
from oracle import shivae_oracle as so
from shivae_amd import _native as N
from test_gpu_step import _synth, FWD_RTOL, FWD_ATOL, LOSS_RTOL, GRAD_RTOL

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["f32_mfma", "split_bf16"], autouse=True)
def gemm_mode(request):
    """Every test of this module runs in both arithmetic modes of the 128x128-tile GEMMs (shivae_set_gemm_split): the
    fp32-input MFMA and the exact 3-way bf16 split with six bf16 MFMA products -- same tolerances."""
    N.lib().shivae_set_gemm_split(1 if request.param == "split_bf16" else 0)
    yield request.param
    N.lib().shivae_set_gemm_split(0)

MIXES = {
    # name: (types_list, T, B, H, Z, K, miss, chunk_rows, micro)
    "cfg3_mix": (shivae_amd.utils.make_types_list(64, 64, 64, [4] * 64), 3, 24, 256, 64, 16, 0.3, 20, 16),
    "cfg5_mix": (shivae_amd.utils.make_types_list(52, 51, 51, [8] * 358), 4, 20, 48, 16, 16, 0.9, 32, 0),
}


from _golden import dekink_decoder as _dekink_decoder  # noqa: E402


@pytest.mark.parametrize("name", list(MIXES))
def test_step_at_baseline_attribute_mix(name):
    tl, T, B, H, Z, K, miss, chunk, micro = MIXES[name]
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    P = so.init_params(tl, H, Z, K, True, seed=2)
    for k in P:
        if "bias" in k:
            P[k] = (rng.standard_normal(P[k].shape) * 0.1).astype(np.float32)
    x, mask = _synth(rng, tl, T, B, miss)
    u = rng.random((T, B, K)).astype(np.float32)
    eps = rng.standard_normal((T, B, Z)).astype(np.float32)
    beta, temp = 0.9, 2.0
    om = so.OracleShiVAE(P, tl, H, Z, K, True)
    c = om.forward(x, mask, u, eps, beta, temp)
    if _dekink_decoder(P, tl, c["emb"]):       # (the decoder's input does not depend on the decoder's biases)
        om = so.OracleShiVAE(P, tl, H, Z, K, True)
        c = om.forward(x, mask, u, eps, beta, temp)
    G = om.backward()
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    m.chunk_rows = chunk                       # >= 2 row chunks per micro-batch
    if micro:
        m.micro_batch = lambda T_, B_: micro   # 2 micro-batches (16 + 8 sequences)
        assert len(m.micro_batches(T, B)) == 2
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.train()
    m.set_noise(u, eps)
    loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=beta, temp=temp)
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(float(loss[k]), c["losses"][k], rtol=LOSS_RTOL, atol=1e-5, err_msg=k)
    loss["n_elbo"].backward()
    bad = _golden.bad_grads([(k, p.grad.cpu().numpy()) for k, p in m.named_parameters()], G)
    assert not bad, bad[:8]
    # forward tensors of the dense (imputation) form of the same decoder
    m.eval()
    m.set_noise(u, eps)
    z_tup, zp_tup, s_tup, likes = m.feed2model(torch.from_numpy(x), temp)
    np.testing.assert_allclose(z_tup[0].cpu().numpy(), c["z"], rtol=FWD_RTOL, atol=FWD_ATOL)
    dp = m._last["dec_params"].cpu().numpy()
    off = 0
    for i, t in enumerate(tl):
        ref = c["dec_params"][i]
        if t["type"] in ("real", "pos"):
            np.testing.assert_allclose(dp[..., off], ref[0], rtol=FWD_RTOL, atol=5 * FWD_ATOL)
            np.testing.assert_allclose(so.softplus(dp[..., off + 1].astype(np.float64)), ref[1], rtol=FWD_RTOL, atol=5 * FWD_ATOL)
            off += 2
        elif t["type"] == "bin":
            np.testing.assert_allclose(dp[..., off], ref, rtol=FWD_RTOL, atol=5 * FWD_ATOL)
            off += 1
        else:
            C = int(t["dim"])
            np.testing.assert_allclose(dp[..., off:off + C - 1], ref[..., 1:], rtol=FWD_RTOL, atol=5 * FWD_ATOL)
            off += C - 1


def _random_dec_params(rng, tl, T, B):
    """Random likelihood parameters: oracle form (list per attribute) and the raw [T,B,P] head outputs."""
    par, cols = [], []
    for t in tl:
        ty = t["type"]
        if ty in ("real", "pos"):
            mu, raw = rng.standard_normal((T, B)), rng.standard_normal((T, B)) * 1.5
            par.append((mu, so.softplus(raw)))
            cols += [mu[..., None], raw[..., None]]
        elif ty == "bin":
            lg = rng.standard_normal((T, B)) * 2
            par.append(lg)
            cols.append(lg[..., None])
        else:
            C = int(t["dim"])
            th = np.concatenate([np.zeros((T, B, 1)), rng.standard_normal((T, B, C - 1)) * 2], -1)
            par.append(th)
            cols.append(th[..., 1:])
    return par, np.concatenate(cols, -1).astype(np.float32)


NLL_SHAPES = {
    # name: (types_list, rows T x B, miss, expected kernel form)
    "cfg3_P576": (shivae_amd.utils.make_types_list(64, 64, 64, [4] * 64), (5, 77), 0.3, "one group of 256"),
    "cfg5_P2763": (shivae_amd.utils.make_types_list(52, 51, 51, [8] * 358), (3, 41), 0.9, "8 groups of 64"),
    "wide_cats_fallback": (shivae_amd.utils.make_types_list(2, 1, 1, [60] * 64), (2, 19), 0.5, "> 64 KiB of LDS: heter_nll_kernel"),
}


@pytest.mark.parametrize("name", list(NLL_SHAPES))
def test_heter_nll_standalone_at_baseline_widths(name):
    tl, (T, B), miss, _ = NLL_SHAPES[name]
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x, mask = _synth(rng, tl, T, B, miss)
    par, raw = _random_dec_params(rng, tl, T, B)
    # the oracle sees the parameters the kernel sees (fp32 head outputs)
    par32, off = [], 0
    for t in tl:
        ty = t["type"]
        if ty in ("real", "pos"):
            par32.append((raw[..., off].astype(np.float64), so.softplus(raw[..., off + 1].astype(np.float64)))); off += 2
        elif ty == "bin":
            par32.append(raw[..., off].astype(np.float64)); off += 1
        else:
            C = int(t["dim"])
            par32.append(np.concatenate([np.zeros((T, B, 1)), raw[..., off:off + C - 1].astype(np.float64)], -1)); off += C - 1
    ref, per_attr = so.heter_nll(x.astype(np.float64), mask, par32, tl)
    plan = N.Plan(T, B, 16, 4, 4, [N.TYPE_CODE[t["type"]] for t in tl],
                  [int(t["dim"]) if t["type"] == "cat" else 1 for t in tl], True)
    assert plan.p_dim == raw.shape[-1]
    xt = torch.as_tensor(x, device=DEV)
    mt = torch.as_tensor(mask, device=DEV)
    Pt = torch.as_tensor(raw, device=DEV)
    dP = torch.full_like(Pt, 7.0)
    rl = torch.zeros(T * B, 6, device=DEV)
    gs = 0.25
    N.check(N.lib().shivae_heter_nll(plan.h, N.ptr(xt), N.ptr(mt.view(torch.uint8)), N.ptr(Pt), plan.p_dim, N.ptr(dP),
                                     T * B, gs, N.ptr(rl), N.stream()))
    torch.cuda.synchronize()
    rl = rl.cpu().numpy().reshape(T, B, 6)
    per_t = {ty: np.zeros((T, B)) for ty in ("real", "pos", "bin", "cat")}
    for i, t in enumerate(tl):
        per_t[t["type"]] += per_attr[i]
    for i, ty in enumerate(("real", "pos", "bin", "cat")):
        np.testing.assert_allclose(rl[..., i], per_t[ty], rtol=2e-5, atol=2e-4 * max(1, len(tl) // 64), err_msg=ty)
        np.testing.assert_allclose(rl[..., i].sum(0), ref[ty], rtol=2e-5, atol=1e-3)
    # gradient wrt the raw head outputs (zero where the entry is missing)
    tidx = so.transform_idx(tl)
    dPn = dP.cpu().numpy()
    off = 0
    for i, t in enumerate(tl):
        lo, dim, ty = tidx[i], int(t["dim"]), t["type"]
        obs = (~mask[:, :, lo + dim - 1]) * gs
        gr = so.attr_nll_grad(ty, x[:, :, lo:lo + dim].astype(np.float64), par32[i])
        if ty in ("real", "pos"):
            np.testing.assert_allclose(dPn[..., off], gr[0] * obs, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(dPn[..., off + 1], gr[1] * obs * so.softplus_grad_from_out(par32[i][1]), rtol=1e-4, atol=1e-5)
            off += 2
        elif ty == "bin":
            np.testing.assert_allclose(dPn[..., off], gr * obs, rtol=1e-4, atol=1e-6); off += 1
        else:
            np.testing.assert_allclose(dPn[..., off:off + dim - 1], (gr * obs[..., None])[..., 1:], rtol=1e-4, atol=1e-6)
            off += dim - 1
    assert off == raw.shape[-1]
