"""GPU parity of the full fused step (ShiVAE host API -> C ABI -> HIP kernels) against the golden
vectors of the reference and against the oracle on larger seeded inputs."""
import zlib
import numpy as np
import pytest
import torch

import _golden
import shivae_amd
from oracle import shivae_oracle as so
from shivae_amd import _native as N

pytestmark = pytest.mark.gpu

# stated fp32 tolerances (north_star "within a stated fp32 tolerance")
FWD_RTOL, FWD_ATOL = 1e-4, 2e-5
LOSS_RTOL = 2e-5
GRAD_RTOL = _golden.GRAD_RTOL  # relative to the largest |grad| entry of each tensor; next to it, every step test holds
#                               ||got - ref||_2 / ||ref||_2 < _golden.GRAD_L2_RTOL per tensor (_golden.bad_grads)


def _grads(m):
    return [(k, p.grad.cpu().numpy()) for k, p in m.named_parameters()]


@pytest.fixture(params=["f32_mfma", "split_bf16"], autouse=True)
def gemm_mode(request):
    """Every test of this module runs in both arithmetic modes of the 128x128-tile GEMMs (shivae_set_gemm_split): the
    fp32-input MFMA and the exact 3-way bf16 split with six bf16 MFMA products -- same tolerances."""
    N.lib().shivae_set_gemm_split(1 if request.param == "split_bf16" else 0)
    yield request.param
    N.lib().shivae_set_gemm_split(0)


def _model(cfg, sd):
    m = shivae_amd.ShiVAE(cfg["H"], cfg["Z"], cfg["K"], 1, types_list=cfg["types_list"], learn_std=cfg["learn_std"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


@pytest.mark.parametrize("name", _golden.CASES)
def test_forward_outputs_match_reference(name):
    g, cfg, sd = _golden.load(name)
    m = _model(cfg, sd)
    m.set_noise(g["u"], g["eps"])
    z_tup, zp_tup, s_tup, likes = m.feed2model(torch.from_numpy(g["x"]), cfg["temp"])
    for t, k in zip(z_tup + zp_tup + s_tup, ("z", "z_mean", "z_std", "zp_mean", "zp_std", "s", "s_probs")):
        np.testing.assert_allclose(t.cpu().numpy(), g[k], rtol=FWD_RTOL, atol=FWD_ATOL, err_msg=k)
    gp = _golden.golden_dec_params(g, cfg["types_list"])
    for i, t in enumerate(cfg["types_list"]):
        lk = likes[t["name"]]
        if t["type"] in ("real", "pos"):
            np.testing.assert_allclose(lk.loc[..., 0].cpu().numpy(), gp[i][0], rtol=FWD_RTOL, atol=FWD_ATOL)
            np.testing.assert_allclose(lk.scale[..., 0].cpu().numpy(), gp[i][1], rtol=FWD_RTOL, atol=FWD_ATOL)
        elif t["type"] == "bin":
            np.testing.assert_allclose(lk.logits[..., 0].cpu().numpy(), gp[i], rtol=FWD_RTOL, atol=FWD_ATOL)
        else:
            lg = lk.logits.cpu().numpy()
            np.testing.assert_allclose(lg - lg[..., :1], gp[i], rtol=FWD_RTOL, atol=FWD_ATOL)


@pytest.mark.parametrize("name", _golden.CASES)
def test_losses_and_gradients_match_reference(name):
    g, cfg, sd = _golden.load(name)
    m = _model(cfg, sd)
    m.train()
    m.set_noise(g["u"], g["eps"])
    loss = m(torch.from_numpy(g["x"]), torch.from_numpy(g["mask"]), beta=cfg["beta"], temp=cfg["temp"])
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(float(loss[k]), g["loss/" + k], rtol=LOSS_RTOL, atol=1e-5, err_msg=k)
    loss["n_elbo"].backward()
    assert all(p.grad is not None for p in m.parameters())
    bad = _golden.bad_grads(_grads(m), {k: g["grad/" + k] for k, _ in m.named_parameters()})
    assert not bad, bad[:8]
    # per-sequence sums over t of the six terms (base_missvae.py:284,318,343)
    ps = m._last["persum"].cpu().numpy()
    assert ps.shape == (cfg["B"], 6)
    for i, ty in enumerate(("real", "pos", "bin", "cat")):
        np.testing.assert_allclose(ps[:, i], g["nll_b/nll_" + ty], rtol=1e-4, atol=2e-4, err_msg=ty)
    np.testing.assert_allclose(ps[:, 4], g["klz_b"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ps[:, 5], g["kls_b"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", _golden.CASES)
def test_captured_step_matches_reference(name):
    """Small models replay their step from a HIP graph from the third call with the same shapes and scalars on
    (ShiVAE._step_graphed): first call eager, second captures and replays, third replays.  Every one of them must give
    the reference's losses and gradients."""
    g, cfg, sd = _golden.load(name)
    m = _model(cfg, sd)
    m.train()
    m.use_graphs = True
    x, mask = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    for call in range(3):
        m.zero_grad()
        m.set_noise(g["u"], g["eps"])
        loss = m(x, mask, beta=cfg["beta"], temp=cfg["temp"])
        for k in shivae_amd.LOSS_KEYS:
            np.testing.assert_allclose(float(loss[k]), g["loss/" + k], rtol=LOSS_RTOL, atol=1e-5, err_msg="%s call %d" % (k, call))
        loss["n_elbo"].backward()
        bad = _golden.bad_grads(_grads(m), {k: g["grad/" + k] for k, _ in m.named_parameters()})
        assert not bad, (call, bad[:8])
        ps = m._last["persum"].cpu().numpy()
        np.testing.assert_allclose(ps[:, 4], g["klz_b"], rtol=1e-4, atol=1e-4)
    assert len(m._graphs) == 1, "the step was not captured"
    # other scalars -> another graph; the first one stays valid
    m.zero_grad(); m.set_noise(g["u"], g["eps"])
    l2 = m(x, mask, beta=0.5 * cfg["beta"] + 0.1, temp=cfg["temp"])
    m.zero_grad(); m.set_noise(g["u"], g["eps"])
    l3 = m(x, mask, beta=cfg["beta"], temp=cfg["temp"])
    np.testing.assert_allclose(float(l3["n_elbo"]), g["loss/n_elbo"], rtol=LOSS_RTOL, atol=1e-5)
    assert abs(float(l2["n_elbo"]) - float(l3["n_elbo"])) > 0


@pytest.mark.parametrize("name", _golden.CASES)
def test_fused_clip_adam_on_golden_gradients(name):
    """trainers.py:222-224 clip_grad_norm_(0.5) + Adam(lr=5e-3).step() on the reference's own gradients: the HIP
    optimiser alone (gradient buffer filled from the golden vectors), then compared with the reference's
    parameters after that step and its total gradient norm."""
    g, cfg, sd = _golden.load(name)
    m = _model(cfg, sd)
    opt = shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5)
    opt.zero_grad()
    for k, p in m.named_parameters():
        p.grad.copy_(torch.from_numpy(g["grad/" + k]))  # views of the flat gradient buffer
    opt.step()
    np.testing.assert_allclose(float(opt.last_grad_sqnorm.sqrt()), g["grad_norm"], rtol=1e-5)
    for k, p in m.named_parameters():
        # compare the UPDATE (parameters move by about lr = 5e-3 per element)
        upd, ref = p.detach().cpu().numpy() - sd[k], g["adam1/" + k] - sd[k]
        assert np.abs(upd - ref).max() <= 2e-6, (k, np.abs(upd - ref).max())


@pytest.mark.parametrize("name", ["cfg1_T100_B8", "tiny_all"])
def test_per_sequence_terms_and_reconstruction(name):
    g, cfg, sd = _golden.load(name)
    m = _model(cfg, sd)
    m.eval()
    x, mask = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    m.set_noise(g["u"], g["eps"])
    losses, outs = m._step(x, mask, cfg["beta"], cfg["temp"], grad=False, collect=[N.O["ROWLOSS"]])
    ps = m._last["persum"].cpu().numpy()
    for i, ty in enumerate(("real", "pos", "bin", "cat")):
        np.testing.assert_allclose(ps[:, i], g["nll_b/nll_" + ty], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(ps[:, 4], g["klz_b"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ps[:, 5], g["kls_b"], rtol=1e-4, atol=1e-4)
    # imputation parameters at temp 0.01 (base_missvae.py:345-359)
    m.set_noise(g["u"], g["eps"])
    (likes, x_hat), _ = m.reconstruction(x, temp=0.01)
    gp = _golden.golden_dec_params(g, cfg["types_list"], prefix="rec")
    for i, t in enumerate(cfg["types_list"]):
        lk = likes[t["name"]]
        if t["type"] in ("real", "pos"):
            np.testing.assert_allclose(lk.loc[..., 0].cpu().numpy(), gp[i][0], rtol=1e-3, atol=1e-4)
        elif t["type"] == "bin":
            np.testing.assert_allclose(lk.logits[..., 0].cpu().numpy(), gp[i], rtol=1e-3, atol=1e-4)
        else:
            lg = lk.logits.cpu().numpy()
            np.testing.assert_allclose(lg - lg[..., :1], gp[i], rtol=1e-3, atol=1e-4)
    assert tuple(x_hat.shape) == tuple(g["x"].shape)


def _synth(rng, tl, T, B, miss):
    cols, mcols = [], []
    for t in tl:
        dim = int(t["dim"])
        mk = rng.random((T, B)) < miss
        if t["type"] in ("real", "pos"):
            v = rng.standard_normal((T, B, 1))
        elif t["type"] == "bin":
            v = (rng.random((T, B, 1)) < 0.5).astype(np.float64)
        else:
            v = np.eye(dim)[rng.integers(0, dim, size=(T, B))]
        cols.append(v); mcols.append(np.repeat(mk[..., None], dim, -1))
    mask = np.concatenate(mcols, -1)
    return np.where(mask, 0, np.concatenate(cols, -1)).astype(np.float32), mask


CASES = {
    # (types, T, B, H, Z, K, learn_std, miss, chunk_rows): shapes the golden set does not cover
    "aligned_h32": (shivae_amd.utils.make_types_list(3, 2, 2, [4, 6], interleave=True), 9, 37, 32, 16, 16, True, 0.3, 0),
    "ragged_chunks": (shivae_amd.utils.make_types_list(2, 2, 1, [3]), 11, 21, 20, 5, 4, True, 0.5, 64),
    "h80_multiwave": (shivae_amd.utils.make_types_list(1, 1, 1, [5]), 6, 18, 80, 24, 6, False, 0.3, 0),
    "cat_heavy_miss90": (shivae_amd.utils.make_types_list(1, 1, 1, [8] * 7), 8, 19, 24, 8, 8, True, 0.9, 0),
    # edge cases: K > 16 (two k tiles: unfused logits/softmax stage), a single timestep, a single attribute type,
    # every entry missing, 16 waves per scan workgroup (H = 256) with the 128x128 GEMM tiles and fused bias sums
    "k20_two_ktiles": (shivae_amd.utils.make_types_list(2, 1, 1, [3]), 5, 17, 32, 6, 20, True, 0.3, 0),
    "single_timestep": (shivae_amd.utils.make_types_list(1, 1, 1, [4]), 1, 33, 16, 3, 3, True, 0.3, 0),
    "only_real": (shivae_amd.utils.make_types_list(5, 0, 0, []), 7, 9, 16, 4, 3, False, 0.2, 0),
    "only_cat": (shivae_amd.utils.make_types_list(0, 0, 0, [2, 3, 9]), 7, 9, 16, 4, 3, True, 0.2, 0),
    "all_missing": (shivae_amd.utils.make_types_list(1, 1, 1, [3]), 6, 8, 16, 4, 3, True, 1.0, 0),
    "h256_16waves": (shivae_amd.utils.make_types_list(1, 1, 1, [3]), 4, 40, 256, 64, 16, True, 0.3, 0),
    # the scan instantiation the cfg3 benchmark runs (H=256, Z=64, K=16: 8 waves, two hidden tiles per wave) at a
    # sequence length where its software pipelines are in steady state: one full and one ragged 16-sequence tile
    "h256_T64": (shivae_amd.utils.make_types_list(1, 1, 1, [4]), 64, 20, 256, 64, 16, True, 0.3, 0),
}


@pytest.mark.parametrize("name", list(CASES))
def test_step_matches_oracle_on_seeded_inputs(name):
    _check_case(name)


@pytest.mark.parametrize("gather", ["1", "0"])
@pytest.mark.parametrize("name", ["ragged_chunks", "cat_heavy_miss90", "single_timestep", "aligned_h32", "all_missing"])
def test_compact_layer0_input_gradient_matches_oracle(name, gather, monkeypatch):
    """SHIVAE_DEMB=compact forces the per-attribute form of the decoder's layer-0 input gradient at any observed
    fraction; SHIVAE_L0D_GATHER picks between its two sums over attributes -- product rows stored and gathered per
    chunk row (1, default) or added atomically (0).  E % 4 != 0 (single_timestep: E = 35) takes the gather's scalar form."""
    monkeypatch.setenv("SHIVAE_DEMB", "compact")      # both read at plan creation
    monkeypatch.setenv("SHIVAE_L0D_GATHER", gather)
    _check_case(name)


@pytest.mark.parametrize("name,chunk,act,overflow", [
    ("cat_heavy_miss90", 64, 32, False),   # ~6 observed rows per attribute and chunk of 64: fits 32 activation rows
    ("cat_heavy_miss90", 152, 48, False),  # one chunk = the whole batch
    ("aligned_h32", 128, 32, True),        # ~90 observed rows per chunk of 128: every chunk is re-run in pieces of 32
    ("ragged_chunks", 100, 48, True),      # ~50 of 100 observed: some chunks fit, some do not
])
def test_narrow_activation_buffers_match_oracle(name, chunk, act, overflow):
    """shivae_plan_set_activation_rows: activation buffers with fewer rows than the row chunk (what the mask-aware
    decoder needs when most entries are missing).  Same oracle, same tolerances; a chunk whose counts exceed the
    buffers is detected on the host and run in pieces."""
    m = _check_case(name, chunk=chunk, act=act)
    tl, T, B = CASES[name][:3]
    plan = m._plans[(T, B)]
    assert plan.chunk_rows == min(chunk, T * B) and plan.activation_rows == act
    assert (plan.overflow_chunks > 0) == overflow, plan.overflow_chunks
    # dense passes on the same plan (forward only: every row of every attribute) step through the rows `act` at a time
    x, mask = _synth(np.random.default_rng(5), tl, T, B, CASES[name][7])
    u, eps = np.random.default_rng(6).random((T, B, CASES[name][5])).astype(np.float32), \
        np.random.default_rng(7).standard_normal((T, B, CASES[name][4])).astype(np.float32)
    ref = shivae_amd.ShiVAE(*CASES[name][3:6], 1, types_list=tl, learn_std=CASES[name][6])
    ref.load_state_dict(m.state_dict())
    out = []
    for mod in (m, ref):
        mod.eval()
        mod.set_noise(u, eps)
        with torch.no_grad():
            out.append(mod(torch.from_numpy(x), torch.from_numpy(mask), beta=1.0, temp=1.0))
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(float(out[0][k]), float(out[1][k]), rtol=1e-5, atol=1e-5, err_msg=k)


def test_chunk_grows_into_what_narrow_activations_free(monkeypatch):
    """The host side of the same: with a workspace budget that holds two default chunks, a 90 %-missing training batch
    gets activation buffers sized by its observed fraction and a chunk several times larger; losses and gradients
    equal those of the plain layout (fp32 summation order aside)."""
    monkeypatch.setenv("SHIVAE_CHUNK_BUDGET", "1")  # the library's default chunk: its minimum of 256 rows
    tl = shivae_amd.utils.make_types_list(2, 2, 2, [4, 6], interleave=True)
    T, B, H, Z, K = 40, 64, 64, 8, 4
    rng = np.random.default_rng(17)
    x, mask = _synth(rng, tl, T, B, 0.9)
    u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
    res, plans = {}, {}
    for narrow in (False, True):
        torch.manual_seed(3)
        m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
        m.narrow_activations = narrow
        p0 = m._make_plan(T, B)
        assert p0.chunk_rows == 256
        ws0 = p0.workspace_bytes
        ws1 = m._make_plan(T, B, chunk_rows=512).workspace_bytes
        m._plans.clear()
        m.workspace_limit_bytes = ws0 + int((ws1 - ws0) * 1.2)  # room for ~2 default chunks
        m.train()
        m.set_noise(u, eps)
        loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.7, temp=2.0)
        loss["n_elbo"].backward()
        plans[narrow] = m._plans[(T, B)]
        res[narrow] = ({k: float(loss[k].detach()) for k in shivae_amd.LOSS_KEYS},
                       {k: p.grad.cpu().numpy().copy() for k, p in m.named_parameters()})
    assert plans[False].activation_rows == plans[False].chunk_rows < T * B
    assert plans[True].chunk_rows >= 2 * plans[False].chunk_rows, (plans[True].chunk_rows, plans[False].chunk_rows)
    assert plans[True].activation_rows < plans[True].chunk_rows and plans[True].overflow_chunks == 0
    assert plans[True].workspace_bytes <= ws0 + int((ws1 - ws0) * 1.2)
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(res[True][0][k], res[False][0][k], rtol=1e-6, atol=1e-6, err_msg=k)
    for k, gd in res[False][1].items():
        err = np.abs(res[True][1][k] - gd).max() / max(np.abs(gd).max(), 1e-3)
        assert err < 5e-5, (k, err)


def _check_case(name, chunk=None, act=0):
    tl, T, B, H, Z, K, learn_std, miss, chunk0 = CASES[name]
    chunk = chunk0 if chunk is None else chunk
    rng = np.random.default_rng(zlib.crc32(name.encode()))  # (str hashes change per process)
    P = so.init_params(tl, H, Z, K, learn_std, seed=1)
    for k in P:
        if "bias" in k:
            P[k] = (rng.standard_normal(P[k].shape) * 0.1).astype(np.float32)
    x, mask = _synth(rng, tl, T, B, miss)
    u = rng.random((T, B, K)).astype(np.float32)
    eps = rng.standard_normal((T, B, Z)).astype(np.float32)
    beta, temp = 0.8, 1.5
    om = so.OracleShiVAE(P, tl, H, Z, K, learn_std)
    c = om.forward(x, mask, u, eps, beta, temp)
    # (H = 256 over 1 280 rows is ~5e6 decoder ReLU inputs: move the units that sit within fp32 rounding of a kink,
    # where the fp32 GPU and the fp64 oracle gate differently -- _golden.dekink_decoder)
    if H >= 256 and T * B > 1000 and _golden.dekink_decoder(P, tl, c["emb"]):
        om = so.OracleShiVAE(P, tl, H, Z, K, learn_std)
        c = om.forward(x, mask, u, eps, beta, temp)
    G = om.backward()
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=learn_std)
    m.chunk_rows = chunk
    m.act_rows = act
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.train()
    m.set_noise(u, eps)
    loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=beta, temp=temp)
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(float(loss[k]), c["losses"][k], rtol=LOSS_RTOL, atol=1e-5, err_msg=k)
    loss["n_elbo"].backward()
    bad = _golden.bad_grads(_grads(m), G)
    assert not bad, bad[:8]
    return m


def test_mask_aware_decoder_matches_dense(monkeypatch):
    """Training steps run every attribute's decoder only on its observed rows (SHIVAE_COMPACT, default on);
    SHIVAE_COMPACT=0 runs all rows and masks afterwards, as the reference does (base_missvae.py:271-285).
    Same losses and gradients, up to fp32 summation order; 128x128 GEMM tiles, three ragged row chunks, one
    attribute never observed, one always."""
    tl = shivae_amd.utils.make_types_list(3, 2, 2, [4, 6], interleave=True)
    T, B, H, Z, K = 24, 50, 128, 16, 8
    rng = np.random.default_rng(99)
    x, mask = _synth(rng, tl, T, B, 0.4)
    off = 0
    for i, t in enumerate(tl):
        dim = int(t["dim"])
        if i == 0: mask[..., off:off + dim] = mask[0, 0, off]                    # one value everywhere
        if i == 1: mask[..., off:off + dim] = ~mask[0, 0, 0] if mask.dtype == bool else 1 - mask[0, 0, 0]
        off += dim
    x = np.where(mask, 0, x).astype(np.float32)
    u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SHIVAE_COMPACT", mode)
        torch.manual_seed(3)
        m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
        m.chunk_rows = 512
        m.train()
        m.set_noise(u, eps)
        loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.7, temp=2.0)
        loss["n_elbo"].backward()
        res[mode] = ({k: float(loss[k].detach()) for k in shivae_amd.LOSS_KEYS},
                     {k: p.grad.cpu().numpy().copy() for k, p in m.named_parameters()})
    for k in shivae_amd.LOSS_KEYS:
        np.testing.assert_allclose(res["1"][0][k], res["0"][0][k], rtol=1e-6, atol=1e-6, err_msg=k)
    for k, gd in res["0"][1].items():
        err = np.abs(res["1"][1][k] - gd).max() / max(np.abs(gd).max(), 1e-3)
        assert err < 5e-5, (k, err)


def test_captured_training_run_matches_eager_stepwise():
    """A training run of a small model through the captured (HIP-graph) step, every replay re-computed eagerly on the
    same parameters and noise tensors.  Regression test: with the d_emb fill captured as a memset node the 77th replay
    of exactly this run handed the reverse scan garbage (gradients of 1e14, losses unaffected) and the run went to
    NaN two steps later; captured steps now consist of kernel nodes only (plan.hip zero_async)."""
    from shivae_amd.synth import synth_batch
    tl = shivae_amd.utils.make_types_list(35, 0, 0, [])
    T, B, H, Z, K = 48, 64, 10, 35, 10
    x, mask = synth_batch(tl, T, B, miss=0.3, seed=1234, device="cuda", chunk=B, seq0=0)
    torch.manual_seed(0)
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    m.check_nan = False
    m.train()
    opt = shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5)
    ref = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    ref.use_graphs = False
    ref.train()
    for it in range(110):
        before = m.flat_params.clone()
        opt.zero_grad()
        loss = m(x, mask, beta=1.0, temp=3.0)
        loss["n_elbo"].backward()
        g = m.flat_grads.clone()
        u, eps = m._last.get("u"), m._last.get("eps")
        if it >= 2:
            assert len(m._graphs) == 1 and u is not None, "the step was not replayed from a graph"
            ref.flat_params.copy_(before)
            ref.zero_grad()
            ref.set_noise(u, eps)
            l2 = ref(x, mask, beta=1.0, temp=3.0)
            l2["n_elbo"].backward()
            np.testing.assert_allclose(float(loss["n_elbo"].detach()), float(l2["n_elbo"].detach()), rtol=1e-5, err_msg="step %d" % it)
            err = float((g - ref.flat_grads).abs().max()) / float(ref.flat_grads.abs().max())
            assert err < 1e-3, (it, err)
        opt.step()


def test_captured_step_with_fills_inside_gemms_matches_oracle():
    """A captured step whose GEMM helper zero-fills inside the step (K = D*2H = 8192: the layer-0 input gradient splits
    K and adds atomically into cleared columns, main launch and the 16-column remainder of E = 528 alike): under capture
    those fills are store kernels (plan.hip zero2d_async).  Eager call, capture + replay, replay -- each against the oracle."""
    tl = shivae_amd.utils.make_types_list(4, 4, 4, [3] * 4, interleave=True)
    T, B, H, Z, K = 3, 24, 256, 64, 16
    rng = np.random.default_rng(2024)
    P = so.init_params(tl, H, Z, K, True, seed=3)
    x, mask = _synth(rng, tl, T, B, 0.3)
    u = rng.random((T, B, K)).astype(np.float32)
    eps = rng.standard_normal((T, B, Z)).astype(np.float32)
    om = so.OracleShiVAE(P, tl, H, Z, K, True)
    c = om.forward(x, mask, u, eps, 0.8, 1.5)
    if _golden.dekink_decoder(P, tl, c["emb"]):
        om = so.OracleShiVAE(P, tl, H, Z, K, True)
        c = om.forward(x, mask, u, eps, 0.8, 1.5)
    G = om.backward()
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.train()
    for call in range(3):
        m.zero_grad()
        m.set_noise(u, eps)
        loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.8, temp=1.5)
        loss["n_elbo"].backward()
        assert len(m._graphs) == (1 if call else 0)
        for k in shivae_amd.LOSS_KEYS:
            np.testing.assert_allclose(float(loss[k].detach()), c["losses"][k], rtol=LOSS_RTOL, atol=1e-5, err_msg="%s call %d" % (k, call))
        bad = _golden.bad_grads(_grads(m), G)
        assert not bad, (call, bad[:8])


def test_micro_batches_equal_one_batch():
    """Splitting the batch (what data parallelism and large-B runs do) leaves loss and grads unchanged."""
    tl, T, B, H, Z, K, learn_std, miss, _ = CASES["aligned_h32"]
    rng = np.random.default_rng(7)
    x, mask = _synth(rng, tl, T, B, miss)
    u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
    torch.manual_seed(0)
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=learn_std)
    m.train()
    m.set_noise(u, eps)
    l1 = m(torch.from_numpy(x), torch.from_numpy(mask))
    l1["n_elbo"].backward()
    g1 = m.flat_grads.clone()
    m.zero_grad()
    m.micro_batch = lambda T_, B_: 16
    m.set_noise(u, eps)
    l2 = m(torch.from_numpy(x), torch.from_numpy(mask))
    l2["n_elbo"].backward()
    np.testing.assert_allclose(float(l2["n_elbo"]), float(l1["n_elbo"]), rtol=1e-5)
    scale = float(g1.abs().max())
    assert float((m.flat_grads - g1).abs().max()) / scale < 1e-4


def test_train_loop_reduces_loss():
    """A few Adam steps through the reference-style loop (trainers.py:209-224) on the HIP path."""
    tl = shivae_amd.utils.make_types_list(1, 1, 1, [3])
    rng = np.random.default_rng(1)
    x, mask = _synth(rng, tl, 20, 32, 0.3)
    torch.manual_seed(0)
    m = shivae_amd.ShiVAE(10, 2, 3, 1, types_list=tl, learn_std=True)
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    xt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(mask).cuda()
    vals = []
    for _ in range(30):
        opt.zero_grad()
        loss = m(xt, mt, beta=1.0, temp=3)
        loss["n_elbo"].backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        opt.step()
        vals.append(float(loss["n_elbo"]))
    assert vals[-1] < vals[0]
