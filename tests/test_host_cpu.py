"""CPU-only checks of the host layer: the C-ABI library loads and exports every declared symbol,
the model mirrors the reference's state_dict, and there is no silent CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import _golden
import shivae_amd
from shivae_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "shivae_hip.h")).read()
    declared = set(re.findall(r"\b(shivae_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"shivae_plan", "shivae_stream", "shivae_config"}
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    lib = ctypes.CDLL(N.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in N.lib().shivae_version()


@pytest.mark.parametrize("name", _golden.CASES)
def test_state_dict_keys_and_shapes_match_reference(name):
    g, cfg, sd = _golden.load(name)
    m = shivae_amd.ShiVAE(cfg["H"], cfg["Z"], cfg["K"], 1, types_list=cfg["types_list"], learn_std=cfg["learn_std"],
                          device="cpu")
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())  # same names in the same registration order
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    # parameters stay views of the one flat buffer after loading
    flat = m.flat_params
    for k, p in m.named_parameters():
        assert flat.data_ptr() <= p.data_ptr() < flat.data_ptr() + 4 * flat.numel(), k
        np.testing.assert_array_equal(p.detach().numpy(), sd[k])
    assert sum(p.numel() for p in m.parameters()) == sum(v.size for v in sd.values())


def test_param_count_cfg1():
    tl = shivae_amd.utils.make_types_list(1, 1, 1, [3])
    m = shivae_amd.ShiVAE(10, 2, 3, 1, types_list=tl, learn_std=True, device="cpu")
    assert sum(p.numel() for p in m.parameters()) == 5368


def test_init_scheme():
    tl = shivae_amd.utils.make_types_list(2, 1, 1, [4])
    m = shivae_amd.ShiVAE(16, 4, 3, 1, types_list=tl, learn_std=True, device="cpu")
    for k, p in m.named_parameters():
        if "bias" in k:
            assert float(p.abs().max()) == 0.0
    w = m.rnn.weight_hh_l0.detach()
    np.testing.assert_allclose((w.T @ w).numpy(), np.eye(16), atol=1e-5)  # orthogonal


def test_no_cpu_fallback():
    tl = shivae_amd.utils.make_types_list(1, 1, 1, [3])
    m = shivae_amd.ShiVAE(10, 2, 3, 1, types_list=tl, learn_std=True, device="cpu")
    x = torch.zeros(4, 2, 6)
    with pytest.raises(N.ShiVAENativeError):
        m(x, torch.zeros_like(x, dtype=torch.bool))


def test_plan_rejects_bad_config():
    with pytest.raises(N.ShiVAENativeError):
        N.Plan(4, 2, 10, 2, 3, [7], [1], True)
    with pytest.raises(N.ShiVAENativeError):
        N.Plan(4, 2, 10, 2, 3, [N.CAT], [1], True)
    with pytest.raises(N.ShiVAENativeError):
        N.Plan(4, 2, 1024, 2, 3, [N.REAL], [1], True)


def test_plan_layout_is_disjoint_and_complete():
    p = N.Plan(3, 2, 12, 3, 5, [0, 1, 2, 3, 3], [1, 1, 1, 4, 8], True)
    spans = []
    for tid in range(len(N.P)):
        for a in range(5):
            info = p.param_info(tid, a)
            if info is None:
                continue
            off, r, c = info
            spans.append((off, off + r * c))
    spans = sorted(set(spans))
    # stacked tensors appear both whole (attr-independent ids < DEC0) and per attribute: drop containers
    leaves = [s for s in spans if not any(o[0] <= s[0] and s[1] <= o[1] and o != s for o in spans)]
    for a, b in zip(leaves, leaves[1:]):
        assert a[1] <= b[0]
    assert leaves[-1][1] <= p.num_params
    assert p.x_dim == 1 + 1 + 1 + 4 + 8 and p.p_dim == 2 + 2 + 1 + 3 + 7


def test_synthetic_batch_shards_by_sequence_block():
    """bench.py --gpus N: rank r generates exactly rows [r*B/N, (r+1)*B/N) of the single-process batch."""
    import torch
    from shivae_amd.synth import synth_batch
    tl = shivae_amd.utils.make_types_list(2, 1, 1, [3, 4])
    x, m = synth_batch(tl, 50, 64, miss=0.3, seed=9, device="cpu", chunk=16)
    for r in range(4):
        xs, ms = synth_batch(tl, 50, 16, miss=0.3, seed=9, device="cpu", chunk=16, seq0=16 * r)
        assert torch.equal(xs, x[:, 16 * r:16 * (r + 1)]) and torch.equal(ms, m[:, 16 * r:16 * (r + 1)])
    assert 0.2 < float(m.float().mean()) < 0.4
    assert float(x[m].abs().max()) == 0.0  # zero-filled where missing


def test_bench_refuses_rank_counts_it_cannot_run():
    """bench.py --gpus N: more ranks than devices over RCCL, or a launch whose WORLD_SIZE differs from --gpus, exit
    non-zero without printing a result line (no GPU is touched before that decision)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SHIVAE_BENCH_BACKEND")}
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "%d ranks requested, %d device" % (n, n - 2) in r.stderr, r.stderr[-500:]
    assert "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr, r.stderr[-500:]
