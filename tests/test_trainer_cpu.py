"""CPU checks of the host logic around the step: schedules, sharding arithmetic, gloo all-reduce."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import shivae_amd
from shivae_amd.trainers import KL_Annealing, LinearTempScheduler, EarlyStopping

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kl_annealing_matches_reference_formula():
    a = KL_Annealing(20)  # train.py:51-69
    assert a.update_beta(0) == 0.0 and a.update_beta(10) == 0.25 and a.update_beta(20) == 1.0 and a.update_beta(99) == 1.0
    assert KL_Annealing(0).update_beta(3) == 1.0


def test_temperature_schedule():
    s = LinearTempScheduler(init_temp=3, end_temp=0.1, annealing_epochs=9)  # trainers.py:109-111 at n_epochs=10
    np.testing.assert_allclose(s.update_temp(1), 3 - (2.9 / 9))
    assert s.update_temp(100) == 0.1


def test_early_stopping():
    e = EarlyStopping(patience=2, min_delta=0.2)
    assert [e.stop(v) for v in (10.0, 9.0, 8.95, 8.9)] == [False, False, False, True]


def test_gloo_allreduce_of_flat_gradients_world2():
    """Two CPU ranks: each fills its flat gradient buffer with its shard's contribution (scaled by
    1/B_global, as the step does); allreduce_grads() must leave the global-batch gradient on both."""
    script = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import shivae_amd
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        tl = shivae_amd.utils.make_types_list(1, 1, 1, [3])
        torch.manual_seed(0)
        m = shivae_amd.ShiVAE(10, 2, 3, 1, types_list=tl, learn_std=True, device="cpu")
        g = torch.Generator().manual_seed(5)
        per_seq = torch.randn(8, m.flat_grads.numel(), generator=g)      # gradient of each of 8 sequences
        shard = per_seq[r * 4:(r + 1) * 4]
        m.flat_grads.copy_(shard.sum(0) / 8.0)                           # local step with inv_batch = 1/B_global
        m.allreduce_grads()
        ref = per_seq.mean(0)
        assert torch.allclose(m.flat_grads, ref, atol=1e-6), float((m.flat_grads - ref).abs().max())
        same = [torch.zeros_like(m.flat_params) for _ in range(w)]
        dist.all_gather(same, m.flat_params)
        assert torch.equal(same[0], same[1])                             # identical replicas (same seed)
        dist.destroy_process_group()
        print("rank", r, "ok")
    """ % ROOT)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(script)
        path = f.name
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    try:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", "29533", path],
                             capture_output=True, text=True, env=env, timeout=240)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2
