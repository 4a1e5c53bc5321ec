"""The N>1 path of bench.py (batch sharding, gradient all-reduce, max-over-ranks timing, one JSON line from rank 0),
rehearsed with two ranks that share the one GPU of the test box over gloo.  RCCL needs one GPU per rank and is
exercised by the 8-GPU scaling runs only."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SHIVAE_BENCH_T0")}
    env.update(kw)
    return env


def test_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` itself starts the two ranks (bench.py launch_ranks)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config",
           "cfg1", "--batch", "16", "--T", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(SHIVAE_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "strong"
    assert j["config"]["global_batch"] == 16 and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and j["n_elbo"] == j["n_elbo"]  # finite
    assert j["roofline"] is not None and "cpu_baseline" not in j


def test_more_ranks_than_devices_is_refused():
    """RCCL needs one GPU per rank: `bench.py --gpus 8` on a box with fewer devices must fail, not print a line."""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "cfg1"], cwd=ROOT,
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "%d ranks requested, %d device" % (n, n - 1) in r.stderr, r.stderr[-1000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_rank_count_mismatch_is_refused():
    """A launch whose WORLD_SIZE differs from --gpus must not label its line with either number."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg1"], cwd=ROOT,
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr, r.stderr[-1000:]


_DP_SCRIPT = r'''
import os, sys, tempfile, types
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import shivae_amd
from test_gpu_step import _synth

dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
tl = shivae_amd.utils.make_types_list(2, 1, 1, [3, 5])
T, B, H, Z, K = 12, 32, 24, 6, 4
rng = np.random.default_rng(11)          # the SAME global batch and noise on every rank
x, mask = _synth(rng, tl, T, B, 0.3)
u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
xt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(mask).cuda()
torch.manual_seed(0)
m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
m.train()
p0 = m.flat_params.clone()

# single-process step on the global batch (trainers.py:209-224)
opt = shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5)
opt.zero_grad()
m.set_noise(u, eps)
m(xt, mt, beta=1.0, temp=3)["n_elbo"].backward()
g_ref = m.flat_grads.clone()
opt.step()
p_ref = m.flat_params.clone()

# the same step data-parallel: rank r takes rows [r*B/w, (r+1)*B/w) of x, mask, u, eps (SURVEY.md 8e)
sl = slice(r * B // w, (r + 1) * B // w)
m.flat_params.copy_(p0)
opt = shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5)
opt.zero_grad()
m.set_noise(u[:, sl], eps[:, sl])
m(xt[:, sl], mt[:, sl], beta=1.0, temp=3, inv_batch=1.0 / B)["n_elbo"].backward()
m.allreduce_grads()
eg = float((m.flat_grads - g_ref).abs().max() / g_ref.abs().max())
opt.step()
ep = float((m.flat_params - p_ref).abs().max())
assert eg < 1e-4, ("all-reduced gradient", eg)
assert ep < 1e-4, ("parameters after clip+Adam", ep)

# and through Trainer.train_epoch, which has to find 1/B_global by itself
m.flat_params.copy_(p0)
args = types.SimpleNamespace(l_rate=5e-3, n_epochs=1, kl_annealing_epochs=0, clip=0.5, save_every=100, print_every=100,
                             save_model=False, result_dir=tempfile.mkdtemp(prefix="shivae_dp%%d_" %% r), restore=False)
tr = shivae_amd.Trainer(m, shivae_amd.FusedClipAdam(m, lr=5e-3), args)
m.set_noise(u[:, sl], eps[:, sl])
shard = (np.ascontiguousarray(x[:, sl].transpose(1, 0, 2)), np.ascontiguousarray(mask[:, sl].transpose(1, 0, 2)), None)
assert m.overlap_allreduce is False
logged = tr.train_epoch([shard], 1)
assert m.overlap_allreduce is False          # (train_epoch restores the caller's setting)
et = float((m.flat_params - p_ref).abs().max())
assert et < 1e-4, ("Trainer.train_epoch under data parallel", et)
single = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
single.flat_params.copy_(p0); single.eval(); single.set_noise(u, eps)
ref_loss = float(single(xt, mt, beta=1.0, temp=3)["n_elbo"])
assert abs(logged["n_elbo"] - ref_loss) < 1e-4 * abs(ref_loss), (logged["n_elbo"], ref_loss)
# two batches per epoch: the second iteration's collectives come while the first one's kernels may still be in flight
tr.train_epoch([shard, shard], 2)
assert bool(torch.isfinite(m.flat_params).all())
# ranks whose loaders disagree on the number of batches must fail loudly instead of hanging in a collective
try:
    tr.train_epoch([shard] * (1 + r), 3)
    raise SystemExit("mismatched loaders were not refused")
except RuntimeError as e:
    assert "same number" in str(e)
print("rank %%d ok: grad %%.2e params %%.2e trainer %%.2e" %% (r, eg, ep, et), flush=True)
dist.destroy_process_group()
'''


def test_two_ranks_match_single_process(tmp_path):
    """Sharded + all-reduced gradients and the parameters after clip+Adam equal the single-process step on the same
    global batch (SURVEY.md 8e partitioning), directly and through Trainer.train_epoch."""
    script = tmp_path / "dp_check.py"
    script.write_text(_DP_SCRIPT % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count(" ok: ") == 2, r.stdout[-1500:]


_OVERLAP_SCRIPT = r'''
import os, sys, ctypes
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import shivae_amd
from shivae_amd import _native as N
from test_gpu_step import _synth

os.environ["SHIVAE_FORCE_DP"] = "1"           # take the data-parallel code path with one rank
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
tl = shivae_amd.utils.make_types_list(2, 1, 1, [3, 5])
T, B, H, Z, K = 10, 40, 24, 6, 4
rng = np.random.default_rng(3)
x, mask = _synth(rng, tl, T, B, 0.3)
u, eps = rng.random((T, B, K)).astype(np.float32), rng.standard_normal((T, B, Z)).astype(np.float32)
res = []
for overlap, micro in ((False, 0), (True, 0), (True, 16)):
    torch.manual_seed(0)
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    m.train()
    m.overlap_allreduce = overlap
    m.bucket_floats = 1000                     # several buckets
    if micro:
        m.micro_batch = lambda T_, B_: micro
    m.zero_grad()
    m.set_noise(u, eps)
    loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.9, temp=2.0)
    assert bool(m._pending_reduce) == overlap
    loss["n_elbo"].backward()
    m.allreduce_grads()
    assert not m._pending_reduce
    torch.cuda.synchronize()
    res.append((float(loss["n_elbo"]), m.flat_grads.clone()))
for l, g in res[1:]:
    assert abs(l - res[0][0]) < 1e-5 * abs(res[0][0]), (l, res[0][0])
    assert float((g - res[0][1]).abs().max()) < 1e-4 * float(res[0][1].abs().max())

# gradient accumulation: a second training forward while the first one's buckets are pending must be refused (it
# would re-reduce already rank-summed values); with the overlap off, two forwards accumulate and reduce once
torch.manual_seed(0)
m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
m.train(); m.overlap_allreduce = True; m.bucket_floats = 1000
m.zero_grad(); m.set_noise(u, eps)
m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.9, temp=2.0)
assert m._pending_reduce
try:
    m.set_noise(u, eps)
    m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.9, temp=2.0)
    raise SystemExit("second forward with pending buckets was not refused")
except RuntimeError as e:
    assert "allreduce_grads" in str(e)
m.allreduce_grads()
assert not m._pending_reduce
m.overlap_allreduce = False
m.zero_grad()
for _ in range(2):
    m.set_noise(u, eps)
    m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.9, temp=2.0)["n_elbo"].backward()
m.allreduce_grads()
torch.cuda.synchronize()
assert float((m.flat_grads - 2 * res[0][1]).abs().max()) < 2e-4 * float(res[0][1].abs().max())

# the C-ABI all-reduce on a communicator of its own (one rank: the sum is the buffer itself)
L = N.lib()
uid = ctypes.create_string_buffer(128)
N.check(L.shivae_comm_unique_id(uid))
comm = ctypes.c_void_p()
N.check(L.shivae_comm_init(ctypes.byref(comm), 1, 0, uid))
g = torch.randn(100003, device="cuda")
g0 = g.clone()
N.check(L.shivae_allreduce_grads(N.ptr(g), g.numel(), comm, 30000, N.stream()))
torch.cuda.synchronize()
assert torch.equal(g, g0)
N.check(L.shivae_comm_destroy(comm))
dist.destroy_process_group()
print("overlap ok", flush=True)
'''


def test_bucketed_allreduce_path_single_rank_rccl(tmp_path):
    """The overlapped, bucketed gradient all-reduce (phases of the step + async RCCL buckets + the small remainder)
    gives the gradients of the plain step; and shivae_allreduce_grads runs on an RCCL communicator made through the
    C ABI.  One rank only: this box has one GPU, and RCCL refuses two ranks on one device."""
    script = tmp_path / "overlap_check.py"
    script.write_text(_OVERLAP_SCRIPT % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "overlap ok" in r.stdout
