#!/usr/bin/env python3
"""Benchmark of the Shi-VAE train step (ELBO forward + backward + grad all-reduce + clip + Adam).

  python bench.py --gpus N --steps K --warmup W [--config cfg3]

One process per GPU (torch.distributed over RCCL when N > 1).  Run under torch.distributed.run it is one rank;
run plainly with --gpus N > 1 it starts the N ranks itself as a child process (launch_ranks) -- and refuses, with a
non-zero exit and no result line, a rank count it cannot run (more ranks than devices; --gpus != WORLD_SIZE).
Rank 0 prints ONE JSON line.  Metric: train sequence-timesteps/s = B_global*T / median step time.
Default workload = BASELINE.json configs[2] ("cfg3"): synthetic heterogeneous HMM, global batch 8192,
T=1000, D=256 mixed types (64 real/64 pos/64 bin/64 cat, C=4), h_dim=256, z_dim=64, K=16, 30 % missing;
N GPUs shard the SAME global batch by sequence (strong scaling, configs[3]): rank r generates exactly rows
[r*B/N, (r+1)*B/N) of the batch a single process would generate.

Wall-clock budget: one full-size fp32 step is tens of seconds, so the requested warm-up / step counts are
clamped to what fits SHIVAE_BENCH_BUDGET_S (default 480 s, counted from process start; at least 2 warm-up and
3 timed steps).  The JSON line reports the counts that actually ran (`steps`, `warmup`) next to the requested
ones.  Order: set-up -> bounded CPU baseline (rank 0, N=1) -> cold warm-up step -> one warm step under the
library's per-launch HIP events (the roofline figures; never inside the timed region) -> remaining warm-ups ->
timed steps (per-step host clock after a device sync; max over ranks; the median is reported) -> one step of the
OTHER arithmetic of the decoder GEMMs when >= 8 headline steps fit beside it.

Arithmetic (--arith): both compute fp32 products of fp32 data into fp32.  `split` (default; DESIGN.md s.4 "Split
mode") forms every product from six bf16 MFMA products of an exact 3-way bf16 split, with error at or below the
fp32-input MFMA's on adversarial operands (tests/test_gpu_split_mode.py); `f32` is the fp32-input MFMA itself.  The
line names the arithmetic of `value` and carries the other one as `f32_mfma_mode` / `split_bf16_mode`."""
import argparse
import json
import math
import os
import sys
import time

T_PROC0 = time.perf_counter()
if os.environ.get("SHIVAE_BENCH_T0"):  # a rank started by launch_ranks(): the budget counts from the parent's start
    T_PROC0 -= max(0.0, time.time() - float(os.environ["SHIVAE_BENCH_T0"]))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# torch and the package are imported by main() AFTER the rank launcher below has had its say: the parent of an
# N-rank run never touches the GPU
torch = shivae_amd = N = synth_batch = None


def _imports():
    global torch, shivae_amd, N, synth_batch
    import torch as _torch
    import shivae_amd as _pkg
    from shivae_amd import _native as _native
    from shivae_amd.synth import synth_batch as _synth
    torch, shivae_amd, N, synth_batch = _torch, _pkg, _native, _synth

CONFIGS = {
    # name: (n_real, n_pos, n_bin, cat_classes, T, B, H, Z, K, miss)
    "cfg1": (1, 1, 1, [3], 100, 64, 10, 2, 3, 0.3),                      # bundled HMM set (main_hmm.py)
    "cfg2": (35, 0, 0, [], 48, 64, 10, 35, 10, 0.3),                     # physionet-shaped stand-in
    "cfg3": (64, 64, 64, [4] * 64, 1000, 8192, 256, 64, 16, 0.3),        # BASELINE.json configs[2]
    "cfg5": (52, 51, 51, [8] * 358, 1000, 8192, 256, 64, 16, 0.9),       # categorical-heavy, 90 % missing
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
# split arithmetic: an fp32 product = 6 bf16 MFMA products, so its ceiling is the dense bf16 MFMA peak (~2.5 PFLOP/s,
# same guide) / 6, in fp32-equivalent flops
PEAK_SPLIT_TFLOPS = 2500.0 / 6.0
ARITH = {
    "split": "fp32 in / fp32 out; every fp32 operand split exactly into 3 bf16 terms (nearest), 6 bf16 MFMA products per "
             "fp32 product, leading product and corrections in separate fp32 accumulators (shivae_set_gemm_split(1)); "
             "error <= the fp32-input MFMA's on adversarial operands (tests/test_gpu_split_mode.py)",
    "f32": "fp32-input MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain; shivae_set_gemm_split(0))",
}
PEAK_HBM_GBS = 8000.0


def host_cores():
    """(physical cores of the host, CPUs this process may run on)."""
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    return len(phys) or (os.cpu_count() or 1), aff


def cpu_baseline(cfgname, types_list, H, Z, K, seconds=12.0):
    """The numpy oracle (a port of the reference algorithm) timed on the host cores on a bounded sample."""
    import numpy as np
    from oracle import shivae_oracle as so
    Bc, Tc = (64, 100) if cfgname in ("cfg1", "cfg2") else (16, 16)
    rng = np.random.default_rng(0)
    P = so.init_params(types_list, H, Z, K, True, seed=0, dtype=np.float32)
    X = so.x_dim(types_list)
    x = rng.standard_normal((Tc, Bc, X)).astype(np.float32)
    mask = rng.random((Tc, Bc, X)) < 0.3
    x = np.where(mask, 0, x).astype(np.float32)
    u = rng.random((Tc, Bc, K)).astype(np.float32)
    eps = rng.standard_normal((Tc, Bc, Z)).astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v = {k: np.zeros_like(v) for k, v in P.items()}
    times = []
    t_end = time.time() + seconds
    for step in range(1, 9):
        t0 = time.time()
        om = so.OracleShiVAE(P, types_list, H, Z, K, True, dtype=np.float32)
        om.forward(x, mask, u, eps, 1.0, 3.0)
        G = om.backward()
        so.clip_adam_step(P, G, m, v, step)
        times.append(time.time() - t0)
        if step >= 3 and time.time() > t_end:
            break
    dt = sorted(times)[len(times) // 2]
    # SURVEY.md 8d's like-for-like figure: stock PyTorch-CPU ops arranged as the reference arranges them (Python loop
    # over t, inside it a Python loop over the attribute decoders, autograd backward, clip_grad_norm_, Adam)
    torch_loop = None
    try:
        from oracle import torch_cpu_step as tc
        Pt = tc.make_params(so.init_params(types_list, H, Z, K, True, seed=0, dtype=np.float32))
        opt = torch.optim.Adam(list(Pt.values()), lr=5e-3)
        tt, t_end = [], time.time() + 0.6 * seconds
        for _ in range(4):
            t0 = time.time()
            tc.train_step(Pt, opt, types_list, H, Z, K, x, mask, u, eps, 1.0, 3.0, True, 0.5)
            tt.append(time.time() - t0)
            if len(tt) >= 2 and time.time() > t_end:
                break
        torch_loop = {"value": Bc * Tc / min(tt[1:] or tt), "unit": "sequence-timesteps/s", "torch_threads": torch.get_num_threads(),
                      "sample": "oracle/torch_cpu_step.py (fp32, reference-style loops, autograd), same model and "
                                "B=%d T=%d, best of %d steps after one warm-up" % (Bc, Tc, max(1, len(tt) - 1))}
        del Pt, opt
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        torch_loop = {"error": repr(e)}
    threads = None
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        pass
    phys, aff = host_cores()
    if threads is None:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    used = min(threads, aff)
    return {"value": Bc * Tc / dt, "unit": "sequence-timesteps/s", "cores": used, "kind": "port",
            "host_physical_cores": phys, "cpus_allowed": aff,
            "threads": "BLAS pool of %d threads (OMP_NUM_THREADS=%s)" % (threads, os.environ.get("OMP_NUM_THREADS", "unset")),
            "sample": "numpy oracle (fp32), %s model, B=%d T=%d, fwd+bwd+clip+Adam, median of %d steps"
                      % (cfgname, Bc, Tc, len(times)),
            "torch_cpu_reference_style": torch_loop}


def pmc_traffic(cfgname, B, T, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this workload
    (profiles/, collected per MI355X_MICROARCH.md), or None when no pass of this configuration is on file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tab = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    for e in tab.get("entries", []):
        if e.get("config") == cfgname and e.get("global_batch") == B and e.get("seq_len") == T and e.get("kernel") == kernel:
            return e.get("hbm_bytes_per_launch"), e.get("source")
    return None, None


def visible_gpus():
    """Devices this process could use, WITHOUT initialising the GPU runtime (the launcher parent must stay clean)."""
    import torch as _torch
    return int(_torch.cuda.device_count())


def launch_ranks(a, argv):
    """`bench.py --gpus N` outside a torch.distributed.run launch: start N ranks (one per GPU, RCCL) as a CHILD
    process -- this parent has not touched the GPU and never execs --, forward the child's output and exit code.
    SHIVAE_BENCH_BACKEND=gloo lets the ranks share devices (rehearsals of the N>1 path on a box with fewer GPUs)."""
    import socket
    import subprocess
    backend = os.environ.get("SHIVAE_BENCH_BACKEND", "nccl")
    ndev = visible_gpus()
    if backend == "nccl" and a.gpus > ndev:
        print("bench.py: %d ranks requested, %d device%s visible (one rank per GPU over RCCL; set "
              "SHIVAE_BENCH_BACKEND=gloo to rehearse with shared devices)" % (a.gpus, ndev, "" if ndev == 1 else "s"),
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, SHIVAE_BENCH_T0=repr(time.time() - (time.perf_counter() - T_PROC0)))
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override the global batch (parity/debug runs)")
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-step", action="store_true",
                    help="skip the warm-up step under the library's HIP events (runs under rocprofv3 --pmc)")
    ap.add_argument("--arith", choices=["split", "f32"], default=os.environ.get("SHIVAE_BENCH_ARITH", "split"),
                    help="arithmetic of the decoder GEMMs in the timed steps (both compute fp32 products of fp32 data)")
    ap.add_argument("--no-other-mode", "--no-split-mode", dest="no_other_mode", action="store_true",
                    help="skip the short comparison leg in the other arithmetic (reported beside the headline)")
    ap.add_argument("--budget", type=float, default=float(os.environ.get("SHIVAE_BENCH_BUDGET_S", "480")),
                    help="wall-clock seconds for the whole process; counts are clamped to fit")
    a = ap.parse_args()

    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:  # never label an N-rank request with another rank count
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or plain `python bench.py "
                  "--gpus %d`, which starts the ranks itself)" % (a.gpus, world, a.gpus, a.gpus), file=sys.stderr, flush=True)
        sys.exit(2)
    _imports()
    # (one rank per GPU; SHIVAE_BENCH_BACKEND=gloo lets several ranks rehearse the N>1 path on a box with fewer GPUs)
    backend = os.environ.get("SHIVAE_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if backend == "nccl":
        assert local < ndev, "one rank per GPU: LOCAL_RANK %d but %d devices" % (local, ndev)
    local %= ndev
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    nr, npos, nb, cats, T, B, H, Z, K, miss = CONFIGS[a.config]
    B = a.batch or B
    T = a.T or T
    assert B % world == 0, "global batch must divide over the ranks"
    Bl = B // world

    def say(msg):
        if rank == 0:
            print("[bench %6.1f s] %s" % (time.perf_counter() - T_PROC0, msg), file=sys.stderr, flush=True)

    tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
    torch.manual_seed(0)  # same replica on every rank (reference init scheme, seed 0)
    model = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    model.check_nan = False
    model.overlap_allreduce = True  # N > 1: the decoder's gradient buckets start before the reverse-time scan
    model.train()
    N.lib().shivae_set_gemm_split(1 if a.arith == "split" else 0)
    opt = shivae_amd.FusedClipAdam(model, lr=5e-3, clip=0.5)
    # one global batch (seed 1234), sharded by sequence: blocks of `blk` sequences have their own generator streams
    blk = max(1, math.gcd(512, B // 16)) if B % 16 == 0 else B
    assert (rank * Bl) % blk == 0, "shard of %d sequences does not start on a block of %d" % (Bl, blk)
    x, mask = synth_batch(tl, T, Bl, miss=miss, seed=1234, device="cuda", chunk=blk, seq0=rank * Bl)
    n_missing = sum(int(torch.count_nonzero(mask[t0:t0 + 8])) for t0 in range(0, T, 8))  # (in slabs: no full-size temporaries)
    torch.manual_seed(100 + rank)  # per-rank noise stream (device generator)
    micro = model.micro_batches(T, Bl)
    torch.cuda.synchronize()
    say("set-up done: %s B=%d (local %d, micro-batches %s) T=%d, %d parameters"
        % (a.config, B, Bl, micro, T, sum(p.numel() for p in model.parameters())))

    trace = os.environ.get("SHIVAE_BENCH_TRACE") is not None
    if trace and os.environ["SHIVAE_BENCH_TRACE"].startswith("dump"):  # e.g. dump40: all Python stacks after 40 s
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SHIVAE_BENCH_TRACE"][4:] or 40), exit=False)

    def tr(msg):
        if trace:
            if os.environ["SHIVAE_BENCH_TRACE"] == "sync":
                torch.cuda.synchronize()
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    def step():
        opt.zero_grad()
        tr("zero_grad done")
        loss = model(x, mask, beta=1.0, temp=3.0, inv_batch=1.0 / B)
        tr("forward done")
        loss["n_elbo"].backward()
        tr("backward done")
        model.allreduce_grads()
        tr("allreduce done")
        opt.step()
        tr("optimizer done")
        return loss

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:  # a full-size step takes most of a minute: keep stderr alive for watchdogs (no GPU work here)
        import threading

        def heartbeat():
            while True:
                time.sleep(30)
                say("running")
        threading.Thread(target=heartbeat, daemon=True).start()

    # ---- CPU baseline first, so that the one JSON line can go out the moment the timed steps end
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.config, tl, H, Z, K)
        say("cpu baseline: %.1f seq-ts/s on %d threads" % (cpu["value"], cpu["cores"]))

    # ---- warm-up 1 (cold: workspace allocation, plan binding), warm-up 2 under the per-launch HIP events
    fence()
    loss = step()
    fence()
    say("cold warm-up step done")
    prof, n_warm = [], 1
    if a.no_profile_step:
        prof_dt = 1e-3  # (no projection: the caller asked for exact small counts)
    else:
        if rank == 0:
            N.profile_start()
        t1 = time.perf_counter()
        loss = step()
        fence()
        prof_dt = time.perf_counter() - t1
        prof = N.profile_stop() if rank == 0 else []
        n_warm = 2
        say("profiled warm-up step: %.3f s" % prof_dt)

    # ---- how many more steps fit the budget (same decision on every rank)
    # (the other arithmetic is timed afterwards on 1 warm-up + 1 step, when that fits next to at least 8 headline steps;
    # an fp32-MFMA step takes ~1.35 split steps)
    left = a.budget - (time.perf_counter() - T_PROC0) - 5.0
    other = "f32" if a.arith == "split" else "split"
    want_other = not a.no_other_mode and not a.no_profile_step
    other_cost = 2 * (1.4 if other == "f32" else 0.8) * prof_dt if want_other else 0.0
    if want_other and (left - other_cost) / (prof_dt * 1.02) < min(a.steps, 8):
        want_other, other_cost = False, 0.0
    left -= other_cost
    afford = int(left / (prof_dt * 1.02))
    if dist is not None:
        tt = torch.tensor([afford], device="cuda", dtype=torch.int64)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        afford = int(tt)
    steps = max(min(a.steps, 3), min(a.steps, afford))
    extra_warm = max(0, min(a.warmup - n_warm, afford - steps))
    for _ in range(extra_warm):
        loss = step()
    n_warm += extra_warm
    say("timing %d steps (requested %d) after %d warm-ups (requested %d)" % (steps, a.steps, n_warm, a.warmup))

    # ---- timed region: no profiler, no host work besides the launches; one device sync + clock read per step
    fence()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(steps):
        loss = step()
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    # ---- the same step in the OTHER arithmetic of the decoder GEMMs, reported beside the headline, never instead of it
    split_times = []
    if dist is not None:
        ws = torch.tensor([1 if want_other else 0], device="cuda")
        dist.all_reduce(ws, op=dist.ReduceOp.MIN)
        want_other = bool(int(ws))
    if want_other:
        N.lib().shivae_set_gemm_split(1 if other == "split" else 0)
        step()
        fence()
        m0 = time.perf_counter()
        for _ in range(1):
            loss_s = step()
            torch.cuda.synchronize()
            m1 = time.perf_counter()
            split_times.append(m1 - m0)
            m0 = m1
        fence()
        N.lib().shivae_set_gemm_split(1 if a.arith == "split" else 0)
        say("%s arithmetic: %s s" % (other, ", ".join("%.3f" % t for t in split_times)))
    per = torch.tensor([marks[i + 1] - marks[i] for i in range(steps)] + [dt] + split_times, device="cuda", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(per, op=dist.ReduceOp.MAX)
        lv = loss["n_elbo"].detach().clone()
        dist.all_reduce(lv)
    else:
        lv = loss["n_elbo"].detach()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    per = per.tolist()
    split_times = [per.pop() for _ in split_times][::-1]
    dt = per.pop()
    med = sorted(per)[len(per) // 2] if len(per) % 2 else 0.5 * (sorted(per)[len(per) // 2 - 1] + sorted(per)[len(per) // 2])
    ms_step = med * 1e3
    value = B * T / med

    prof_ms = prof_dt * 1e3  # wall time of the profiled step
    gemms = [p for p in prof if p["id"] < 18 or p["id"] in (24, 25)]
    dom = max(gemms, key=lambda p: p["ms"]) if gemms else None
    roof = None
    if dom:
        ach = dom["work"] / (dom["ms"] * 1e-3) / 1e12
        traffic, tsrc = pmc_traffic(a.config, B, T, dom["name"])
        split_kernel = dom["id"] in (24, 25) or dom["id"] % 6 == 5
        peak = PEAK_SPLIT_TFLOPS if split_kernel else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc,
                "kernel": dom["name"], "dtype": "f32",
                "peak_is": ("dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMA products per fp32 product (fp32-equivalent flops)"
                            if split_kernel else "fp32-input MFMA dense peak"),
                "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                "share_of_step": dom["ms"] / prof_ms,
                "measured": "HIP events around every launch of one warm, untimed step"}
    others = []
    for p in prof:
        if dom is not None and p["id"] == dom["id"]:
            continue
        e = {"kernel": p["name"], "launches": p["launches"], "avg_launch_ms": p["ms"] / p["launches"],
             "share_of_step": p["ms"] / prof_ms}
        rate = p["work"] / (p["ms"] * 1e-3)
        if 20 <= p["id"] <= 23:
            e.update(bound="hbm", achieved=rate / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=rate / 1e9 / PEAK_HBM_GBS)
        else:
            pk = PEAK_SPLIT_TFLOPS if (p["id"] in (24, 25) or (p["id"] < 18 and p["id"] % 6 == 5)) else PEAK_F32_MFMA_TFLOPS
            e.update(bound="mfma", achieved=rate / 1e12, peak=pk, unit="TFLOP/s", frac=rate / 1e12 / pk)
        others.append(e)
    out = {"metric": "train sequence-timesteps/sec (ELBO fwd+bwd)", "value": value, "unit": "sequence-timesteps/s",
           "n_gpus": world, "steps": steps, "warmup": n_warm, "steps_requested": a.steps, "warmup_requested": a.warmup,
           "ms_per_step": ms_step, "ms_per_step_mean": dt / steps * 1e3, "ms_per_step_min": min(per) * 1e3,
           "ms_per_step_max": max(per) * 1e3, "statistic": "median of the timed steps (max over ranks per step)",
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "arithmetic": ARITH[a.arith],
           "config": {"workload": "%s: synthetic heterogeneous HMM, global batch %d, T=%d, D=%d (%d real/%d pos/%d bin/%d cat), "
                                  "h_dim=%d, z_dim=%d, K=%d, %d%% missing; fwd+bwd+allreduce+clip+Adam"
                                  % (a.config, B, T, len(tl), nr, npos, nb, len(cats), H, Z, K, int(miss * 100)),
                      "global_batch": B, "seq_len": T, "parallelism": "dp%d" % world,
                      "micro_batch": micro[0], "micro_batches_per_rank": len(micro),
                      "params": int(sum(p.numel() for p in model.parameters())),
                      "missing_frac": round(n_missing / mask.numel(), 4)},
           "wall_s": time.perf_counter() - T_PROC0, "budget_s": a.budget,
           "n_elbo": float(lv), "roofline": roof, "roofline_other_kernels": others}
    if split_times:
        sm = sorted(split_times)[len(split_times) // 2]
        out["f32_mfma_mode" if other == "f32" else "split_bf16_mode"] = {
            "value": B * T / sm, "unit": "sequence-timesteps/s", "ms_per_step": sm * 1e3, "steps": len(split_times),
            "warmup": 1, "n_elbo": float(loss_s["n_elbo"].detach()) if dist is None else None,
            "arithmetic": ARITH[other]}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
