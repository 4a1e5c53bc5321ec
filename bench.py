#!/usr/bin/env python3
"""Benchmark of the Shi-VAE train step (ELBO forward + backward + grad all-reduce + clip + Adam).

  python bench.py --gpus N --steps K --warmup W [--config cfg3]

One process per GPU (torch.distributed over RCCL when N > 1, launched by torch.distributed.run);
rank 0 prints ONE JSON line.  Metric: train sequence-timesteps/s = B_global*T / step time.
Default workload = BASELINE.json configs[2] ("cfg3"): synthetic heterogeneous HMM, global batch 8192,
T=1000, D=256 mixed types (64 real/64 pos/64 bin/64 cat, C=4), h_dim=256, z_dim=64, K=16, 30 % missing;
N GPUs split the SAME global batch (strong scaling, configs[3])."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import shivae_amd  # noqa: E402
from shivae_amd import _native as N  # noqa: E402
from shivae_amd.synth import synth_batch  # noqa: E402

CONFIGS = {
    # name: (n_real, n_pos, n_bin, cat_classes, T, B, H, Z, K, miss)
    "cfg1": (1, 1, 1, [3], 100, 64, 10, 2, 3, 0.3),                      # bundled HMM set (main_hmm.py)
    "cfg2": (35, 0, 0, [], 48, 64, 10, 35, 10, 0.3),                     # physionet-shaped stand-in
    "cfg3": (64, 64, 64, [4] * 64, 1000, 8192, 256, 64, 16, 0.3),        # BASELINE.json configs[2]
    "cfg5": (52, 51, 51, [8] * 358, 1000, 8192, 256, 64, 16, 0.9),       # categorical-heavy, 90 % missing
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cfgname, types_list, H, Z, K):
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
    t_end = time.time() + 25
    for step in range(1, 4):
        t0 = time.time()
        om = so.OracleShiVAE(P, types_list, H, Z, K, True, dtype=np.float32)
        om.forward(x, mask, u, eps, 1.0, 3.0)
        G = om.backward()
        so.clip_adam_step(P, G, m, v, step)
        times.append(time.time() - t0)
        if time.time() > t_end:
            break
    dt = sorted(times)[len(times) // 2]
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    return {"value": Bc * Tc / dt, "unit": "sequence-timesteps/s", "cores": cores, "kind": "port",
            "sample": "numpy oracle (fp32), %s model, B=%d T=%d, fwd+bwd+clip+Adam, median of %d steps"
                      % (cfgname, Bc, Tc, len(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override the global batch (parity/debug runs)")
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
    torch.manual_seed(0)  # same replica on every rank (reference init scheme, seed 0)
    model = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    model.check_nan = False
    model.train()
    opt = shivae_amd.FusedClipAdam(model, lr=5e-3, clip=0.5)
    x, mask = synth_batch(tl, T, Bl, miss=miss, seed=1234 + rank, device="cuda")
    torch.manual_seed(100 + rank)  # per-rank noise stream (device generator)

    trace = os.environ.get("SHIVAE_BENCH_TRACE") is not None

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

    if rank == 0:  # a full-size step takes about a minute: keep stderr alive for watchdogs (no GPU work here)
        import threading
        t_start = time.perf_counter()
        def heartbeat():
            while True:
                time.sleep(30)
                print("[bench] running, %.0f s elapsed" % (time.perf_counter() - t_start), file=sys.stderr, flush=True)
        threading.Thread(target=heartbeat, daemon=True).start()
    for _ in range(a.warmup):
        loss = step()
    fence()
    # Per-kernel HIP events (the roofline figures).  On one GPU they bracket the launches of the timed steps
    # themselves.  With several ranks the timed region stays free of them (no per-launch events or pinned staging
    # next to the collective), and rank 0 profiles one extra local step -- no all-reduce, no optimiser -- afterwards.
    inline_prof = world == 1
    if inline_prof:
        N.profile_start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof = N.profile_stop() if inline_prof else []
    prof_steps = a.steps
    if not inline_prof:
        if rank == 0:
            N.profile_start()
            t1 = time.perf_counter()
            opt.zero_grad()
            model(x, mask, beta=1.0, temp=3.0, inv_batch=1.0 / B)["n_elbo"].backward()
            torch.cuda.synchronize()
            prof_dt = time.perf_counter() - t1
            prof = N.profile_stop()
            prof_steps = 1
        fence()
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
        lv = loss["n_elbo"].detach().clone()
        dist.all_reduce(lv)
    else:
        lv = loss["n_elbo"].detach()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    ms_step = dt / a.steps * 1e3
    value = B * T * a.steps / dt

    prof_ms = dt * 1e3 if inline_prof else prof_dt * 1e3  # wall time of the profiled step(s)
    gemms = [p for p in prof if p["id"] < 18]
    dom = max(gemms, key=lambda p: p["ms"]) if gemms else None
    roof = None
    if dom:
        ach = dom["work"] / (dom["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None, "kernel": dom["name"], "dtype": "f32",
                "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                "share_of_step": dom["ms"] / prof_ms}
    others = []
    for p in prof:
        if dom is not None and p["id"] == dom["id"]:
            continue
        e = {"kernel": p["name"], "launches": p["launches"], "avg_launch_ms": p["ms"] / p["launches"],
             "share_of_step": p["ms"] / prof_ms}
        rate = p["work"] / (p["ms"] * 1e-3)
        if p["id"] >= 20:
            e.update(bound="hbm", achieved=rate / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=rate / 1e9 / PEAK_HBM_GBS)
        else:
            e.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                     frac=rate / 1e12 / PEAK_F32_MFMA_TFLOPS)
        others.append(e)
    out = {"metric": "train sequence-timesteps/sec (ELBO fwd+bwd)", "value": value, "unit": "sequence-timesteps/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s: synthetic heterogeneous HMM, global batch %d, T=%d, D=%d (%d real/%d pos/%d bin/%d cat), "
                                  "h_dim=%d, z_dim=%d, K=%d, %d%% missing; fwd+bwd+allreduce+clip+Adam"
                                  % (a.config, B, T, len(tl), nr, npos, nb, len(cats), H, Z, K, int(miss * 100)),
                      "global_batch": B, "seq_len": T, "parallelism": "dp%d" % world,
                      "micro_batch": model.micro_batch(T, Bl), "params": int(sum(p.numel() for p in model.parameters())),
                      "missing_frac": round(float(mask.float().mean()), 4)},
           "n_elbo": float(lv), "roofline": roof, "roofline_other_kernels": others}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.config, tl, H, Z, K)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
