"""Isolated timing of the likelihood / KL kernels through the C ABI: python tools/loss_bench.py [cfg3|cfg5]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shivae_amd
from shivae_amd import _native as N
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nr, npos, nb, cats, T, B, H, Z, K, miss = CONFIGS[name]
tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
plan = N.Plan(1, 16, H, Z, K, [N.TYPE_CODE[t["type"]] for t in tl], [int(t["dim"]) if t["type"] == "cat" else 1 for t in tl], True)
X, P = plan.x_dim, plan.p_dim
L = N.lib()


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rows in (16640, 36600, 1048576):
    x = torch.randn(rows, X, device="cuda")
    mask = (torch.rand(rows, X, device="cuda") < miss).to(torch.uint8)
    Pt = torch.randn(rows, P, device="cuda")
    dP = torch.empty_like(Pt)
    rl = torch.zeros(rows, 6, device="cuda")
    ms = timeit(lambda: N.check(L.shivae_heter_nll(plan.h, N.ptr(x), N.ptr(mask), N.ptr(Pt), P, N.ptr(dP), rows, 1.0, N.ptr(rl), N.stream())))
    by = rows * (5.0 * X + 8.0 * P)
    print("heter_nll %s rows=%8d X=%d P=%d  %8.3f ms  %7.1f GB/s" % (name, rows, X, P, ms, by / ms / 1e6), flush=True)
    Zp, Kp = (Z + 15) // 16 * 16, (K + 15) // 16 * 16
    mu, sd, mup, sdp = (torch.randn(rows, Zp, device="cuda") for _ in range(4))
    sd, sdp = sd.abs() + 0.1, sdp.abs() + 0.1
    pr = torch.softmax(torch.randn(rows, Kp, device="cuda"), -1)
    g = [torch.empty(rows, Zp, device="cuda") for _ in range(4)] + [torch.empty(rows, Kp, device="cuda")]
    ms = timeit(lambda: N.check(L.shivae_kl_terms(N.ptr(mu), N.ptr(sd), N.ptr(mup), N.ptr(sdp), Zp, N.ptr(pr), Kp, rows, Z, K, 1.0,
                                                  N.ptr(g[0]), N.ptr(g[1]), N.ptr(g[2]), N.ptr(g[3]), N.ptr(g[4]), N.ptr(rl), N.stream())))
    by = rows * (16.0 * Z + 4.0 * K) * 2
    print("kl_terms  %s rows=%8d Z=%d K=%d  %8.3f ms  %7.1f GB/s" % (name, rows, Z, K, ms, by / ms / 1e6), flush=True)
