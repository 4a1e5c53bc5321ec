"""HBM rate of the scaler pre/post pass (csrc/prep.hip): python tools/prep_bench.py [cfg3|cfg5]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shivae_amd
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nr, npos, nb, cats, T, B, H, Z, K, miss = CONFIGS[name]
tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
sc = shivae_amd.HeterogeneousScaler(tl)
D, X = len(tl), sc.x_dim
sc.mean[:] = 0.5; sc.std[:] = 2.0
n, t = 2048, 1000
g = torch.Generator(device="cuda").manual_seed(0)
data = torch.rand(n, t, D, device="cuda", generator=g) * 3.0
for i, ty in enumerate(tl):
    if ty["type"] == "cat":
        data[..., i] = torch.floor(data[..., i] / 3.0 * int(ty["dim"])).clamp(max=int(ty["dim"]) - 1)
mask = torch.rand(n, t, D, device="cuda", generator=g) < miss


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

from shivae_amd import _native as N
attrs, col2attr = sc._device_tables(data.device)
xo = torch.empty(t, n, X, device="cuda"); mo = torch.empty(t, n, X, dtype=torch.uint8, device="cuda")
mu8 = mask.view(torch.uint8)
ms = timeit(lambda: N.check(N.lib().shivae_scale_transform(N.ptr(data), N.ptr(mu8), n, t, D, X, N.ptr(attrs), N.ptr(col2attr),
                                                           N.ptr(xo), N.ptr(mo), 3, N.stream())))
by = n * t * (5.0 * D + 5.0 * X)  # f32 data + u8 mask in, f32 x + u8 mask out
print("scale_transform %s n=%d t=%d D=%d X=%d  %.3f ms  %.1f GB/s (algorithmic %.2f GB)" % (name, n, t, D, X, ms, by / ms / 1e6, by / 1e9))
x, m = sc.transform_to_device(data, mask)
do = torch.empty(n, t, D, device="cuda"); dm = torch.empty(n, t, D, dtype=torch.uint8, device="cuda")
ms = timeit(lambda: N.check(N.lib().shivae_scale_inverse(N.ptr(x), N.ptr(m.view(torch.uint8)), n, t, D, X, N.ptr(attrs), N.ptr(do),
                                                         N.ptr(dm), 1, N.stream())))
print("scale_inverse   %s  %.3f ms  %.1f GB/s" % (name, ms, by / ms / 1e6))
