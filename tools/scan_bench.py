"""Time shivae_scan_fwd / shivae_scan_bwd alone (the recurrent chain of cfg3: H=256, Z=64, K=16):
   python tools/scan_bench.py [B] [T] [reps]      (SHIVAE_SCAN_WAVES=8|16 selects the workgroup geometry)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shivae_amd
from shivae_amd import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, Z, K = 256, 64, 16
tl = shivae_amd.utils.make_types_list(4, 4, 4, [4] * 4)   # (a small decoder: only the chain is timed)
torch.manual_seed(0)
m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
m.zero_grad()
plan = m._activate(T, B)
X = m.x_dim
x = torch.randn(T, B, X, device="cuda")
u, eps = torch.rand(T, B, K, device="cuda"), torch.randn(T, B, Z, device="cuda")
d_emb = torch.randn(T, B, 2 * H + K, device="cuda") * 1e-3
dmu, dsd = torch.randn(T, B, Z, device="cuda") * 1e-3, torch.randn(T, B, Z, device="cuda") * 1e-3
dlg = torch.randn(T, B, K, device="cuda") * 1e-3
L = N.lib()
# (the reverse-time chain multiplies by the transposed matrices of the forward chain: the same MFMA work; the weight
# gradients, the other half of "backward = 2 x forward", are batched TN GEMMs outside the scan)
flops = 2.0 * (H * H + H * K + (H + K) * H + H * H + 2.0 * H * Z + Z * H + 8.0 * H * H) * T * B


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        N.profile_start()
        fn()
        prof = N.profile_stop()
        for p in prof:
            if p["id"] in (18, 19):
                best = min(best, p["ms"] / p["launches"])
    return best


f = timed(lambda: N.check(L.shivae_scan_fwd(plan.h, N.ptr(x), N.ptr(u), N.ptr(eps), 3.0, N.stream())))
b = timed(lambda: N.check(L.shivae_scan_bwd(plan.h, N.ptr(x), N.ptr(u), N.ptr(eps), 3.0, N.ptr(d_emb), N.ptr(dmu), N.ptr(dsd),
                                            N.ptr(dlg), N.stream())))
print("B=%d T=%d waves=%s: scan_fwd %.2f ms = %.1f TFLOP/s (%.3f of 157.3) | scan_bwd %.2f ms = %.1f TFLOP/s (%.3f)" %
      (B, T, os.environ.get("SHIVAE_SCAN_WAVES", "auto"), f, flops / f / 1e9, flops / f / 1e9 / 157.3, b, flops / b / 1e9,
       flops / b / 1e9 / 157.3), flush=True)
