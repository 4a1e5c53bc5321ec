"""Where a 128x128 GEMM workgroup spends its time (in-kernel s_memrealtime stamps, 10 ns; s_memtime stops while a CU idles): python tools/gemm_stamps.py M N K [epi]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shivae_amd import _native as N

L = N.lib()
M, Nn, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn(M, K, device="cuda"); B = torch.randn(Nn, K, device="cuda"); C = torch.zeros(M, Nn, device="cuda")
bias = torch.randn(Nn, device="cuda"); aux = torch.randn(M, Nn, device="cuda")
nblk = (-(-M // 128)) * (-(-Nn // 128)) * 2 + 4096
st = torch.zeros(4 * nblk, dtype=torch.int64, device="cuda")


def go():
    N.check(L.shivae_gemm(0, M, Nn, K, N.ptr(A), K, N.ptr(B), K, N.ptr(C), Nn, N.ptr(bias), N.ptr(aux), Nn, epi, 1, N.stream()))

for _ in range(20): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
N.check(L.shivae_debug_gemm_stamps(N.ptr(st)))
e0.record(); go(); e1.record(); torch.cuda.synchronize()
ms_stamped = e0.elapsed_time(e1)
N.check(L.shivae_debug_gemm_stamps(None))
s = st.cpu().numpy().reshape(-1, 4)
s = s[s[:, 3] > 0]
pro, main, ep = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
tot = s[:, 3] - s[:, 0]
print("M=%d N=%d K=%d epi=%d: %.3f ms, %.1f TF; %d workgroups stamped" % (M, Nn, K, epi, ms, 2.0 * M * Nn * K / ms / 1e9, len(s)))
for nm, v in (("prologue", pro), ("main loop", main), ("epilogue", ep), ("total", tot)):
    print("  %-10s median %7.0f  p10 %7.0f  p90 %7.0f  (x100 MHz ticks)" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
print("  stamped launch %.3f ms; sum(total) / launch time = %.0f workgroups in flight on average (s_memrealtime: 100 MHz, per-XCD bases differ, so no global span)" % (ms_stamped, tot.sum() * 1e-5 / ms_stamped))
