"""Timing of the pre-split-weight NT kernel alone (shivae_gemm_nt_presplit):  python tools/sw_bench.py
SHIVAE_SW_DBG (diagnostic builds only): 1 no weight DMA, 2 no activation path, 4 no MFMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shivae_amd import _native as N
L = N.lib()
torch.manual_seed(0)
for (M, Nn, K) in ((16640 * 16, 256, 256), (16640 * 16, 256, 512), (16640 * 8, 512, 528), (16640, 512, 131072 // 4)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(Nn, K, device="cuda")
    img = torch.empty(L.shivae_presplit_bytes(Nn, K), dtype=torch.uint8, device="cuda")
    C = torch.empty(M, Nn, device="cuda")
    N.check(L.shivae_presplit_weights(N.ptr(W), Nn, K, K, N.ptr(img), N.stream()))
    def go():
        N.check(L.shivae_gemm_nt_presplit(M, Nn, K, N.ptr(A), K, N.ptr(img), N.ptr(C), Nn, None, None, 0, 0, 1, N.stream()))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rounds = (M / 128) * (Nn / 256) / 256   # tiles per CU
    print("M=%7d N=%4d K=%6d  %8.3f ms  %6.1f TFLOP/s (fp32-equivalent)  %6.2f us per tile and CU" %
          (M, Nn, K, ms, 2.0 * M * Nn * K / ms / 1e9, 1e3 * ms / rounds), flush=True)
