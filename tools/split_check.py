"""Accuracy and speed of the split-bf16 GEMM mode against the fp32-input MFMA mode (both against fp64):
   python tools/split_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shivae_amd import _native as N

L = N.lib()
torch.manual_seed(0)


def run(layout, M, Nn, K, split, reps=5, scale=1.0):
    A = torch.randn(M, K, device="cuda") * scale
    B = torch.randn(Nn, K, device="cuda")
    ref = A.double() @ B.double().T
    if layout == 0:
        a, b = A, B
    elif layout == 1:
        a, b = A, B.T.contiguous()
    else:
        a, b = A.T.contiguous(), B.T.contiguous()
    C = torch.zeros(M, Nn, device="cuda")
    L.shivae_set_gemm_split(split)
    def go():
        N.check(L.shivae_gemm(layout, M, Nn, K, N.ptr(a), a.shape[1], N.ptr(b), b.shape[1], N.ptr(C), Nn, None, None, 0, 0, 1, N.stream()))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    err = (C.double() - ref).abs()
    # error relative to the natural scale of one output: sqrt(K) * |a| * |b|
    nat = (A.double().abs() @ B.double().abs().T)
    return ms, float(err.max() / ref.abs().max()), float((err / nat).max()), 2.0 * M * Nn * K / ms / 1e9


for layout, name in ((0, "NT"), (1, "NN"), (2, "TN")):
    for (M, Nn, K) in ((32768, 256, 512), (32768, 512, 528), (512, 528, 32768), (8192, 256, 256), (130, 132, 100)):
        if layout == 2 and M > 1024:
            M2, N2, K2 = Nn, 512 if Nn == 256 else Nn, M     # weight-gradient shapes: long K
        else:
            M2, N2, K2 = M, Nn, K
        r0 = run(layout, M2, N2, K2, 0)
        r1 = run(layout, M2, N2, K2, 1)
        print("%s M=%6d N=%4d K=%6d | fp32 MFMA %7.3f ms %6.1f TF/s err/max %.2e err/nat %.2e | split-bf16 %7.3f ms %6.1f TF/s err/max %.2e err/nat %.2e"
              % (name, M2, N2, K2, r0[0], r0[3], r0[1], r0[2], r1[0], r1[3], r1[1], r1[2]), flush=True)
L.shivae_set_gemm_split(0)
