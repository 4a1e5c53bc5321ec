"""One GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/gemm_one.py layout M N K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shivae_amd import _native as N
L = N.lib()
layout, M, Nn, K = (int(v) for v in sys.argv[1:5])
rowsA, colsA = (M, K) if layout != 2 else (K, M)
rowsB, colsB = (Nn, K) if layout == 0 else (K, Nn)
A = torch.randn(rowsA, colsA, device="cuda"); B = torch.randn(rowsB, colsB, device="cuda"); C = torch.zeros(M, Nn, device="cuda")
for _ in range(3):
    N.check(L.shivae_gemm(layout, M, Nn, K, N.ptr(A), colsA, N.ptr(B), colsB, N.ptr(C), Nn, None, None, 0, 0, 1, N.stream()))
torch.cuda.synchronize()
