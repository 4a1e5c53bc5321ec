"""Isolated timing of the fp32 MFMA GEMM through the C ABI (shivae_gemm): python tools/gemm_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shivae_amd import _native as N

L = N.lib()

def run(layout, M, Nn, K, lda, ldb, ldc, epi=0, splitk=1, iters=5, tag="", warm=1):
    rowsA = M if layout != 2 else K
    rowsB = Nn if layout == 0 else K
    A = torch.randn(rowsA if lda else 1, lda if lda else (K if layout != 2 else M), device="cuda")
    B = torch.randn(rowsB, ldb, device="cuda")
    C = torch.zeros(M if ldc else 1, ldc if ldc else Nn, device="cuda")
    bias = torch.randn(Nn, device="cuda") if epi & 1 else None
    def go():
        N.check(L.shivae_gemm(layout, M, Nn, K, N.ptr(A), lda, N.ptr(B), ldb, N.ptr(C), ldc, N.ptr(bias) if epi & 1 else None, None, 0, epi, splitk, N.stream()))
    for _ in range(warm): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-34s layout=%d M=%6d N=%6d K=%6d lda=%7d ldb=%7d  %8.3f ms  %6.1f TF" % (tag, layout, M, Nn, K, lda, ldb, ms, 2.0*M*Nn*K/ms/1e9), flush=True)

def main():
    R = 16384
    run(0, R, 512, 528, 528, 528, 512, tag="NT contiguous")
    run(0, R, 4096, 528, 528, 528, 4096, tag="NT wide N")
    run(0, R, 256, 512, 512, 512, 256, tag="NT L1 contiguous")
    run(0, R, 256, 512, 131072, 512, 65536, tag="NT L1 lda=131072 (pow2)")
    run(0, R, 256, 512, 131072 + 64, 512, 65536 + 64, tag="NT L1 lda padded")
    run(0, 4096, 4096, 4096, 4096, 4096, 4096, tag="NT 4096^3")
    run(1, R, 512, 256, 256, 512, 512, tag="NN dL1 contiguous")
    run(1, R, 512, 256, 65536, 512, 131072, tag="NN dL1 pow2 ld")
    run(1, R, 512, 256, 65536 + 64, 512, 131072 + 64, tag="NN dL1 padded ld")
    run(1, R, 528, 8192, 8192, 528, 528, tag="NN dL0 K=8192")
    run(1, 4096, 4096, 4096, 4096, 4096, 4096, tag="NN 4096^3")
    run(2, 512, 528, R, 512, 528, 528, epi=32, splitk=8, tag="TN wgrad splitk8")
    run(2, 256, 512, R, 65536, 131072, 512, epi=32, splitk=8, tag="TN wL1 pow2 ld")
    run(2, 256, 512, R, 65536 + 64, 131072 + 64, 512, epi=32, splitk=8, tag="TN wL1 padded ld")
    run(2, 4096, 4096, 4096, 4096, 4096, 4096, tag="TN 4096^3")

def long_k():
    """long-K and wide-ld shapes (formerly gemm_bench2.py)"""
    R = 16384
    run(1, R, 512, 8192, 8192, 512, 512, tag="NN K=8192 lda=8192")
    run(1, R, 512, 8192, 131136, 512, 512, tag="NN K=8192 lda=131136")
    run(1, R, 512, 8192, 8192, 528, 512, tag="NN K=8192 ldb=528")
    run(0, R, 512, 8192, 8192, 8192, 512, tag="NT K=8192")
    run(0, R, 512, 8192, 131136, 8192, 512, tag="NT K=8192 lda=131136")
    run(2, 512, 512, R, 512, 512, 512, epi=32, splitk=16, tag="TN K=16384 splitk16")
    run(1, R, 2048, 2048, 2048, 2048, 2048, tag="NN 16384x2048x2048")
    run(0, R, 2048, 2048, 2048, 2048, 2048, tag="NT 16384x2048x2048")


def layer0():
    """layer-0 forward shapes: wide N, short K (formerly gemm_bench3.py)"""
    R = 16640
    for n in (32768, 131072):
        run(0, R, n, 528, 528, 528, n + 64, tag="NT N=%d" % n)
        run(0, R, n, 528, 528, 528, n + 64, epi=3, tag="NT N=%d bias+relu" % n)


def one(layout, M, Nn, K):
    """one shape, three launches, for rocprofv3 --pmc passes (formerly gemm_one.py)"""
    run(layout, M, Nn, K, K if layout != 2 else M, K if layout == 0 else Nn, Nn, tag="one", iters=3, warm=0)


def decoder():
    """the decoder's K = 256 / 512 shapes at a chunk's worth of rows: NT (weights [N, K]) against NN (weights [K, N])"""
    R = 1 << 20
    for (n, k) in ((256, 256), (512, 256), (256, 512), (512, 528)):
        run(0, R, n, k, k, k, n, tag="NT N=%d K=%d" % (n, k), iters=3)
        run(1, R, n, k, k, n, n, tag="NN N=%d K=%d" % (n, k), iters=3)
        run(0, R, n, k, k, k, n, epi=1 | 2, tag="NT bias+relu", iters=3)
        run(1, R, n, k, k, n, n, epi=1 | 2, tag="NN bias+relu", iters=3)


def nomem():
    """where the time of a short-K NT GEMM goes: the same launch with the A rows and / or the C rows all at one address
    (ld = 0: the operand stays in cache, the MFMA work and the instruction stream are unchanged)"""
    R = 1 << 20
    for (n, k) in ((256, 256), (512, 256), (256, 512)):
        for (la, lc, tag) in ((k, n, "A, C in HBM"), (0, n, "A cached"), (k, 0, "C cached"), (0, 0, "A, C cached")):
            run(0, R, n, k, la, k, lc, tag="NT N=%d K=%d %s" % (n, k, tag), iters=3)


def mscale():
    """is the short-K shortfall per launch or per tile?  the K = 256 shape at 2^18 .. 2^22 rows"""
    for sh in (18, 20, 22):
        run(0, 1 << sh, 256, 256, 256, 256, 256, epi=3, tag="NT N=256 K=256 M=2^%d" % sh, iters=3, warm=2)
        run(0, 1 << sh, 256, 512, 512, 512, 256, epi=3, tag="NT N=256 K=512 M=2^%d" % sh, iters=3, warm=2)


def epiparts():
    """vector (LDS-staged, 16-B stores) against scalar (registers -> 4-B stores, 128-B segments) epilogue: ldc = N + 1 takes the latter"""
    for k in (256, 512):
        run(0, 1 << 22, 256, k, k, k, 256, epi=3, tag="NT N=256 K=%d staged epilogue" % k, iters=3, warm=2)
        run(0, 1 << 22, 256, k, k, k, 257, epi=3, tag="NT N=256 K=%d direct epilogue" % k, iters=3, warm=2)


if __name__ == "__main__":
    import sys
    what = sys.argv[1] if len(sys.argv) > 1 else "main"
    if what == "one":
        one(*(int(v) for v in sys.argv[2:6]))
    else:
        {"main": main, "long_k": long_k, "layer0": layer0, "decoder": decoder, "nomem": nomem, "mscale": mscale, "epiparts": epiparts}[what]()
