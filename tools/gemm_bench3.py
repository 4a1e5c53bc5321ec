"""L0 forward shape (wide N, short K): python tools/gemm_bench3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run as _run
def run(*a, **k): _run(*a, iters=10, warm=10, **k)
R = 16640
run(0, R, 32768, 528, 528, 528, 32768 + 64, tag="NT N=32768")
run(0, R, 32768, 528, 528, 528, 32768 + 64, epi=3, tag="NT N=32768 bias+relu")
run(0, R, 131072, 528, 528, 528, 131072 + 64, tag="NT N=131072")
run(0, R, 131072, 528, 528, 528, 131072 + 64, epi=3, tag="NT N=131072 bias+relu")
run(0, 8960, 131072, 528, 528, 528, 131072 + 64, epi=3, tag="NT N=131072 bias+relu M=8960")
