import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
R = 16384
run(1, R, 512, 8192, 8192, 512, 512, tag="NN K=8192 lda=8192")
run(1, R, 512, 8192, 131136, 512, 512, tag="NN K=8192 lda=131136")
run(1, R, 512, 8192, 8192, 528, 512, tag="NN K=8192 ldb=528")
run(0, R, 512, 8192, 8192, 8192, 512, tag="NT K=8192")
run(0, R, 512, 8192, 131136, 8192, 512, tag="NT K=8192 lda=131136")
run(2, 512, 512, R, 512, 512, 512, epi=32, splitk=16, tag="TN K=16384 splitk16")
run(1, R, 2048, 2048, 2048, 2048, 2048, tag="NN 16384x2048x2048")
run(0, R, 2048, 2048, 2048, 2048, 2048, tag="NT 16384x2048x2048")
