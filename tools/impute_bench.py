"""Imputation forward (reconstruction at temp 0.01 + 100-sample Monte-Carlo means) on the cfg3 model:
python tools/impute_bench.py [B] [T]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shivae_amd
from shivae_amd.synth import synth_batch
from bench import CONFIGS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nr, npos, nb, cats, _, _, H, Z, K, miss = CONFIGS["cfg3"]
tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
torch.manual_seed(0)
m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True).cuda()
m.eval()
x, mask = synth_batch(tl, T, B, miss=miss, seed=7, device="cuda")
for label, fn in (("reconstruction (forward, all entries)", lambda: m.reconstruction(x, temp=0.01)),):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-40s B=%d T=%d  %8.1f ms  %10.0f sequence-timesteps/s" % (label, B, T, dt * 1e3, B * T / dt), flush=True)
(likes, x_hat), _ = out
t0 = time.perf_counter()
for _ in range(3): xs = m.generate_heter_samples(likes, n_samples=100, seed=1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("%-40s B=%d T=%d  %8.1f ms  %10.0f sequence-timesteps/s" % ("generate_heter_samples (100 draws)", B, T, dt * 1e3, B * T / dt), flush=True)
