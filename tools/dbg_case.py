"""Debug: per-parameter gradient error of one oracle case of tests/test_gpu_step.py: python tools/dbg_case.py NAME"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import shivae_amd
import test_gpu_step as tg
so = tg.so
name = sys.argv[1]
seeds = [int(v) for v in sys.argv[2:]] or [12345]
for seed in seeds:
    print("seed", seed)
    tl, T, B, H, Z, K, learn_std, miss, chunk = tg.CASES[name]
    rng = np.random.default_rng(seed)
    P = so.init_params(tl, H, Z, K, learn_std, seed=1)
    for k in P:
        if "bias" in k:
            P[k] = (rng.standard_normal(P[k].shape) * 0.1).astype(np.float32)
    x, mask = tg._synth(rng, tl, T, B, miss)
    u = rng.random((T, B, K)).astype(np.float32)
    eps = rng.standard_normal((T, B, Z)).astype(np.float32)
    om = so.OracleShiVAE(P, tl, H, Z, K, learn_std)
    c = om.forward(x, mask, u, eps, 0.8, 1.5)
    G = om.backward()
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=learn_std)
    m.chunk_rows = chunk
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.train(); m.set_noise(u, eps)
    loss = m(torch.from_numpy(x), torch.from_numpy(mask), beta=0.8, temp=1.5)
    print({k: (float(loss[k]), float(c["losses"][k])) for k in shivae_amd.LOSS_KEYS})
    loss["n_elbo"].backward()
    print("observed per attribute:", [(int((~mask[..., i]).sum()) if mask.dtype == bool else int((mask[..., i] == 0).sum())) for i in range(mask.shape[-1])])
    for k, p in m.named_parameters():
        err = np.abs(p.grad.cpu().numpy() - G[k]).max() / max(np.abs(G[k]).max(), 1e-3)
        if err > 1e-4: print("%-50s %.3e" % (k, err))
