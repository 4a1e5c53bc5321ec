"""Locate the first non-finite loss of a captured small-model training run with the scans' own noise streams:
which loss terms, what the noise tensors looked like, and what an eager step on the same parameters and tensors gives.
   python tools/graph_inf_probe.py [cfg1|cfg2] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shivae_amd
from shivae_amd.synth import synth_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nr, npos, nb, cats, T, B, H, Z, K, miss = {"cfg1": (1, 1, 1, [3], 100, 64, 10, 2, 3, 0.3),
                                            "cfg2": (35, 0, 0, [], 48, 64, 10, 35, 10, 0.3)}[cfg]
tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
x, mask = synth_batch(tl, T, B, miss=miss, seed=1234, device="cuda", chunk=B, seq0=0)
torch.manual_seed(0)
m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
m.check_nan = False
m.train()
opt = shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5)
ref = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
ref.use_graphs = False
ref.check_nan = False
ref.train()
for it in range(steps):
    before = m.flat_params.clone()
    opt.zero_grad()
    loss = m(x, mask, beta=1.0, temp=3.0)
    loss["n_elbo"].backward()
    vals = {k: float(v.detach()) for k, v in loss.items()}
    u, eps = m._last.get("u"), m._last.get("eps")
    bad = any(v != v or abs(v) == float("inf") for v in vals.values())
    if bad or it % 20 == 0:
        print("step %d graphs=%d losses=%s" % (it, len(m._graphs), {k: round(v, 3) for k, v in vals.items()}), flush=True)
    if bad:
        print("  params before the step finite: %s, max|p| %.3e" % (bool(torch.isfinite(before).all()), float(before.abs().max())))
        if u is not None:
            print("  u: finite %s min %.3e max %.9f | eps: finite %s min %.4f max %.4f"
                  % (bool(torch.isfinite(u).all()), float(u.min()), float(u.max()), bool(torch.isfinite(eps).all()),
                     float(eps.min()), float(eps.max())))
            ref.flat_params.copy_(before)
            ref.zero_grad()
            ref.set_noise(u, eps)
            l2 = ref(x, mask, beta=1.0, temp=3.0)
            print("  eager step on the same parameters and noise tensors: %s" % {k: round(float(v.detach()), 3) for k, v in l2.items()})
            # and the captured step once more on the same inputs
            m.flat_params.copy_(before)
            m.zero_grad()
            m.set_noise(u, eps)
            l3 = m(x, mask, beta=1.0, temp=3.0)
            print("  captured step again, same parameters and noise tensors: %s" % {k: round(float(v.detach()), 3) for k, v in l3.items()})
        break
    opt.step()
