"""Train the same small model twice on the same batch and the same noise -- once through the captured HIP-graph step,
once eagerly -- and print both loss trajectories and the parameter difference (where do they part?):
   python tools/graph_vs_eager.py [cfg1|cfg2] [steps] [device]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shivae_amd
from shivae_amd.synth import synth_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
device_noise = len(sys.argv) > 3 and sys.argv[3] == "device"   # the scans' own noise streams instead of shared tensors
nr, npos, nb, cats, T, B, H, Z, K, miss = {"cfg1": (1, 1, 1, [3], 100, 64, 10, 2, 3, 0.3),
                                            "cfg2": (35, 0, 0, [], 48, 64, 10, 35, 10, 0.3)}[cfg]
tl = shivae_amd.utils.make_types_list(nr, npos, nb, cats)
x, mask = synth_batch(tl, T, B, miss=miss, seed=1234, device="cuda", chunk=B, seq0=0)
models, opts = [], []
for graphs in (True, False):
    torch.manual_seed(0)
    m = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
    m.use_graphs = graphs
    m.check_nan = False
    m.train()
    models.append(m)
    opts.append(shivae_amd.FusedClipAdam(m, lr=5e-3, clip=0.5))
gen = torch.Generator(device="cuda")
gen.manual_seed(7)
done_probe = False
for it in range(steps):
    u = torch.rand(T, B, K, device="cuda", generator=gen)
    eps = torch.randn(T, B, Z, device="cuda", generator=gen)
    vals = []
    for m, o in zip(models, opts):
        before = m.flat_params.clone()
        o.zero_grad()
        if not device_noise:
            m.set_noise(u, eps)
        loss = m(x, mask, beta=1.0, temp=3.0)
        loss["n_elbo"].backward()
        gn = float(m.flat_grads.norm())
        v0 = float(loss["n_elbo"].detach())
        if m is models[0] and (v0 != v0 or abs(v0) == float("inf")) and not done_probe:
            done_probe = True
            lu, le = m._last.get("u"), m._last.get("eps")
            print("  first non-finite captured step %d: %s" % (it, {k: float(v.detach()) for k, v in loss.items()}))
            print("  params before finite %s max|p| %.3e; u finite %s [%.3e, %.9f]; eps finite %s [%.4f, %.4f]"
                  % (bool(torch.isfinite(before).all()), float(before.abs().max()), bool(torch.isfinite(lu).all()), float(lu.min()),
                     float(lu.max()), bool(torch.isfinite(le).all()), float(le.min()), float(le.max())))
            probe = shivae_amd.ShiVAE(H, Z, K, 1, types_list=tl, learn_std=True)
            probe.use_graphs = False
            probe.check_nan = False
            probe.train()
            probe.flat_params.copy_(before)
            probe.set_noise(lu, le)
            l2 = probe(x, mask, beta=1.0, temp=3.0)
            print("  eager step, same parameters and noise tensors: %s" % {k: float(v.detach()) for k, v in l2.items()})
        o.step()
        vals.append((v0, gn))
    dp = float((models[0].flat_params - models[1].flat_params).abs().max())
    if it < 12 or it % 10 == 0 or vals[0][0] != vals[0][0] or abs(vals[0][0]) == float("inf"):
        print("step %3d  graph: n_elbo %.5f |g| %.4e   eager: n_elbo %.5f |g| %.4e   max|dparam| %.3e  graphs cached %d"
              % (it, vals[0][0], vals[0][1], vals[1][0], vals[1][1], dp, len(models[0]._graphs)), flush=True)
