"""Every captured training step of a small model re-computed eagerly on the same parameters and noise tensors: prints
the steps whose loss or gradient differ by more than fp32 summation order allows.
   python tools/graph_check_steps.py [cfg1|cfg2] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shivae_amd
from shivae_amd.synth import synth_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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
nbad = 0
for it in range(steps):
    before = m.flat_params.clone()
    opt.zero_grad()
    loss = m(x, mask, beta=1.0, temp=3.0)
    loss["n_elbo"].backward()
    g = m.flat_grads.clone()
    u, eps = m._last.get("u"), m._last.get("eps")
    v = float(loss["n_elbo"].detach())
    if u is not None:
        ref.flat_params.copy_(before)
        ref.zero_grad()
        ref.set_noise(u, eps)
        l2 = ref(x, mask, beta=1.0, temp=3.0)
        l2["n_elbo"].backward()
        v2 = float(l2["n_elbo"].detach())
        gerr = float((g - ref.flat_grads).abs().max()) / max(float(ref.flat_grads.abs().max()), 1e-30)
        lerr = abs(v - v2) / max(abs(v2), 1.0)
        if not (gerr < 1e-3 and lerr < 1e-4):
            nbad += 1
            i = int((g - ref.flat_grads).abs().argmax())
            print("step %d (graphs %d): captured n_elbo %.6f eager %.6f | grad err %.3e at flat index %d (captured %.4e, eager %.4e), "
                  "non-finite captured grads %d" % (it, len(m._graphs), v, v2, gerr, i, float(g[i]), float(ref.flat_grads[i]),
                                                    int((~torch.isfinite(g)).sum())), flush=True)
            if nbad == 1:
                for name, prm in m.named_parameters():
                    off = prm.data_ptr() - m.flat_params.data_ptr()
                    off //= 4
                    n = prm.numel()
                    e = float((g[off:off + n] - ref.flat_grads[off:off + n]).abs().max())
                    sc = float(ref.flat_grads[off:off + n].abs().max())
                    if e > 1e-3 * max(sc, 1e-6):
                        print("    %-28s max err %.3e (max|eager grad| %.3e)" % (name, e, sc))
                # which records of the step differ (the workspace still holds both steps' records)
                from shivae_amd import _native as N
                L = N.lib()
                pm, pr = m._plans[(T, B)], ref._plans[(T, B)]
                for w, nm in enumerate(N._O_NAMES):
                    if nm in ("DEC_PARAMS",):
                        continue
                    outs = []
                    for mod, pl in ((m, pm), (ref, pr)):
                        mod._activate(T, B)
                        o = torch.empty(T, B, pl.output_width(w), device="cuda")
                        N.check(L.shivae_get_output(pl.h, w, N.ptr(o), N.stream()))
                        outs.append(o)
                    d = (outs[0] - outs[1]).abs()
                    d = torch.where(torch.isfinite(d), d, torch.full_like(d, 1e38))
                    if float(d.max()) > 1e-3 * max(float(outs[1].abs().max()), 1e-6):
                        badpos = (d > 1e-3 * max(float(outs[1].abs().max()), 1e-6)).nonzero()
                        ts = sorted(set(int(v) for v in badpos[:, 0].tolist()))
                        bs = sorted(set(int(v) for v in badpos[:, 1].tolist()))
                        cs = sorted(set(int(v) for v in badpos[:, 2].tolist()))
                        print("    record %-14s max diff %.3e (max|eager| %.3e): %d bad entries, t in [%d..%d] (%d distinct), b %s, cols %s"
                              % (nm, float(d.max()), float(outs[1].abs().max()), badpos.shape[0], ts[0], ts[-1], len(ts),
                                 bs[:8] + (["..."] if len(bs) > 8 else []), cs[:8] + (["..."] if len(cs) > 8 else [])))
                # the captured step once more on the same parameters and noise: same garbage, or not?
                keep = m.flat_params.clone()
                m.flat_params.copy_(before)
                for rep in range(3):
                    m.zero_grad()
                    m.set_noise(u, eps)
                    l3 = m(x, mask, beta=1.0, temp=3.0)
                    l3["n_elbo"].backward()
                    print("    captured again (%d): grad err %.3e" % (rep, float((m.flat_grads - ref.flat_grads).abs().max()) /
                                                                   max(float(ref.flat_grads.abs().max()), 1e-30)))
                m.flat_params.copy_(keep)
                m.zero_grad()
                m.flat_grads.copy_(g)
            if nbad >= 3:
                break
    opt.step()
print("checked %d steps, %d bad" % (it + 1, nbad))
