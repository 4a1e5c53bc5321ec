"""Loader for tests/golden/*.npz (made by tests/golden/make_golden.py from the reference)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
               if not os.path.basename(p).startswith(("scaler_", "metrics_", "gen_")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    T, B, H, Z, K = (int(v) for v in g["dims"])
    types_list = []
    for row in g["types_csv"]:
        n, ty, dim, nc, idx = str(row).split(",")
        types_list.append(dict(name=n, type=ty, dim=dim, nclass=nc, index=idx))
    cfg = dict(T=T, B=B, H=H, Z=Z, K=K, learn_std=bool(g["learn_std"]), beta=float(g["beta"]),
               temp=float(g["temp"]), types_list=types_list)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    return g, cfg, sd


def golden_dec_params(g, types_list, prefix="dec"):
    out = []
    for i, t in enumerate(types_list):
        if t["type"] in ("real", "pos"):
            out.append((g["%s/%d/mean" % (prefix, i)], g.get("%s/%d/std" % (prefix, i))))
        elif t["type"] == "bin":
            out.append(g["%s/%d/logits" % (prefix, i)])
        else:
            out.append(g["%s/%d/theta" % (prefix, i)])
    return out


# ---- the two gradient criteria of the step tests (stated fp32 tolerances) ----
GRAD_RTOL = 5e-4     # max|got - ref| relative to the largest |ref| entry of the tensor
GRAD_L2_RTOL = 1e-3  # ||got - ref||_2 relative to ||ref||_2 of the tensor


def grad_errors(got, ref):
    """(max-norm error, L2 error) of one gradient tensor against its reference.  The max-norm figure alone lets rows
    of small magnitude (the bias of an attribute observed 10 % of the time next to one observed always) be wrong by
    100 %; the L2 figure is per tensor -- per attribute for the decoder's parameters -- and has no such blind spot.
    Floors: tensors whose reference gradient is exactly zero (never-observed attributes) must come out ~zero."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    d = got - ref
    e_max = np.abs(d).max() / max(np.abs(ref).max(), 1e-3)
    e_l2 = np.sqrt((d * d).sum()) / max(np.sqrt((ref * ref).sum()), 1e-3 * np.sqrt(ref.size))
    return float(e_max), float(e_l2)


def bad_grads(named_grads, G, rtol=GRAD_RTOL, l2_rtol=GRAD_L2_RTOL):
    """[(name, max-norm error, L2 error)] of the tensors that miss either criterion, worst first."""
    bad = []
    for k, g in named_grads:
        e_max, e_l2 = grad_errors(g, G[k])
        if not (e_max < rtol and e_l2 < l2_rtol):
            bad.append((k, e_max, e_l2))
    return sorted(bad, key=lambda t: -max(t[1] / rtol, t[2] / l2_rtol))
