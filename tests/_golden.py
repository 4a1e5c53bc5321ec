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
