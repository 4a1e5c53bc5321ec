"""Golden vectors for the imputation metrics, made by calling the REFERENCE's lib/loss.py in the build container:
    python tests/golden/make_golden_metrics.py
compute_avg_error runs there for bin / cat attributes and for the burst cross-correlation; its real / pos branch
calls sklearn's mean_squared_error(squared=False) (loss.py:145,156), which the installed scikit-learn rejects, so
for those two types the script evaluates the same expression with np.sqrt(mean_squared_error(...))."""
import os, sys, types
import numpy as np

REF = "/root/reference/src"
sys.path.insert(0, REF)
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
from lib import loss as rl, utils as ru  # noqa: E402
from sklearn.metrics import mean_squared_error  # noqa: E402

rng = np.random.default_rng(21)
N, T = 7, 40
tl = [dict(name="r", type="real", dim="1", nclass="1", index="0"), dict(name="p", type="pos", dim="1", nclass="1", index="1"),
      dict(name="b", type="bin", dim="1", nclass="2", index="2"), dict(name="c", type="cat", dim="4", nclass="4", index="3")]
x = np.stack([rng.normal(0, 2, (N, T)), np.exp(rng.normal(1, 1, (N, T))), (rng.random((N, T)) < 0.5).astype(float),
              rng.integers(0, 4, (N, T)).astype(float)], -1)
xh = x.copy()
xh[..., 0] += rng.normal(0, 0.5, (N, T)); xh[..., 1] *= np.exp(rng.normal(0, 0.2, (N, T)))
flip = rng.random((N, T)) < 0.2; xh[..., 2] = np.where(flip, 1 - x[..., 2], x[..., 2]) * 0.9 + 0.04
sw = rng.random((N, T)) < 0.3; xh[..., 3] = np.where(sw, rng.integers(0, 4, (N, T)), x[..., 3])
mask = np.zeros((N, T, 4), dtype=int)
for n in range(N):
    for d in range(4):
        for _ in range(3):
            s = rng.integers(0, T - 8); mask[n, s:s + rng.integers(2, 8), d] = 1
em, cm = {}, {}
for t in tl:
    one = [dict(t, index=t["index"])]
    if t["type"] in ("bin", "cat"):
        _, e, _, c = rl.compute_avg_error(x.copy(), xh.copy(), mask, one, plot_imp=False)
        em[t["name"]], cm[t["name"]] = e[t["name"]], c[t["name"]]
    else:
        i = int(t["index"]); xt, ht, mt = x[:, :, i:i + 1].copy(), xh[:, :, i:i + 1].copy(), mask[:, :, i:i + 1]
        if t["type"] == "pos":
            xt[xt < 0] = 0; ht[ht < 0] = 0
        norm = xt.max() - xt.min()
        em[t["name"]] = np.sqrt(mean_squared_error(xt[mt == 1], ht[mt == 1])) / norm
        cm[t["name"]] = rl.burst_cross_correlation(xt, ht, mt) / mt.sum()
bursts = ru.artmask2burst(mask[0, :, 0])
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_mixed.npz"), x=x, xh=xh, mask=mask,
         err=np.array([em[t["name"]] for t in tl]), corr=np.array([cm[t["name"]] for t in tl]), bursts=bursts,
         types=np.array([t["type"] for t in tl]), dims=np.array([int(t["dim"]) for t in tl]))
print(em, cm)
