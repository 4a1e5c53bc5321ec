"""Golden vectors for the scaler pre/post pass, made by calling the REFERENCE's scaler classes
(/root/reference/src/lib/scalers.py) in the build container:  python tests/golden/make_golden_scaler.py

What runs under the installed numpy 2 / scikit-learn 1.7: StandardScaler, PositiveScaler (all methods) and
HeterogeneousScaler.transform / inverse_transform for real+pos+bin attributes once the per-attribute moments
are set (HeterogeneousScaler.fit itself uses np.NaN, removed in numpy 2, so the script fits the per-attribute
scalers directly, which is what fit() does, scalers.py:79-83).  CategoricalScaler cannot be constructed
(OneHotEncoder(sparse=False)): no categorical golden exists -- see oracle/scaler_oracle.py."""
import os, sys, types
import numpy as np

REF = "/root/reference/src"
sys.path.insert(0, REF)
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
from lib import scalers as rs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)
N, T = 6, 9
tl = [dict(name="r0", type="real", dim="1", nclass="1", index="0"), dict(name="p0", type="pos", dim="1", nclass="1", index="1"),
      dict(name="b0", type="bin", dim="1", nclass="2", index="2"), dict(name="r1", type="real", dim="1", nclass="1", index="3"),
      dict(name="p1", type="pos", dim="1", nclass="1", index="4")]
data = np.stack([rng.normal(3.0, 2.0, (N, T)), np.exp(rng.normal(0.5, 0.8, (N, T))), (rng.random((N, T)) < 0.4).astype(float),
                 rng.normal(-10.0, 0.1, (N, T)), np.exp(rng.normal(2.0, 0.3, (N, T)))], axis=-1)
mask = rng.random((N, T, len(tl))) < 0.3
hs = rs.HeterogeneousScaler(tl)
flat = np.where(mask, np.nan, data).reshape(-1, len(tl))
for i, sc in enumerate(hs.scalers):
    if sc is not None:
        sc.fit(flat[:, i].copy())
moments = np.array([[sc.mean, sc.std] if sc is not None else [0.0, 1.0] for sc in hs.scalers])
data_nan = np.where(mask, np.nan, data)          # the datasets hand NaN at missing entries
x, xm = hs.transform(data_nan.copy(), mask.copy())
xT = np.where(xm, 0.0, x).transpose(1, 0, 2)     # what preprocess_batch + zero_fill feed the model
inv, invm = hs.inverse_transform(xT.copy(), xm.transpose(1, 0, 2).copy(), transpose=True)
std_hat = np.abs(rng.normal(0.5, 0.2, (T, N, 1)))
std_inv = np.stack([hs.scalers[0].std_inverse_transform(std_hat), hs.scalers[1].std_inverse_transform(std_hat)])
np.savez(os.path.join(OUT, "scaler_mixed.npz"), types=np.array([t["type"] for t in tl]), data=data_nan, mask=mask,
         moments=moments, x=x, xmask=xm, xT=xT, inv=inv, invmask=invm, std_hat=std_hat, std_inv=std_inv)
print("wrote scaler_mixed.npz", x.shape, inv.shape, moments)
