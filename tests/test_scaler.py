"""Scaler pre/post pass: oracle vs the reference's golden vectors (CPU), HIP kernels vs
oracle and golden (GPU).  The categorical part has no reference golden (see oracle/scaler_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import scaler_oracle as so

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "scaler_mixed.npz"))
TL = [dict(name="a%d" % i, type=str(t), dim="1", nclass="1", index=str(i)) for i, t in enumerate(G["types"])]
MOM = [tuple(m) for m in G["moments"]]


def _mixed_case(seed=3, n=5, t=11):
    """All four types, cat with C=3 and C=5, interleaved; NaN at missing entries."""
    rng = np.random.default_rng(seed)
    tl, cols = [], []
    spec = [("real", 1), ("cat", 3), ("pos", 1), ("bin", 1), ("cat", 5), ("real", 1)]
    for i, (ty, c) in enumerate(spec):
        tl.append(dict(name="v%d" % i, type=ty, dim=str(c), nclass=str(max(c, 2 if ty == "bin" else 1)), index=str(i)))
        if ty == "real":
            cols.append(rng.normal(1.0, 3.0, (n, t)))
        elif ty == "pos":
            cols.append(np.exp(rng.normal(0.0, 1.0, (n, t))))
        elif ty == "bin":
            cols.append((rng.random((n, t)) < 0.5).astype(float))
        else:
            cols.append(rng.integers(0, c, (n, t)).astype(float))
    data = np.stack(cols, -1)
    mask = rng.random(data.shape) < 0.35
    return tl, np.where(mask, np.nan, data), mask


def test_oracle_fit_matches_reference_moments():
    mom = so.fit(TL, G["data"], G["mask"])
    np.testing.assert_allclose(np.array(mom), G["moments"], rtol=1e-12, atol=0)


def test_oracle_transform_matches_reference():
    x, m = so.transform(TL, MOM, G["data"], G["mask"])
    np.testing.assert_array_equal(m, G["xmask"])
    np.testing.assert_allclose(x, G["x"], rtol=1e-12, atol=0, equal_nan=True)


def test_oracle_inverse_matches_reference():
    d, m = so.inverse_transform(TL, MOM, G["xT"], G["xmask"].transpose(1, 0, 2), transpose=True)
    np.testing.assert_array_equal(m, G["invmask"])
    np.testing.assert_allclose(d, G["inv"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(so.std_inverse_transform("real", MOM[0], G["std_hat"]), G["std_inv"][0], rtol=1e-12)
    np.testing.assert_allclose(so.std_inverse_transform("pos", MOM[1], G["std_hat"]), G["std_inv"][1], rtol=1e-12)


def test_oracle_categorical_round_trip():
    tl, data, mask = _mixed_case()
    mom = so.fit(tl, data, mask)
    x, m = so.transform(tl, mom, data, mask)
    assert x.shape[-1] == 1 + 3 + 1 + 1 + 5 + 1
    back, bm = so.inverse_transform(tl, mom, x, m, transpose=False)
    np.testing.assert_array_equal(bm, mask)
    np.testing.assert_allclose(back, data, rtol=1e-9, atol=1e-9, equal_nan=True)


def test_scaler_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import shivae_amd
    sc = shivae_amd.HeterogeneousScaler(TL)
    with pytest.raises(shivae_amd._native.ShiVAENativeError):
        sc.transform(G["data"], G["mask"])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_transform_vs_reference_golden():
    import shivae_amd
    sc = shivae_amd.HeterogeneousScaler(TL)
    sc.fit(G["data"], G["mask"])
    np.testing.assert_allclose(np.stack([sc.mean, sc.std], -1), G["moments"], rtol=1e-12)
    x, m = sc.transform(G["data"], G["mask"])
    np.testing.assert_array_equal(m, G["xmask"])
    np.testing.assert_allclose(x, G["x"], rtol=2e-6, atol=2e-6, equal_nan=True)   # fp32 on the device, fp64 in the reference
    xt, mt = sc.transform_to_device(G["data"], G["mask"], transpose=True, zero_fill=True)
    np.testing.assert_allclose(xt.cpu().numpy(), G["xT"], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(mt.cpu().numpy(), G["xmask"].transpose(1, 0, 2))
    d, dm = sc.inverse_transform(G["xT"], G["xmask"].transpose(1, 0, 2), transpose=True)
    np.testing.assert_array_equal(dm, G["invmask"])
    np.testing.assert_allclose(d, G["inv"], rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_gpu_categorical_vs_oracle_and_round_trip():
    import shivae_amd
    tl, data, mask = _mixed_case(seed=11, n=37, t=23)
    sc = shivae_amd.HeterogeneousScaler(tl)
    sc.fit(data, mask)
    mom = so.fit(tl, data, mask)
    np.testing.assert_allclose(np.stack([sc.mean, sc.std], -1), np.array(mom), rtol=1e-12)
    xo, mo = so.transform(tl, mom, data, mask)
    x, m = sc.transform(data, mask)
    np.testing.assert_array_equal(m, mo)
    np.testing.assert_allclose(x, xo, rtol=2e-6, atol=2e-6, equal_nan=True)
    # mask=None -> isnan(data) (scalers.py:98-99)
    x2, m2 = sc.transform(data, None)
    np.testing.assert_array_equal(m2, mo)
    back, bm = sc.inverse_transform(x, m, transpose=False)
    np.testing.assert_array_equal(bm, mask)
    np.testing.assert_allclose(back, data, rtol=2e-5, atol=2e-5, equal_nan=True)
    bo, _ = so.inverse_transform(tl, mom, xo, mo, transpose=True)
    bt, _ = sc.inverse_transform(x, m, transpose=True)
    np.testing.assert_allclose(bt, bo, rtol=2e-5, atol=2e-5, equal_nan=True)


@pytest.mark.gpu
def test_gpu_trainer_preprocess_uses_device_pass():
    import types as _t
    import shivae_amd
    tl, data, mask = _mixed_case(seed=5, n=16, t=12)
    sc = shivae_amd.HeterogeneousScaler(tl)
    sc.fit(data, mask)
    model = shivae_amd.ShiVAE(16, 4, 3, 1, types_list=tl, learn_std=True).to("cuda")
    args = _t.SimpleNamespace(l_rate=1e-3, n_epochs=1, kl_annealing_epochs=1, clip=0.5, save_every=10, print_every=10,
                              save_model=False, result_dir="/tmp/shivae_scaler_test", model="shivae", restore=False)
    tr = shivae_amd.Trainer(model, shivae_amd.FusedClipAdam(model, lr=1e-3, clip=0.5), args, scaler=sc)
    x, m = tr.preprocess_batch(data, mask)
    assert x.is_cuda and x.shape == (12, 16, sc.x_dim) and m.dtype == torch.bool
    xo, mo = so.transform(tl, so.fit(tl, data, mask), data, mask)
    np.testing.assert_allclose(x.cpu().numpy(), np.where(mo, 0.0, xo).transpose(1, 0, 2), rtol=2e-6, atol=2e-6)
    loss = model(x, m, beta=1.0, temp=1.0)
    assert np.isfinite(float(loss["n_elbo"]))
