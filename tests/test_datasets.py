"""Synthetic heterogeneous-HMM generator + on-disk formats + loaders and the main_hmm.py
flow end to end: generator -> pickle/csv -> HMMDatset -> DataLoader(standard_collate) -> HeterogeneousScaler ->
Trainer.train (GPU).  The generator has no reference golden (hmmlearn is not installed): statistical checks."""
import types as _t

import numpy as np
import pytest
import torch


def _toy(tmp_path, N=60, T=100, seed=3):
    from shivae_amd import datasets
    g = datasets.ToyDatasetGenerator(1, 1, 1, 1, [3], 3, T, N, save_root=str(tmp_path), seed=seed)
    return g, g.generate_toydataset((0.3, 0.5), num_masks=2)


def test_generator_files_and_statistics(tmp_path):
    from shivae_amd import datasets, metrics
    g, d = _toy(tmp_path)
    assert g.toy_name == "hmm_heter_60_1real_1pos_1bin_1cat_3_100"          # heterHMMdataset.py:71-73
    for k in ("X", "Z", "trans_mat", "start_prob", "theta", "train_index", "val_index", "test_index", "mask_30_1",
              "mask_30_2", "mask_50_1", "mask_50_2"):
        assert k in d
    X, Z, theta, mask, (tr, va, te) = datasets.load_data_pickle(g.toy_name, str(tmp_path), percent_miss=30, mask_i=2)
    assert X.shape == (60, 100, 4) and Z.shape == (60, 100) and mask.shape == X.shape and mask.dtype == bool
    assert (len(tr), len(va), len(te)) == (48, 6, 6) and len(set(tr) | set(va) | set(te)) == 60
    assert (mask.sum(1) == 30).all()                                        # exactly round(T*p) per dimension
    runs = [np.bincount(metrics.artmask2burst(mask[n, :, dd]))[1:] for n in range(60) for dd in range(4)]
    assert np.mean(np.concatenate(runs)) > 2.5                              # bursts, not isolated points
    tl = datasets.read_types(str(tmp_path / g.toy_name / "data_types.csv"))
    assert [t["type"] for t in tl] == ["real", "pos", "bin", "cat"] and tl[3]["dim"] == "3" and tl[2]["nclass"] == "2"
    # chain statistics follow the K=3 transition matrix (heterHMMdataset.py:76-80)
    cnt = np.zeros((3, 3))
    for z in Z:
        np.add.at(cnt, (z[:-1], z[1:]), 1)
    np.testing.assert_allclose(cnt / cnt.sum(1, keepdims=True), g.transmat, atol=0.04)
    assert (X[..., 1] > 0).all() and set(np.unique(X[..., 2])) <= {0.0, 1.0} and set(np.unique(X[..., 3])) <= {0.0, 1.0, 2.0}
    # emissions: real mean per state, deterministic binary per state, dominant categorical class per state
    for k in range(3):
        sel = Z == k
        assert abs(X[..., 0][sel].mean() - g.means_gauss[k, 0]) < 0.2
        assert len(np.unique(X[..., 2][sel])) == 1
        assert np.bincount(X[..., 3][sel].astype(int), minlength=3).argmax() == g.prob_cat[0][k].argmax()


def test_dataset_and_collate(tmp_path):
    from shivae_amd import datasets
    g, d = _toy(tmp_path, N=20, T=30)
    tl = g.types_list()
    X = np.where(d["mask_30_1"], np.nan, d["X"])
    ds = datasets.HMMDatset(X, d["mask_30_1"], data_full=d["X"], mask_artificial=d["mask_30_1"], types_list=tl,
                            index=d["train_index"], Z=d["Z"], theta=d["theta"])
    assert len(ds) == 16
    loader = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, collate_fn=datasets.standard_collate)
    data, mask, attrs = next(iter(loader))
    assert data.shape == (5, 30, 4) and mask.dtype == bool and attrs["sequence_lengths"] == [30] * 5
    np.testing.assert_array_equal(attrs["index"], np.arange(5))
    ds.set_eval(True)
    data, mask, full, art, attrs = next(iter(loader))
    np.testing.assert_array_equal(full, d["X"][d["train_index"][:5]])
    assert art.dtype == bool


@pytest.mark.gpu
def test_gpu_main_hmm_flow(tmp_path):
    """main_hmm.py:60-160 on the HIP path: two epochs lower the -ELBO, imputation error is finite."""
    import shivae_amd
    from shivae_amd import datasets, metrics
    g, d = _toy(tmp_path, N=80, T=40, seed=5)
    tl = g.types_list()
    mask = d["mask_30_1"]
    X = np.where(mask, np.nan, d["X"])
    scaler = shivae_amd.HeterogeneousScaler(tl)
    scaler.fit(X[d["train_index"]], mask[d["train_index"]])
    mk = lambda idx: torch.utils.data.DataLoader(  # noqa: E731
        datasets.HMMDatset(X, mask, data_full=d["X"], mask_artificial=mask, types_list=tl, index=idx, Z=d["Z"], theta=d["theta"]),
        batch_size=32, shuffle=False, collate_fn=datasets.standard_collate)
    torch.manual_seed(0)
    model = shivae_amd.ShiVAE(10, 2, 3, 1, types_list=tl, learn_std=True).to("cuda")
    args = _t.SimpleNamespace(l_rate=5e-3, n_epochs=6, kl_annealing_epochs=2, clip=0.5, save_every=100, print_every=100,
                              save_model=False, result_dir=str(tmp_path / "run"), model="shivae", restore=False)
    tr = shivae_amd.Trainer(model, shivae_amd.FusedClipAdam(model, lr=5e-3, clip=0.5), args, scaler=scaler)
    first = tr.train_epoch(mk(d["train_index"]), 1)["nll_loss"]
    for ep in range(2, 7):
        last = tr.train_epoch(mk(d["train_index"]), ep)["nll_loss"]
    assert np.isfinite(last) and last < first
    # imputation on the test split: reconstruct, de-scale, score the artificially missing entries
    te = d["test_index"]
    x, m = scaler.transform_to_device(X[te], mask[te])
    (likes, xhat), _ = model.reconstruction(x, temp=0.01)
    data_hat, _ = scaler.inverse_transform(xhat, m, transpose=True)
    assert data_hat.shape == d["X"][te].shape
    _, err, _, corr = metrics.compute_avg_error(d["X"][te], data_hat, mask[te].astype(int), tl)
    assert all(np.isfinite(v) for v in err.values()) and all(np.isfinite(v) for v in corr.values())
    # the same evaluation inside valid_epoch (trainers.py:286-333), every args.plot_every epochs
    args.plot_every = 1
    stop, va = tr.valid_epoch(mk(d["val_index"]), 7)
    assert np.isfinite(va["n_elbo"]) and tr.last_avg_error is not None
    err_v, corr_v = tr.last_avg_error
    assert set(err_v) == {t["name"] for t in tl} and all(np.isfinite(v) for v in err_v.values())
