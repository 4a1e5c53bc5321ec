"""Imputation evaluation: metrics vs golden vectors from the reference's lib/loss.py (CPU);
Monte-Carlo imputation kernel vs the exact laws of its outputs (GPU).  The reference draws its 100 samples from
torch's device generator, so sample streams cannot be compared -- parity of the MC path is on distributions."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_mixed.npz"))
TL = [dict(name="v%d" % i, type=str(t), dim=str(d), nclass=str(d), index=str(i)) for i, (t, d) in enumerate(zip(G["types"], G["dims"]))]


def test_metrics_match_reference():
    from shivae_amd import metrics
    np.testing.assert_array_equal(metrics.artmask2burst(G["mask"][0, :, 0]), G["bursts"])
    _, err, _, corr = metrics.compute_avg_error(G["x"].copy(), G["xh"].copy(), G["mask"], TL)
    np.testing.assert_allclose([err[t["name"]] for t in TL], G["err"], rtol=1e-12)
    np.testing.assert_allclose([corr[t["name"]] for t in TL], G["corr"], rtol=1e-10)
    by_type = metrics.get_type_error(err, TL)
    assert set(by_type) == {"real", "pos", "bin", "cat"}


def test_metrics_edge_cases():
    from shivae_amd import metrics
    x = np.zeros((2, 5, 1)); xh = np.ones((2, 5, 1)); m = np.zeros((2, 5, 1), dtype=int)
    tl = [dict(name="a", type="real", dim="1", nclass="1", index="0")]
    _, err, _, corr = metrics.compute_avg_error(x, xh, m, tl)          # nothing missing (loss.py:138-140)
    assert err["a"] == 0 and corr["a"] == 0
    m[0, 2, 0] = 1                                                      # one missing entry, constant signal -> norm 1
    _, err, _, _ = metrics.compute_avg_error(x, xh, m, tl)
    assert err["a"] == pytest.approx(1.0)
    np.testing.assert_array_equal(metrics.artmask2burst(np.array([0, 1, 1, 1, 0, 0, 0, 1, 1, 1])), [0, 1, 1, 1, 0, 0, 0, 2, 2, 2])
    np.testing.assert_array_equal(metrics.artmask2burst(np.array([1, 0, 1])), [1, 0, 2])


@pytest.mark.gpu
def test_gpu_mc_imputation_laws():
    from scipy import stats
    import shivae_amd
    tl = shivae_amd.utils.make_types_list(2, 2, 2, [3, 5], interleave=True)
    model = shivae_amd.ShiVAE(16, 4, 3, 1, types_list=tl, learn_std=True).to("cuda")
    T, B, n = 50, 400, 100
    g = torch.Generator(device="cuda").manual_seed(0)
    P = torch.randn(T, B, model.p_dim, device="cuda", generator=g)
    likes = model._likes_from_params(P)
    xh = model.generate_heter_samples(likes, n_samples=n, seed=1234)
    assert xh.shape == (T, B, model.x_dim)
    xh2 = model.generate_heter_samples(likes, n_samples=n, seed=1234)
    assert torch.equal(xh, xh2)                                         # deterministic per seed
    assert not torch.equal(xh, model.generate_heter_samples(likes, n_samples=n, seed=99))
    xh = xh.cpu().numpy().reshape(T * B, -1)
    xo = 0
    for t in tl:
        lik, C = likes[t["name"]], int(t["dim"])
        if t["type"] in ("real", "pos"):                               # mean of n Normal draws ~ N(mu, sd^2/n)
            mu, sd = lik.loc.cpu().numpy().reshape(-1), lik.scale.cpu().numpy().reshape(-1)
            z = (xh[:, xo] - mu) / (sd / np.sqrt(n))
            assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 0.05
            assert stats.kstest(z, "norm").pvalue > 1e-4
        elif t["type"] == "bin":                                        # round(Binomial(n,p)/n): 1 iff count > n/2
            p = lik.probs.cpu().numpy().reshape(-1)
            assert set(np.unique(xh[:, xo])) <= {0.0, 1.0}
            expect = stats.binom.sf(n // 2, n, p)
            assert abs(xh[:, xo].mean() - expect.mean()) < 5 * np.sqrt((expect * (1 - expect)).sum()) / p.size + 1e-3
        else:                                                           # per class: 1 iff its multinomial count > n/2
            pr = lik.probs.cpu().numpy().reshape(-1, C)
            oh = xh[:, xo:xo + C]
            assert set(np.unique(oh)) <= {0.0, 1.0} and (oh.sum(-1) <= 1).all()
            expect = stats.binom.sf(n // 2, n, pr)
            dev = np.abs(oh.mean(0) - expect.mean(0))
            assert (dev < 5 * np.sqrt((expect * (1 - expect)).sum(0)) / pr.shape[0] + 1e-3).all()
        xo += C


@pytest.mark.gpu
def test_gpu_reconstruction_end_to_end():
    import shivae_amd
    from shivae_amd.synth import synth_batch
    tl = shivae_amd.utils.make_types_list(2, 1, 1, [3])
    torch.manual_seed(0)
    model = shivae_amd.ShiVAE(16, 4, 3, 1, types_list=tl, learn_std=True).to("cuda")
    x, mask = synth_batch(tl, 20, 32, miss=0.3, seed=5, device="cuda")
    (likes, xhat), (z_tup, s_tup) = model.reconstruction(x, temp=0.01)
    assert xhat.shape == x.shape and torch.isfinite(xhat).all()
    # MC mean of the Normal attributes stays within 6 sigma/sqrt(100) of the likelihood mean
    lik = likes[tl[0]["name"]]
    assert ((xhat[..., :1] - lik.loc).abs() <= 0.6 * lik.scale + 1e-6).all()


@pytest.mark.gpu
def test_heter_nll_on_attribute_subsets():
    """base_missvae.py:233-252 takes the list of attributes to evaluate: the terms of two complementary subsets add up
    to the terms of the whole list, per type and per sequence; unknown attributes are refused."""
    import _golden
    import shivae_amd
    g, cfg, sd = _golden.load("tiny_all")
    m = shivae_amd.ShiVAE(cfg["H"], cfg["Z"], cfg["K"], 1, types_list=cfg["types_list"], learn_std=cfg["learn_std"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    x, mask = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    m.set_noise(g["u"], g["eps"])
    _, _, _, likes = m.feed2model(x, cfg["temp"])
    full = m.heter_nll(x, mask, likes)
    tl = m.types_list
    a, b = tl[::2], tl[1::2]
    na, nb = m.heter_nll(x, mask, likes, types_list=a), m.heter_nll(x, mask, {t["name"]: likes[t["name"]] for t in b}, types_list=b)
    for k in full:
        np.testing.assert_allclose((na[k] + nb[k]).cpu().numpy(), full[k].cpu().numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(full[k].cpu().numpy(), g["nll_b/" + k], rtol=1e-4, atol=2e-4, err_msg=k)
    assert float(sum(v.abs().sum() for v in na.values())) > 0 and float(sum(v.abs().sum() for v in nb.values())) > 0
    with pytest.raises(KeyError):
        m.heter_nll(x, mask, likes, types_list=[dict(name="no_such_var", type="real", dim="1", nclass="", index="0")])
