"""Generate the golden vectors in tests/golden/*.npz by IMPORTING the reference model.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py

Two shims are applied by this script, not by the reference:
  1. a stub ``seaborn`` module (``lib/utils.py:14`` imports it at module top, not installed);
  2. ``torch.distributions`` argument validation off (torch 1.6, the pinned version, did not
     validate by default; modern torch rejects the zero-filled one-hot rows that
     ``OneHotCategorical.log_prob`` sees at ``base_missvae.py:277``).

Noise replay: the reference draws, per timestep, ``torch.rand(B,K)`` (samplers.py:25) then
``torch.FloatTensor(B,Z).normal_()`` (samplers.py:60) on the CPU generator.  We pre-draw the
same stream after ``torch.manual_seed(seed)``, save it as ``u``/``eps``, re-seed and run the
reference, which then consumes exactly those numbers.

Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def make_types(n_real, n_pos, n_bin, cat_classes, interleave=True):
    groups = [("real", n_real, 1), ("pos", n_pos, 1), ("bin", n_bin, 1)]
    attrs = []
    for ty, n, _ in groups:
        attrs += [(ty, 1, 1 if ty != "bin" else 2)] * n
    attrs += [("cat", c, c) for c in cat_classes]
    if interleave:  # mix the types so nothing relies on type-sorted attribute order
        order = []
        buckets = {}
        for a in attrs:
            buckets.setdefault(a[0], []).append(a)
        while any(buckets.values()):
            for ty in ("real", "pos", "bin", "cat"):
                if buckets.get(ty):
                    order.append(buckets[ty].pop(0))
        attrs = order
    return [dict(name="var%d" % i, type=ty, dim=str(dim), nclass=str(nc), index=str(i))
            for i, (ty, dim, nc) in enumerate(attrs)]


def synth_batch(rng, types_list, T, B, miss):
    """Scaled, one-hot expanded, zero-filled batch + mask (True = missing), [T,B,X]."""
    cols, mcols = [], []
    for t in types_list:
        ty, dim = t["type"], int(t["dim"])
        m = rng.random((T, B)) < miss
        if ty in ("real", "pos"):
            v = rng.standard_normal((T, B, 1))
        elif ty == "bin":
            v = (rng.random((T, B, 1)) < 0.5).astype(np.float64)
        else:
            idx = rng.integers(0, dim, size=(T, B))
            v = np.eye(dim)[idx]
        cols.append(v)
        mcols.append(np.repeat(m[..., None], dim, axis=-1))
    x = np.concatenate(cols, -1).astype(np.float32)
    mask = np.concatenate(mcols, -1)
    x = np.where(mask, 0, x).astype(np.float32)
    return x, mask


CASES = {
    # cfg-1 shape (main_hmm.py:24, data_types.csv, process_args.py defaults), reduced B
    "cfg1_T100_B8": dict(types=make_types(1, 1, 1, [3], interleave=False), T=100, B=8, H=10, Z=2, K=3,
                         learn_std=True, beta=1.0, temp=3.0, miss=0.3, seed=11),
    "tiny_all": dict(types=make_types(2, 2, 2, [3, 5]), T=7, B=5, H=16, Z=4, K=3,
                     learn_std=True, beta=0.7, temp=1.0, miss=0.3, seed=12),
    "tiny_fixedstd": dict(types=make_types(2, 2, 2, [3, 5]), T=7, B=5, H=16, Z=4, K=3,
                          learn_std=False, beta=1.0, temp=3.0, miss=0.3, seed=13),
    "tiny_lowtemp": dict(types=make_types(2, 2, 2, [3, 5]), T=7, B=5, H=16, Z=4, K=3,
                         learn_std=True, beta=1.0, temp=0.1, miss=0.3, seed=14),
    "tiny_miss90": dict(types=make_types(1, 1, 1, [4, 8, 8]), T=9, B=6, H=12, Z=3, K=5,
                        learn_std=True, beta=0.25, temp=2.0, miss=0.9, seed=15),
    # physionet-like stand-in (main_physionet.py:25: z_dim=35, K=10; all-real variant), reduced
    "cfg2like": dict(types=make_types(9, 0, 0, [], interleave=False), T=12, B=4, H=10, Z=35, K=10,
                     learn_std=True, beta=1.0, temp=3.0, miss=0.4, seed=16),
}


def main():
    sys.path.insert(0, REF)
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    import torch
    torch.distributions.Distribution.set_default_validate_args(False)
    from models.shivae import ShiVAE  # the reference
    torch.set_num_threads(1)

    for name, cfg in CASES.items():
        tl, T, B, H, Z, K = cfg["types"], cfg["T"], cfg["B"], cfg["H"], cfg["Z"], cfg["K"]
        rng = np.random.default_rng(cfg["seed"])
        x, mask = synth_batch(rng, tl, T, B, cfg["miss"])
        torch.manual_seed(cfg["seed"])
        model = ShiVAE(H, Z, K, 1, types_list=tl, learn_std=cfg["learn_std"])
        # non-zero biases so bias gradients/paths are exercised (init sets them to 0)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "bias" in n:
                    p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.1))
        sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}

        nseed = 1000 + cfg["seed"]
        torch.manual_seed(nseed)
        u = np.zeros((T, B, K), np.float32)
        eps = np.zeros((T, B, Z), np.float32)
        for t in range(T):
            u[t] = torch.rand(B, K).numpy()
            eps[t] = torch.FloatTensor(B, Z).normal_().numpy()

        xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
        out = {"x": x, "mask": mask, "u": u, "eps": eps,
               "beta": np.float32(cfg["beta"]), "temp": np.float32(cfg["temp"]),
               "dims": np.array([T, B, H, Z, K], np.int64), "learn_std": np.bool_(cfg["learn_std"]),
               "types_csv": np.array(["%s,%s,%s,%s,%s" % (t["name"], t["type"], t["dim"], t["nclass"], t["index"])
                                      for t in tl])}
        for k, v in sd0.items():
            out["sd/" + k] = v

        # ---- forward (feed2model) with replayed noise ----
        torch.manual_seed(nseed)
        z_tup, z_prior_tup, s_tup, likes = model.feed2model(xt, cfg["temp"])
        out["z"], out["z_mean"], out["z_std"] = (a.detach().numpy() for a in z_tup)
        out["zp_mean"], out["zp_std"] = (a.detach().numpy() for a in z_prior_tup)
        out["s"], out["s_probs"] = (a.detach().numpy() for a in s_tup)
        tidx = model.transform_idx
        for i, t in enumerate(tl):
            lk = likes[t["name"]]
            if t["type"] in ("real", "pos"):
                out["dec/%d/mean" % i] = lk.loc.detach().numpy()[..., 0]
                out["dec/%d/std" % i] = lk.scale.detach().numpy()[..., 0]
            elif t["type"] == "bin":
                out["dec/%d/logits" % i] = lk.logits.detach().numpy()[..., 0]
            else:
                # OneHotCategorical normalises its logits; recover theta with theta[0]=0
                lg = lk.logits.detach().numpy()
                out["dec/%d/theta" % i] = lg - lg[..., :1]
            lo, dim = tidx[i], int(t["dim"])
            nll = -lk.log_prob(xt[:, :, lo:lo + dim])
            if nll.dim() == 3:
                nll = nll.squeeze(-1)
            m_d = mt[:, :, lo:lo + dim][..., -1]
            out["nll_attr/%d" % i] = torch.where(m_d, torch.zeros_like(nll), nll).detach().numpy()
        nll_dict = model.heter_nll(xt, mt, likes, types_list=model.types_list, eval_obs=False)
        for k, v in nll_dict.items():
            out["nll_b/" + k] = v.detach().numpy() * np.ones(B, np.float32)
        out["klz_b"] = model.kl_q_z(z_tup, z_prior_tup).detach().numpy()
        out["kls_b"] = model.kl_q_s(s_tup).detach().numpy()

        # ---- full step: model(x, mask, beta, temp) -> backward -> clip -> Adam ----
        model.zero_grad()
        opt = torch.optim.Adam(model.parameters(), lr=5e-3)
        torch.manual_seed(nseed)
        with contextlib.redirect_stdout(io.StringIO()):
            loss = model(xt, mask=mt, beta=cfg["beta"], temp=cfg["temp"])
        for k, v in loss.items():
            out["loss/" + k] = np.float32(v.detach().numpy())
        loss["n_elbo"].backward()
        for n, p in model.named_parameters():
            out["grad/" + n] = p.grad.detach().numpy().copy()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        out["grad_norm"] = np.float32(gn)
        opt.step()
        for n, p in model.named_parameters():
            out["adam1/" + n] = p.detach().numpy().copy()

        # ---- imputation parameters (reconstruction at temp=0.01, base_missvae.py:345-359) ----
        if name in ("cfg1_T100_B8", "tiny_all"):
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
            torch.manual_seed(nseed)
            _, _, _, likes_r = model.feed2model(xt, 0.01)
            for i, t in enumerate(tl):
                lk = likes_r[t["name"]]
                if t["type"] in ("real", "pos"):
                    out["rec/%d/mean" % i] = lk.loc.detach().numpy()[..., 0]
                elif t["type"] == "bin":
                    out["rec/%d/logits" % i] = lk.logits.detach().numpy()[..., 0]
                else:
                    lg = lk.logits.detach().numpy()
                    out["rec/%d/theta" % i] = lg - lg[..., :1]

        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "n_elbo=%.5f" % out["loss/n_elbo"], "grad_norm=%.5f" % out["grad_norm"],
              "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
