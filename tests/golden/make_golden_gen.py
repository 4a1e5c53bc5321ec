"""Golden vectors for generation from the prior, made by IMPORTING the reference model (same shims and noise
replay as make_golden.py; sample_from_prior draws torch.rand(B,K) then FloatTensor(B,Z).normal_() per step,
samplers.py:25,60):   python tests/golden/make_golden_gen.py
Weights come from existing fixtures (tiny_all, cfg2like) so no new state_dict is stored."""
import os, sys, types
import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _golden  # noqa: E402

sys.path.insert(0, REF)
sys.modules["seaborn"] = types.ModuleType("seaborn")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
import torch  # noqa: E402
torch.distributions.Distribution.set_default_validate_args(False)
from models.shivae import ShiVAE  # noqa: E402

for base, T, B, temp, seed in (("tiny_all", 9, 6, 0.5, 31), ("cfg2like", 14, 5, 0.02, 32)):
    g, cfg, sd = _golden.load(base)
    tl, H, Z, K = cfg["types_list"], cfg["H"], cfg["Z"], cfg["K"]
    model = ShiVAE(H, Z, K, 1, types_list=tl, learn_std=cfg["learn_std"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    torch.manual_seed(seed)
    u, eps = np.zeros((T, B, K), np.float32), np.zeros((T, B, Z), np.float32)
    for t in range(T):
        u[t] = torch.rand(B, K).numpy()
        eps[t] = torch.FloatTensor(B, Z).normal_().numpy()
    torch.manual_seed(seed)
    with torch.no_grad():
        z_prior_tup, s_tup, likes = model.sample_from_prior(B, T, temp)
    out = dict(base=np.array(base), u=u, eps=eps, temp=np.float32(temp))
    out["z"], out["mean"], out["std"] = (a.numpy() for a in z_prior_tup)
    out["s"], out["s_probs"] = (a.numpy() for a in s_tup)
    for i, t in enumerate(tl):
        lk = likes[t["name"]]
        if t["type"] in ("real", "pos"):
            out["dec/%d/mean" % i], out["dec/%d/std" % i] = lk.loc.numpy()[..., 0], lk.scale.numpy()[..., 0]
        elif t["type"] == "bin":
            out["dec/%d/logits" % i] = lk.logits.numpy()[..., 0]
        else:
            lg = lk.logits.numpy()
            out["dec/%d/theta" % i] = lg - lg[..., :1]
    np.savez_compressed(os.path.join(HERE, "gen_%s.npz" % base), **out)
    print("gen_%s" % base, out["z"].shape, float(np.abs(out["z"]).mean()))
