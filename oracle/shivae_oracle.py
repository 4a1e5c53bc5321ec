"""CPU oracle for the Shi-VAE train step (ELBO forward + hand-derived backward).

TEST INFRASTRUCTURE ONLY.  Nothing under ``shi-vae_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
only as the checker / the timed CPU baseline, never as the product path.

It is a from-scratch numpy restatement of the algorithm in the reference
(``/root/reference/src``), with *explicit noise* (``u`` for the Gumbel-softmax, ``eps``
for the Gaussian reparameterisation) and *explicit gradients* (no autograd).  Each
function cites the reference file:line it follows.  Parity is pinned by the golden
vectors in ``tests/golden/*.npz`` which were produced by importing the reference model
in the build container (``tests/golden/make_golden.py``); the reference has no tests or
fixtures of its own.

Third-party arithmetic restated here (the reference delegates it to ``torch``, pinned
``torch==1.6.0`` in ``src/requirements.txt:2``; the golden vectors were made with torch
2.10.0): ``nn.Linear``, ``nn.ReLU``, ``nn.LeakyReLU(0.01)``, ``nn.Softplus`` (beta=1,
threshold=20), single-layer ``nn.LSTM`` (gate order i,f,g,o),
``Normal/Bernoulli/OneHotCategorical.log_prob``, ``kl_divergence`` for Normal-Normal
and Categorical-Categorical (with ``probs_to_logits`` clamp, eps=float32 eps).

All arrays are time-major ``[T, B, .]``; ``mask`` is bool, True = missing.
"""
from __future__ import annotations

import numpy as np

F32_EPS = float(np.finfo(np.float32).eps)  # torch.distributions.utils.clamp_probs eps
HALF_LOG_2PI = 0.5 * np.log(2.0 * np.pi)
FIXED_STD = 0.1  # decoders.py:22 fixed_std


# --------------------------------------------------------------------------------------
# types_list helpers (lib/utils.py:229-241, 275-288, 291-315)
# --------------------------------------------------------------------------------------
def x_dim(types_list):
    """lib/utils.py:229-241 get_x_dim."""
    return int(sum(int(t["dim"]) for t in types_list))


def transform_idx(types_list):
    """lib/utils.py:291-315 get_idx_transform: prefix sum of ``dim``."""
    out, off = [], 0
    for t in types_list:
        out.append(off)
        off += int(t["dim"])
    return out


# --------------------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------------------
def relu(a):
    return np.maximum(a, 0)


def softplus(a):
    """nn.Softplus(beta=1, threshold=20) (encoders.py:68, shivae.py:61, decoders.py:74)."""
    return np.where(a > 20, a, np.log1p(np.exp(np.minimum(a, 20))))


def softplus_grad_from_out(sp):
    """d softplus(a)/da written with the output: sigmoid(a) = 1 - exp(-softplus(a))."""
    return 1.0 - np.exp(-sp)


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def softmax(a):
    m = a.max(axis=-1, keepdims=True)
    e = np.exp(a - m)
    return e / e.sum(axis=-1, keepdims=True)


def gumbel(u):
    """samplers.py:23-26 sample_gumbel, eps=1e-20 in both places."""
    dt = u.dtype.type
    return -np.log(-np.log(u + dt(1e-20)) + dt(1e-20))


def lin(a, W, b):
    return a @ W.T + b


# --------------------------------------------------------------------------------------
# loss terms (standalone; these are what the HIP likelihood/KL kernel is checked against)
# --------------------------------------------------------------------------------------
def attr_nll(ttype, x_d, params_d):
    """Negative log-likelihood of one attribute, *before* masking; [T,B].

    base_missvae.py:257-284 + decparams2likes :111-142.
      real/pos: Normal(mu, std).log_prob          (params_d = (mu, std), each [T,B])
      bin     : Bernoulli(logits=l).log_prob      (params_d = l [T,B])
      cat     : OneHotCategorical(logits=theta)   (params_d = theta [T,B,C], theta[...,0]=0)
    """
    if ttype in ("real", "pos"):
        mu, std = params_d
        x = x_d[..., 0]
        return (x - mu) ** 2 / (2.0 * std ** 2) + np.log(std) + HALF_LOG_2PI
    if ttype == "bin":
        l = params_d
        x = x_d[..., 0]
        return np.maximum(l, 0) - l * x + np.log1p(np.exp(-np.abs(l)))
    if ttype == "cat":
        theta = params_d
        m = theta.max(axis=-1, keepdims=True)
        lse = (m + np.log(np.exp(theta - m).sum(axis=-1, keepdims=True)))[..., 0]
        idx = x_d.argmax(axis=-1)  # first max, as torch.max(-1)[1]
        return lse - np.take_along_axis(theta, idx[..., None], axis=-1)[..., 0]
    raise ValueError(ttype)


def attr_nll_grad(ttype, x_d, params_d):
    """d nll / d params (before masking), same shapes as params_d."""
    if ttype in ("real", "pos"):
        mu, std = params_d
        x = x_d[..., 0]
        dmu = -(x - mu) / std ** 2
        dstd = -(x - mu) ** 2 / std ** 3 + 1.0 / std
        return dmu, dstd
    if ttype == "bin":
        return sigmoid(params_d) - x_d[..., 0]
    if ttype == "cat":
        theta = params_d
        p = softmax(theta)
        idx = x_d.argmax(axis=-1)
        oh = np.zeros_like(theta)
        np.put_along_axis(oh, idx[..., None], 1.0, axis=-1)
        return p - oh
    raise ValueError(ttype)


def attr_missing(ttype, mask_d):
    """base_missvae.py:271-274: single mask column; for cat the LAST one-hot column."""
    return mask_d[..., -1]


def heter_nll(x, mask, dec_params, types_list):
    """base_missvae.py:233-285 (eval_obs=False). Returns dict type -> [B] and per-attr [T,B]."""
    tidx = transform_idx(types_list)
    B = x.shape[1]
    out = {k: np.zeros(B, dtype=x.dtype) for k in ("real", "pos", "bin", "cat")}
    per_attr = []
    for i, t in enumerate(types_list):
        lo, dim = tidx[i], int(t["dim"])
        nll = attr_nll(t["type"], x[:, :, lo:lo + dim], dec_params[i])
        miss = attr_missing(t["type"], mask[:, :, lo:lo + dim])
        nll = np.where(miss, 0, nll)
        per_attr.append(nll)
        out[t["type"]] = out[t["type"]] + nll.sum(axis=0)
    return out, per_attr


def kl_q_z(mu_q, std_q, mu_p, std_p):
    """base_missvae.py:287-318 -> kl[T,B,Z] (caller sums axis (0,2))."""
    sq = np.clip(std_q, 1e-6, 1e20)
    sp = np.clip(std_p, 1e-6, 1e20)
    mq = np.clip(mu_q, -1e20, 1e20)
    mp = np.clip(mu_p, -1e20, 1e20)
    r = (sq / sp) ** 2
    t1 = ((mq - mp) / sp) ** 2
    kl = 0.5 * (r + t1 - 1.0 - np.log(r))
    return np.where(np.isnan(kl), 0, kl)


def kl_q_z_grad(mu_q, std_q, mu_p, std_p):
    """Gradients of kl_q_z elementwise wrt (mu_q, std_q, mu_p, std_p); clamp passes
    gradient only inside its range (torch.clamp backward)."""
    sq = np.clip(std_q, 1e-6, 1e20)
    sp = np.clip(std_p, 1e-6, 1e20)
    in_q = (std_q >= 1e-6) & (std_q <= 1e20)
    in_p = (std_p >= 1e-6) & (std_p <= 1e20)
    d = mu_q - mu_p
    dmq = d / sp ** 2
    dsq = (sq / sp ** 2 - 1.0 / sq) * in_q
    dsp = (-(sq ** 2) / sp ** 3 - d ** 2 / sp ** 3 + 1.0 / sp) * in_p
    return dmq, dsq, -dmq, dsp


def kl_q_s(probs):
    """base_missvae.py:320-343 -> kl[T,B] (caller sums axis 0).

    Categorical(probs) renormalises, logits = log(clamp(p, eps, 1-eps));
    _kl_categorical_categorical: sum p * (logp - logq), p==0 -> 0; prior uniform."""
    K = probs.shape[-1]
    p = probs / probs.sum(axis=-1, keepdims=True)
    logp = np.log(np.clip(p, F32_EPS, 1.0 - F32_EPS))
    q = np.full_like(probs, 1.0 / K)
    q = q / q.sum(axis=-1, keepdims=True)
    logq = np.log(np.clip(q, F32_EPS, 1.0 - F32_EPS))
    t = p * (logp - logq)
    t = np.where(p == 0, 0, t)
    kl = t.sum(axis=-1)
    return np.where(np.isnan(kl), 0, kl)


def kl_q_s_grad(probs):
    """d kl_q_s / d probs, [T,B,K]."""
    K = probs.shape[-1]
    S = probs.sum(axis=-1, keepdims=True)
    p = probs / S
    inr = (p >= F32_EPS) & (p <= 1.0 - F32_EPS)
    logp = np.log(np.clip(p, F32_EPS, 1.0 - F32_EPS))
    q = np.full_like(probs, 1.0 / K)
    q = q / q.sum(axis=-1, keepdims=True)
    logq = np.log(np.clip(q, F32_EPS, 1.0 - F32_EPS))
    dkl_dp = (logp - logq) + inr * 1.0
    dkl_dp = np.where(p == 0, 0, dkl_dp)
    return (dkl_dp - (p * dkl_dp).sum(axis=-1, keepdims=True)) / S


def reparam(mean, std, eps):
    """samplers.py:57-62: z = eps*std + mean."""
    return eps * std + mean


def gumbel_softmax(logits, u, temp):
    """samplers.py:28-30 (one_hot=False at every call site)."""
    return softmax((logits + gumbel(u)) / temp)


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------
class OracleShiVAE:
    """numpy Shi-VAE with the reference's state_dict key names.

    ``params``: dict name -> ndarray (any float dtype; converted to ``dtype``).
    """

    def __init__(self, params, types_list, h_dim, z_dim, s_dim, learn_std=False, dtype=np.float64):
        self.dt = np.dtype(dtype)
        self.P = {k: np.asarray(v, dtype=self.dt) for k, v in params.items()}
        self.types_list = [dict(t) for t in types_list]
        self.H, self.Z, self.K = int(h_dim), int(z_dim), int(s_dim)
        self.X = x_dim(types_list)
        self.learn_std = bool(learn_std)
        self.tidx = transform_idx(types_list)

    # ---------------------------------------------------------------- forward
    def forward(self, x, mask, u, eps, beta=1.0, temp=3.0):
        """shivae.py:72-154 feed2model + base_missvae.py:166-231 compute_loss.

        Returns a dict with every intermediate the kernels are checked against."""
        P, dt = self.P, self.dt
        x = np.asarray(x, dtype=dt)
        u = np.asarray(u, dtype=dt)
        eps = np.asarray(eps, dtype=dt)
        mask = np.asarray(mask, dtype=bool)
        T, B, X = x.shape
        H, Z, K = self.H, self.Z, self.K
        temp = dt.type(temp)
        c = {}  # cache

        # --- hoisted, h-independent parts (encoders.py:82, x-half of :83) ---
        p1 = relu(lin(x, P["encoder.phi_x.0.weight"], P["encoder.phi_x.0.bias"]))
        phi_x = relu(lin(p1, P["encoder.phi_x.2.weight"], P["encoder.phi_x.2.bias"]))
        Wxs0 = P["encoder.phi_x_s.0.weight"]  # [H, X+H], input cat[x_t, h]
        xs_pre = lin(x, Wxs0[:, :X], P["encoder.phi_x_s.0.bias"])
        We0 = P["encoder.enc.0.weight"]  # [H, 2H+K], input cat[phi_x, s, h]
        ex_pre = lin(phi_x, We0[:, :H], P["encoder.enc.0.bias"])
        c.update(p1=p1, phi_x=phi_x, xs_pre=xs_pre, ex_pre=ex_pre)

        names = ["h_prev", "c_prev", "a_s", "logits", "probs", "s", "e1", "e2", "mu", "std",
                 "z", "phi_z", "gi", "gf", "gg", "go", "c_new"]
        dims = dict(h_prev=H, c_prev=H, a_s=H, logits=K, probs=K, s=K, e1=H, e2=H, mu=Z, std=Z,
                    z=Z, phi_z=H, gi=H, gf=H, gg=H, go=H, c_new=H)
        for n in names:
            c[n] = np.zeros((T, B, dims[n]), dtype=dt)

        h = np.zeros((B, H), dtype=dt)  # shivae.py:93-94
        cc = np.zeros((B, H), dtype=dt)
        Wih, Whh = P["rnn.weight_ih_l0"], P["rnn.weight_hh_l0"]
        bg = P["rnn.bias_ih_l0"] + P["rnn.bias_hh_l0"]
        for t in range(T):
            c["h_prev"][t], c["c_prev"][t] = h, cc
            # encoders.py:83-88
            a_s = relu(xs_pre[t] + h @ Wxs0[:, X:].T)
            lg = lin(a_s, P["encoder.phi_x_s.2.weight"], P["encoder.phi_x_s.2.bias"])
            logits = np.where(lg > 0, lg, dt.type(0.01) * lg)  # LeakyReLU(0.01), encoders.py:57
            probs = softmax(logits)
            s = gumbel_softmax(logits, u[t], temp)
            # encoders.py:91-96
            e1 = relu(ex_pre[t] + s @ We0[:, H:H + K].T + h @ We0[:, H + K:].T)
            e2 = relu(lin(e1, P["encoder.enc.2.weight"], P["encoder.enc.2.bias"]))
            mu = lin(e2, P["encoder.enc_mean.weight"], P["encoder.enc_mean.bias"])
            std = softplus(lin(e2, P["encoder.enc_std.0.weight"], P["encoder.enc_std.0.bias"]))
            z = reparam(mu, std, eps[t])
            # shivae.py:132
            phi_z = relu(lin(z, P["phi_z.0.weight"], P["phi_z.0.bias"]))
            # shivae.py:136 nn.LSTM single step, gate order i,f,g,o
            g4 = phi_z @ Wih.T + h @ Whh.T + bg
            gi, gf = sigmoid(g4[:, :H]), sigmoid(g4[:, H:2 * H])
            gg, go = np.tanh(g4[:, 2 * H:3 * H]), sigmoid(g4[:, 3 * H:])
            cc = gf * cc + gi * gg
            h = go * np.tanh(cc)
            for n, v in (("a_s", a_s), ("logits", logits), ("probs", probs), ("s", s), ("e1", e1),
                         ("e2", e2), ("mu", mu), ("std", std), ("z", z), ("phi_z", phi_z),
                         ("gi", gi), ("gf", gf), ("gg", gg), ("go", go), ("c_new", cc)):
                c[n][t] = v
        c["h_last"], c["c_last"] = h, cc

        # --- prior (shivae.py:124-127), batched over [T,B] ---
        Wp = P["prior.0.weight"]  # [H, H+K], input cat[h, s]
        pr = relu(c["h_prev"] @ Wp[:, :H].T + c["s"] @ Wp[:, H:].T + P["prior.0.bias"])
        mu_p = lin(pr, P["prior_mean.weight"], P["prior_mean.bias"])
        std_p = softplus(lin(pr, P["prior_std.0.weight"], P["prior_std.0.bias"]))
        c.update(pr=pr, mu_p=mu_p, std_p=std_p)

        # --- decoder (decoders.py:109-176), batched over [T,B], loop over attributes ---
        emb = np.concatenate([c["phi_z"], c["s"], c["h_prev"]], axis=-1)  # decoders.py:127
        c["emb"] = emb
        dec_params, dec_cache = self._decode(emb)
        c["dec_params"], c["dec_cache"] = dec_params, dec_cache

        # --- losses (base_missvae.py:166-231) ---
        nll_type, nll_attr = heter_nll(x, mask, dec_params, self.types_list)
        klz = kl_q_z(c["mu"], c["std"], mu_p, std_p).sum(axis=(0, 2))
        kls = kl_q_s(c["probs"]).sum(axis=0)
        nll = nll_type["real"] + nll_type["pos"] + nll_type["bin"] + nll_type["cat"]
        beta = dt.type(beta)
        n_elbo = (nll + beta * klz + beta * kls).mean()
        c.update(nll_attr=nll_attr, nll_type_b=nll_type, klz_b=klz, kls_b=kls)
        c["losses"] = dict(n_elbo=n_elbo, nll_loss=nll.mean(), kld=klz.mean() + kls.mean(),
                           kld_q_z=klz.mean(), kld_q_s=kls.mean(),
                           nll_real=nll_type["real"].mean(), nll_pos=nll_type["pos"].mean(),
                           nll_bin=nll_type["bin"].mean(), nll_cat=nll_type["cat"].mean())
        c.update(x=x, mask=mask, u=u, eps=eps, beta=beta, temp=temp)
        self.cache = c
        return c

    def _decode(self, emb):
        """decoders.py:109-176 on emb = cat[phi_z, s, h_prev]: (per-attribute parameters, activations)."""
        P, dt = self.P, self.dt
        dec_params, dec_cache = [], []
        for t_ in self.types_list:
            pre = "dec.decoder_dict.%s." % t_["name"]
            a0 = relu(lin(emb, P[pre + "dec.0.weight"], P[pre + "dec.0.bias"]))
            a1 = relu(lin(a0, P[pre + "dec.2.weight"], P[pre + "dec.2.bias"]))
            a2 = relu(lin(a1, P[pre + "dec.4.weight"], P[pre + "dec.4.bias"]))
            dc = dict(a0=a0, a1=a1, a2=a2)
            ty = t_["type"]
            if ty in ("real", "pos"):
                m = lin(a2, P[pre + "dec_mean.weight"], P[pre + "dec_mean.bias"])[..., 0]
                if self.learn_std:
                    sd = softplus(lin(a2, P[pre + "dec_std.0.weight"], P[pre + "dec_std.0.bias"]))[..., 0]
                else:
                    sd = np.full_like(m, FIXED_STD)  # decoders.py:143,153
                dec_params.append((m, sd))
            elif ty == "bin":
                hh = relu(lin(a2, P[pre + "dec_p.0.weight"], P[pre + "dec_p.0.bias"]))
                dc["hh"] = hh
                dec_params.append(lin(hh, P[pre + "dec_p.2.weight"], P[pre + "dec_p.2.bias"])[..., 0])
            elif ty == "cat":
                hh = relu(lin(a2, P[pre + "dec_p_cat.0.weight"], P[pre + "dec_p_cat.0.bias"]))
                dc["hh"] = hh
                th = lin(hh, P[pre + "dec_p_cat.2.weight"], P[pre + "dec_p_cat.2.bias"])
                dec_params.append(np.concatenate([np.zeros(th.shape[:-1] + (1,), dtype=dt), th], axis=-1))
            else:
                raise ValueError(ty)
            dec_cache.append(dc)
        return dec_params, dec_cache

    def sample_from_prior(self, u, eps, temp=0.02):
        """shivae.py:156-214 with explicit noise u[T,B,K], eps[T,B,Z]: dict z, mean, std (prior), s, s_probs,
        dec_params."""
        P, dt = self.P, self.dt
        u, eps = np.asarray(u, dtype=dt), np.asarray(eps, dtype=dt)
        T, B, K = u.shape
        H, Z = self.H, self.Z
        temp = dt.type(temp)
        h, cc = np.zeros((B, H), dtype=dt), np.zeros((B, H), dtype=dt)           # :170-171
        Wp = P["prior.0.weight"]
        Wih, Whh = P["rnn.weight_ih_l0"], P["rnn.weight_hh_l0"]
        bg = P["rnn.bias_ih_l0"] + P["rnn.bias_hh_l0"]
        out = {k: np.zeros((T, B, d), dtype=dt) for k, d in (("z", Z), ("mean", Z), ("std", Z), ("s", K), ("s_probs", K),
                                                              ("phi_z", H), ("h_prev", H))}
        logp = np.log(np.full((B, K), 1.0 / K, dtype=dt))                        # :179-180
        for t in range(T):
            s = gumbel_softmax(logp, u[t], temp)
            pr = relu(h @ Wp[:, :H].T + s @ Wp[:, H:].T + P["prior.0.bias"])     # :184
            mean = lin(pr, P["prior_mean.weight"], P["prior_mean.bias"])
            std = softplus(lin(pr, P["prior_std.0.weight"], P["prior_std.0.bias"]))
            z = reparam(mean, std, eps[t])                                      # :188
            phi_z = relu(lin(z, P["phi_z.0.weight"], P["phi_z.0.bias"]))         # :194
            for k, v in (("z", z), ("mean", mean), ("std", std), ("s", s), ("phi_z", phi_z), ("h_prev", h)):
                out[k][t] = v
            out["s_probs"][t] = 1.0 / K
            g4 = phi_z @ Wih.T + h @ Whh.T + bg                                  # :198-199
            gi, gf = sigmoid(g4[:, :H]), sigmoid(g4[:, H:2 * H])
            gg, go = np.tanh(g4[:, 2 * H:3 * H]), sigmoid(g4[:, 3 * H:])
            cc = gf * cc + gi * gg
            h = go * np.tanh(cc)
        emb = np.concatenate([out["phi_z"], out["s"], out["h_prev"]], axis=-1)   # :195 dec(phi_z, s, h[-1])
        out["dec_params"], _ = self._decode(emb)
        return out

    # ---------------------------------------------------------------- backward
    def backward(self, grad_scale=None):
        """Gradient of ``n_elbo`` wrt every parameter (what ``n_elbo.backward()`` gives at
        trainers.py:215), hand-derived.  ``grad_scale`` defaults to 1/B (the batch mean of
        base_missvae.py:212); data-parallel callers pass 1/B_global."""
        c, P, dt = self.cache, self.P, self.dt
        x, mask, eps = c["x"], c["mask"], c["eps"]
        T, B, X = x.shape
        H, Z, K = self.H, self.Z, self.K
        gs = dt.type(1.0 / B if grad_scale is None else grad_scale)
        beta, temp = c["beta"], c["temp"]
        G = {k: np.zeros_like(v) for k, v in P.items()}

        def acc_lin(name_w, name_b, d_out, a_in):
            """dW += d_out^T a_in ; db += sum d_out (over all leading dims)."""
            do2 = d_out.reshape(-1, d_out.shape[-1])
            ai2 = a_in.reshape(-1, a_in.shape[-1])
            G[name_w] += do2.T @ ai2
            if name_b is not None:
                G[name_b] += do2.sum(axis=0)

        # ---- loss -> decoder params, z stats, probs ----
        d_emb = np.zeros_like(c["emb"])
        for i, t_ in enumerate(self.types_list):
            ty, lo, dim = t_["type"], self.tidx[i], int(t_["dim"])
            pre = "dec.decoder_dict.%s." % t_["name"]
            dc = c["dec_cache"][i]
            obs = (~attr_missing(ty, mask[:, :, lo:lo + dim])).astype(dt) * gs
            g = attr_nll_grad(ty, x[:, :, lo:lo + dim], c["dec_params"][i])
            a2 = dc["a2"]
            if ty in ("real", "pos"):
                dm = (g[0] * obs)[..., None]
                acc_lin(pre + "dec_mean.weight", pre + "dec_mean.bias", dm, a2)
                da2 = dm @ P[pre + "dec_mean.weight"]
                if self.learn_std:
                    dsp = (g[1] * obs * softplus_grad_from_out(c["dec_params"][i][1]))[..., None]
                    acc_lin(pre + "dec_std.0.weight", pre + "dec_std.0.bias", dsp, a2)
                    da2 = da2 + dsp @ P[pre + "dec_std.0.weight"]
            else:
                if ty == "bin":
                    w0, w2 = pre + "dec_p.0.", pre + "dec_p.2."
                    dl = (g * obs)[..., None]
                else:
                    w0, w2 = pre + "dec_p_cat.0.", pre + "dec_p_cat.2."
                    dl = (g * obs[..., None])[..., 1:]  # first logit is a constant 0 (decoders.py:169-170)
                acc_lin(w2 + "weight", w2 + "bias", dl, dc["hh"])
                dhh = (dl @ P[w2 + "weight"]) * (dc["hh"] > 0)
                acc_lin(w0 + "weight", w0 + "bias", dhh, a2)
                da2 = dhh @ P[w0 + "weight"]
            da2 = da2 * (a2 > 0)
            acc_lin(pre + "dec.4.weight", pre + "dec.4.bias", da2, dc["a1"])
            da1 = (da2 @ P[pre + "dec.4.weight"]) * (dc["a1"] > 0)
            acc_lin(pre + "dec.2.weight", pre + "dec.2.bias", da1, dc["a0"])
            da0 = (da1 @ P[pre + "dec.2.weight"]) * (dc["a0"] > 0)
            acc_lin(pre + "dec.0.weight", pre + "dec.0.bias", da0, c["emb"])
            d_emb += da0 @ P[pre + "dec.0.weight"]
        dphi_z_ext = d_emb[..., :H]
        ds_ext = d_emb[..., H:H + K].copy()
        dh_ext = d_emb[..., H + K:].copy()

        dmq, dsq, dmp, dsp_ = kl_q_z_grad(c["mu"], c["std"], c["mu_p"], c["std_p"])
        kz = beta * gs
        dmu_ext, dstd_ext = dmq * kz, dsq * kz
        dprobs = kl_q_s_grad(c["probs"]) * kz

        # ---- prior backward ----
        dmu_p = dmp * kz
        dstdp_pre = dsp_ * kz * softplus_grad_from_out(c["std_p"])
        pr = c["pr"]
        acc_lin("prior_mean.weight", "prior_mean.bias", dmu_p, pr)
        acc_lin("prior_std.0.weight", "prior_std.0.bias", dstdp_pre, pr)
        dpr = (dmu_p @ P["prior_mean.weight"] + dstdp_pre @ P["prior_std.0.weight"]) * (pr > 0)
        acc_lin("prior.0.weight", "prior.0.bias", dpr, np.concatenate([c["h_prev"], c["s"]], axis=-1))
        Wp = P["prior.0.weight"]
        dh_ext += dpr @ Wp[:, :H]
        ds_ext += dpr @ Wp[:, H:]

        # ---- reverse-time scan (BPTT over the whole T: shivae.py:102-137 never detaches) ----
        Wxs0, We0 = P["encoder.phi_x_s.0.weight"], P["encoder.enc.0.weight"]
        Wih, Whh = P["rnn.weight_ih_l0"], P["rnn.weight_hh_l0"]
        dh = np.zeros((B, H), dtype=dt)
        dcell = np.zeros((B, H), dtype=dt)
        d_gates = np.zeros((T, B, 4 * H), dtype=dt)
        d_phiz_pre = np.zeros((T, B, H), dtype=dt)
        d_mu = np.zeros((T, B, Z), dtype=dt)
        d_std_pre = np.zeros((T, B, Z), dtype=dt)
        d_e2_pre = np.zeros((T, B, H), dtype=dt)
        d_e1_pre = np.zeros((T, B, H), dtype=dt)
        d_lg_pre = np.zeros((T, B, K), dtype=dt)
        d_as_pre = np.zeros((T, B, H), dtype=dt)
        for t in range(T - 1, -1, -1):
            gi, gf, gg, go = c["gi"][t], c["gf"][t], c["gg"][t], c["go"][t]
            tc = np.tanh(c["c_new"][t])
            do = dh * tc
            dct = dcell + dh * go * (1.0 - tc * tc)
            dgt = np.concatenate([dct * gg * gi * (1 - gi), dct * c["c_prev"][t] * gf * (1 - gf),
                                  dct * gi * (1 - gg * gg), do * go * (1 - go)], axis=-1)
            dcell = dct * gf
            dhp = dgt @ Whh
            dphz = (dgt @ Wih + dphi_z_ext[t]) * (c["phi_z"][t] > 0)
            dz = dphz @ P["phi_z.0.weight"]
            dmu = dz + dmu_ext[t]
            dsd = (dz * eps[t] + dstd_ext[t]) * softplus_grad_from_out(c["std"][t])
            de2 = (dmu @ P["encoder.enc_mean.weight"] + dsd @ P["encoder.enc_std.0.weight"]) * (c["e2"][t] > 0)
            de1 = (de2 @ P["encoder.enc.2.weight"]) * (c["e1"][t] > 0)
            ds = de1 @ We0[:, H:H + K] + ds_ext[t]
            dhp += de1 @ We0[:, H + K:]
            s, pb = c["s"][t], c["probs"][t]
            dy = s * (ds - (s * ds).sum(axis=-1, keepdims=True))
            dlg = dy / temp + pb * (dprobs[t] - (pb * dprobs[t]).sum(axis=-1, keepdims=True))
            dlg = dlg * np.where(c["logits"][t] > 0, 1.0, 0.01).astype(dt)
            das = (dlg @ P["encoder.phi_x_s.2.weight"]) * (c["a_s"][t] > 0)
            dhp += das @ Wxs0[:, X:]
            dh = dhp + dh_ext[t]
            d_gates[t], d_phiz_pre[t], d_mu[t], d_std_pre[t] = dgt, dphz, dmu, dsd
            d_e2_pre[t], d_e1_pre[t], d_lg_pre[t], d_as_pre[t] = de2, de1, dlg, das
        c["scan_bwd"] = dict(d_gates=d_gates, d_phiz_pre=d_phiz_pre, d_mu=d_mu, d_std_pre=d_std_pre,
                             d_e2_pre=d_e2_pre, d_e1_pre=d_e1_pre, d_lg_pre=d_lg_pre, d_as_pre=d_as_pre,
                             dphi_z_ext=dphi_z_ext, ds_ext=ds_ext, dh_ext=dh_ext, dmu_ext=dmu_ext,
                             dstd_ext=dstd_ext, dprobs=dprobs)

        # ---- weight gradients of the recurrent cell (batched outer products) ----
        acc_lin("rnn.weight_ih_l0", "rnn.bias_ih_l0", d_gates, c["phi_z"])
        acc_lin("rnn.weight_hh_l0", "rnn.bias_hh_l0", d_gates, c["h_prev"])
        acc_lin("phi_z.0.weight", "phi_z.0.bias", d_phiz_pre, c["z"])
        acc_lin("encoder.enc_mean.weight", "encoder.enc_mean.bias", d_mu, c["e2"])
        acc_lin("encoder.enc_std.0.weight", "encoder.enc_std.0.bias", d_std_pre, c["e2"])
        acc_lin("encoder.enc.2.weight", "encoder.enc.2.bias", d_e2_pre, c["e1"])
        acc_lin("encoder.enc.0.weight", "encoder.enc.0.bias", d_e1_pre,
                np.concatenate([c["phi_x"], c["s"], c["h_prev"]], axis=-1))
        acc_lin("encoder.phi_x_s.2.weight", "encoder.phi_x_s.2.bias", d_lg_pre, c["a_s"])
        acc_lin("encoder.phi_x_s.0.weight", "encoder.phi_x_s.0.bias", d_as_pre,
                np.concatenate([x, c["h_prev"]], axis=-1))
        # ---- hoisted phi_x backward ----
        dphx = (d_e1_pre @ We0[:, :H]) * (c["phi_x"] > 0)
        acc_lin("encoder.phi_x.2.weight", "encoder.phi_x.2.bias", dphx, c["p1"])
        dp1 = (dphx @ P["encoder.phi_x.2.weight"]) * (c["p1"] > 0)
        acc_lin("encoder.phi_x.0.weight", "encoder.phi_x.0.bias", dp1, x)
        self.grads = G
        return G

    # ---------------------------------------------------------------- imputation
    def reconstruction_params(self, x, u, eps, temp=0.01):
        """base_missvae.py:345-359 reconstruction -> likelihood parameters (the reference
        then draws 100 device-RNG samples, :378-404; parity is on the parameters and the
        Monte-Carlo-free means)."""
        c = self.forward(x, np.zeros(x.shape, dtype=bool), u, eps, beta=1.0, temp=temp)
        return c["dec_params"], c


def mc_free_imputation(dec_params, types_list, dtype=np.float64):
    """Monte-Carlo-free counterpart of generate_heter_samples (base_missvae.py:378-404):
    real/pos -> mean; bin -> round(sigmoid(l)); cat -> one-hot(argmax).  [T,B,X]."""
    cols = []
    for t_, p in zip(types_list, dec_params):
        ty = t_["type"]
        if ty in ("real", "pos"):
            cols.append(np.asarray(p[0], dtype=dtype)[..., None])
        elif ty == "bin":
            cols.append((np.asarray(p) > 0).astype(dtype)[..., None])
        else:
            oh = np.zeros(p.shape, dtype=dtype)
            np.put_along_axis(oh, p.argmax(axis=-1)[..., None], 1.0, axis=-1)
            cols.append(oh)
    return np.concatenate(cols, axis=-1)


# --------------------------------------------------------------------------------------
# optimiser step (trainers.py:222-224: clip_grad_norm_(params, clip) then Adam.step())
# --------------------------------------------------------------------------------------
def clip_adam_step(params, grads, m, v, step, lr=5e-3, clip=0.5, b1=0.9, b2=0.999, adam_eps=1e-8):
    """torch.nn.utils.clip_grad_norm_ (coef = clip/(norm+1e-6), clamped to 1) followed by
    torch.optim.Adam (no weight decay, no amsgrad).  ``step`` is the 1-based step count.
    Operates on dicts of arrays in place; returns the total grad norm."""
    tot = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    coef = min(clip / (tot + 1e-6), 1.0)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    for k in params:
        g = grads[k] * coef
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        denom = np.sqrt(v[k]) / np.sqrt(bc2) + adam_eps
        params[k] = params[k] - (lr / bc1) * m[k] / denom
    return tot


def init_params(types_list, h_dim, z_dim, s_dim, learn_std, seed=0, dtype=np.float32):
    """Parameter dict with the reference's key names and shapes and its
    init *scheme* (base_missvae.py:63-85: LSTM weights orthogonal, other weights
    Xavier-normal, biases zero).  The RNG stream differs from torch's, so parity tests load
    the golden state_dict instead; this is for synthetic benchmarks and CPU tests."""
    rng = np.random.default_rng(seed)
    H, Z, K, X = h_dim, z_dim, s_dim, x_dim(types_list)
    P = {}

    def linear(name, n_out, n_in):
        std = np.sqrt(2.0 / (n_in + n_out))
        P[name + ".weight"] = (rng.standard_normal((n_out, n_in)) * std).astype(dtype)
        P[name + ".bias"] = np.zeros(n_out, dtype=dtype)

    linear("encoder.phi_x.0", H, X); linear("encoder.phi_x.2", H, H)
    linear("encoder.phi_x_s.0", H, X + H); linear("encoder.phi_x_s.2", K, H)
    linear("encoder.enc.0", H, 2 * H + K); linear("encoder.enc.2", H, H)
    linear("encoder.enc_mean", Z, H); linear("encoder.enc_std.0", Z, H)
    linear("phi_z.0", H, Z)
    linear("prior.0", H, H + K); linear("prior_mean", Z, H); linear("prior_std.0", Z, H)
    for t_ in types_list:
        pre = "dec.decoder_dict.%s." % t_["name"]
        linear(pre + "dec.0", 2 * H, 2 * H + K); linear(pre + "dec.2", H, 2 * H); linear(pre + "dec.4", H, H)
        ty = t_["type"]
        if ty in ("real", "pos"):
            linear(pre + "dec_mean", 1, H)
            if learn_std:
                linear(pre + "dec_std.0", 1, H)
        elif ty == "bin":
            linear(pre + "dec_p.0", H, H); linear(pre + "dec_p.2", 1, H)
        elif ty == "cat":
            linear(pre + "dec_p_cat.0", H, H); linear(pre + "dec_p_cat.2", int(t_["nclass"]) - 1, H)
    for nm in ("rnn.weight_ih_l0", "rnn.weight_hh_l0"):
        a = rng.standard_normal((4 * H, H))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        P[nm] = q.astype(dtype)
    P["rnn.bias_ih_l0"] = np.zeros(4 * H, dtype=dtype)
    P["rnn.bias_hh_l0"] = np.zeros(4 * H, dtype=dtype)
    return P
