"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product path).

Stock PyTorch-CPU restatement of the reference's train step, arranged as the reference arranges it: a Python loop
over the T timesteps (src/models/shivae.py:102-143), inside it the encoder (src/models/encoders.py:70-98), the prior
(shivae.py:124-127), phi_z, a Python loop over the D attribute decoders (src/models/decoders.py:109-176) and the
LSTM cell; then the masked heterogeneous likelihood, the two KL terms and the batch mean
(src/models/base_missvae.py:166-343), autograd backward, clip_grad_norm_ and Adam (src/models/trainers.py:215-224).
Differences from the reference: the noise is explicit (u, eps) instead of drawn inside the loop, the likelihoods are
written out instead of going through torch.distributions (same formulas), parameters come from a state_dict.

Used for SURVEY.md 8d's like-for-like "reference CPU path" figure in bench.py's cpu_baseline (kind "port") and
pinned against the reference's golden vectors in tests/test_oracle_golden.py."""
import math

import torch
import torch.nn.functional as F

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
FIXED_STD = 0.1
F32_EPS = 1.1920928955078125e-07


def make_params(state_dict):
    """name -> leaf tensor requiring grad (float32 copies)."""
    return {k: torch.as_tensor(v, dtype=torch.float32).clone().requires_grad_(True) for k, v in state_dict.items()}


def forward(P, types_list, H, Z, K, x, mask, u, eps, beta=1.0, temp=3.0, learn_std=True):
    """-> dict of the 9 loss scalars (tensors; n_elbo carries the graph)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    mask = torch.as_tensor(mask, dtype=torch.bool)
    u = torch.as_tensor(u, dtype=torch.float32)
    eps = torch.as_tensor(eps, dtype=torch.float32)
    T, B, X = x.shape
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    lin = F.linear
    mus, sds, mups, sdps, probs_l = [], [], [], [], []
    dec = {t["name"]: [] for t in types_list}
    for t in range(T):
        xt = x[t]
        # encoder (encoders.py:82-96)
        phi_x = F.relu(lin(F.relu(lin(xt, P["encoder.phi_x.0.weight"], P["encoder.phi_x.0.bias"])),
                           P["encoder.phi_x.2.weight"], P["encoder.phi_x.2.bias"]))
        a_s = F.relu(lin(torch.cat([xt, h], -1), P["encoder.phi_x_s.0.weight"], P["encoder.phi_x_s.0.bias"]))
        logits = F.leaky_relu(lin(a_s, P["encoder.phi_x_s.2.weight"], P["encoder.phi_x_s.2.bias"]), 0.01)
        s_probs = torch.softmax(logits, -1)
        g = -torch.log(-torch.log(u[t] + 1e-20) + 1e-20)                 # samplers.py:23-26
        s = torch.softmax((logits + g) / temp, -1)
        e = F.relu(lin(torch.cat([phi_x, s, h], -1), P["encoder.enc.0.weight"], P["encoder.enc.0.bias"]))
        e = F.relu(lin(e, P["encoder.enc.2.weight"], P["encoder.enc.2.bias"]))
        mu = lin(e, P["encoder.enc_mean.weight"], P["encoder.enc_mean.bias"])
        sd = F.softplus(lin(e, P["encoder.enc_std.0.weight"], P["encoder.enc_std.0.bias"]))
        z = eps[t] * sd + mu                                             # samplers.py:57-62
        # prior (shivae.py:124-127)
        pr = F.relu(lin(torch.cat([h, s], -1), P["prior.0.weight"], P["prior.0.bias"]))
        mups.append(lin(pr, P["prior_mean.weight"], P["prior_mean.bias"]))
        sdps.append(F.softplus(lin(pr, P["prior_std.0.weight"], P["prior_std.0.bias"])))
        phi_z = F.relu(lin(z, P["phi_z.0.weight"], P["phi_z.0.bias"]))
        # decoders (decoders.py:127-171), one attribute after the other
        emb = torch.cat([phi_z, s, h], -1)
        for td in types_list:
            pre = "dec.decoder_dict.%s." % td["name"]
            a = F.relu(lin(emb, P[pre + "dec.0.weight"], P[pre + "dec.0.bias"]))
            a = F.relu(lin(a, P[pre + "dec.2.weight"], P[pre + "dec.2.bias"]))
            a = F.relu(lin(a, P[pre + "dec.4.weight"], P[pre + "dec.4.bias"]))
            ty = td["type"]
            if ty in ("real", "pos"):
                m = lin(a, P[pre + "dec_mean.weight"], P[pre + "dec_mean.bias"])
                sdd = F.softplus(lin(a, P[pre + "dec_std.0.weight"], P[pre + "dec_std.0.bias"])) if learn_std \
                    else torch.full_like(m, FIXED_STD)
                dec[td["name"]].append((m, sdd))
            elif ty == "bin":
                hh = F.relu(lin(a, P[pre + "dec_p.0.weight"], P[pre + "dec_p.0.bias"]))
                dec[td["name"]].append(lin(hh, P[pre + "dec_p.2.weight"], P[pre + "dec_p.2.bias"]))
            else:
                hh = F.relu(lin(a, P[pre + "dec_p_cat.0.weight"], P[pre + "dec_p_cat.0.bias"]))
                th = lin(hh, P[pre + "dec_p_cat.2.weight"], P[pre + "dec_p_cat.2.bias"])
                dec[td["name"]].append(torch.cat([torch.zeros_like(th[:, :1]), th], -1))
        # nn.LSTM, one layer, one step (shivae.py:136)
        g4 = lin(phi_z, P["rnn.weight_ih_l0"], P["rnn.bias_ih_l0"]) + lin(h, P["rnn.weight_hh_l0"], P["rnn.bias_hh_l0"])
        gi, gf, gg, go = g4.chunk(4, -1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        mus.append(mu); sds.append(sd); probs_l.append(s_probs)
    mu, sd, mup, sdp, probs = (torch.stack(v) for v in (mus, sds, mups, sdps, probs_l))
    # masked heterogeneous likelihood (base_missvae.py:233-285)
    nll = {k: torch.zeros(B) for k in ("real", "pos", "bin", "cat")}
    lo = 0
    for td in types_list:
        ty, dim = td["type"], int(td["dim"])
        xd, md = x[:, :, lo:lo + dim], mask[:, :, lo + dim - 1]
        if ty in ("real", "pos"):
            m = torch.stack([p[0] for p in dec[td["name"]]])[..., 0]
            s_ = torch.stack([p[1] for p in dec[td["name"]]])[..., 0]
            v = (xd[..., 0] - m) ** 2 / (2 * s_ ** 2) + torch.log(s_) + HALF_LOG_2PI
        elif ty == "bin":
            lg = torch.stack(dec[td["name"]])[..., 0]
            v = F.softplus(lg) - xd[..., 0] * lg
        else:
            th = torch.stack(dec[td["name"]])
            v = torch.logsumexp(th, -1) - th.gather(-1, xd.argmax(-1, keepdim=True))[..., 0]
        nll[ty] = nll[ty] + torch.where(md, torch.zeros_like(v), v).sum(0)
        lo += dim
    # KL terms (base_missvae.py:287-343)
    sq, sp_ = sd.clamp(1e-6, 1e20), sdp.clamp(1e-6, 1e20)
    r = (sq / sp_) ** 2
    klz = (0.5 * (r + ((mu - mup) / sp_) ** 2 - 1 - torch.log(r))).sum(dim=(0, 2))
    p = probs / probs.sum(-1, keepdim=True)
    logp = torch.log(p.clamp(F32_EPS, 1 - F32_EPS))
    kls = (p * (logp - math.log(min(max(1.0 / K, F32_EPS), 1 - F32_EPS)))).sum(-1).sum(0)
    nll_loss = nll["real"] + nll["pos"] + nll["bin"] + nll["cat"]
    n_elbo = (nll_loss + beta * klz + beta * kls).mean()
    return dict(n_elbo=n_elbo, nll_loss=nll_loss.mean(), kld=klz.mean() + kls.mean(), kld_q_z=klz.mean(), kld_q_s=kls.mean(),
                nll_real=nll["real"].mean(), nll_pos=nll["pos"].mean(), nll_bin=nll["bin"].mean(), nll_cat=nll["cat"].mean())


def train_step(P, opt, types_list, H, Z, K, x, mask, u, eps, beta=1.0, temp=3.0, learn_std=True, clip=0.5):
    """trainers.py:209-224: zero_grad, forward, backward, clip_grad_norm_, Adam step.  -> (losses, grad norm)"""
    opt.zero_grad()
    losses = forward(P, types_list, H, Z, K, x, mask, u, eps, beta, temp, learn_std)
    losses["n_elbo"].backward()
    gn = torch.nn.utils.clip_grad_norm_(list(P.values()), clip)
    opt.step()
    return losses, float(gn)
