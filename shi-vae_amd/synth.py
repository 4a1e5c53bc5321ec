"""Device-side synthetic heterogeneous-HMM batches for benchmarks (shape + statistics of
src/hmm_dataset/heterHMMdataset.py: K-state Markov chain :76-94, per-state Gaussian / softplus-Gaussian /
Bernoulli / dominant-class categorical emissions :127-249, burst missing masks :292-363), already
scaled (standardised real, log1p-standardised pos), one-hot expanded and zero-filled, time-major."""
import math

import torch


def synth_batch(types_list, T, B, miss=0.3, n_states=3, seed=1234, device="cuda", chunk=512, seq0=0):
    """Sequences [seq0, seq0 + B) of the synthetic set `seed`.  Every block of `chunk` sequences has a generator
    stream of its own (seeded by its block index), so a rank that asks for its shard of a global batch gets exactly
    the rows a single process would have generated there (seq0 must sit on a block boundary)."""
    assert seq0 % chunk == 0, "a shard starts on a block of %d sequences" % chunk
    g = torch.Generator(device=device)
    x_is_cuda = torch.device(device).type == "cuda"
    xs, ms = [], []
    for b0 in range(0, B, chunk):
        b = min(chunk, B - b0)
        g.manual_seed(int(seed) * 1000003 + (seq0 + b0) // chunk)
        # hidden chain: stay with prob 0.9, else jump to a uniformly random state.  No loop over t: the state at t
        # is the state drawn at the last jump at or before t (t = 0 always draws).
        jump = torch.rand(T, b, generator=g, device=device) < 0.1
        jump[0] = True
        drawn = torch.randint(0, n_states, (T, b), generator=g, device=device)
        tt = torch.arange(T, device=device).unsqueeze(1).expand(T, b)
        last = torch.where(jump, tt, torch.zeros_like(tt)).cummax(0).values
        st = drawn.gather(0, last)
        cols, mcols = [], []
        for d, ty in enumerate(types_list):
            kind, dim = ty["type"], int(ty["dim"])
            if kind in ("real", "pos"):
                means = torch.linspace(-20, 20, n_states, device=device).roll(d % n_states)
                v = means[st] + 0.5 * torch.randn(T, b, generator=g, device=device)
                if kind == "pos":
                    v = torch.log1p(torch.nn.functional.softplus(v))
                v = (v - v.mean()) / (v.std() + 1e-6)
                cols.append(v.unsqueeze(-1))
            elif kind == "bin":
                cols.append((((st + d) % 2) == 0).float().unsqueeze(-1))
            else:
                dom = (st + d) % dim
                other = torch.randint(0, dim, (T, b), generator=g, device=device)
                keep = torch.rand(T, b, generator=g, device=device) < 0.9
                cols.append(torch.nn.functional.one_hot(torch.where(keep, dom, other), dim).float())
            # bursts of length U(3,10) until ~round(T*miss) steps are masked (heterHMMdataset.py:292-363 draws bursts
            # until that count is reached; here the number of bursts is fixed so that, overlaps included, the
            # expected masked fraction is `miss`: 1 - exp(-nb * 6.5 / T) = miss)
            nb = max(1, int(round(-math.log(max(1.0 - miss, 1e-6)) * T / 6.5)))
            start = torch.randint(0, T, (nb, b), generator=g, device=device)
            length = torch.randint(3, 11, (nb, b), generator=g, device=device)
            diff = torch.zeros(T + 1, b, device=device)
            diff.scatter_add_(0, start, torch.ones(nb, b, device=device))
            diff.scatter_add_(0, torch.clamp(start + length, max=T), -torch.ones(nb, b, device=device))
            m = diff.cumsum(0)[:T] > 0
            mcols.append(m.unsqueeze(-1).expand(T, b, dim))
            if x_is_cuda:
                # keep the launch queue shallow (a dozen small kernels per attribute, tens of thousands per batch):
                # `rocprofv3 --pmc` does not survive thousands of dispatches enqueued ahead of the GPU
                torch.cuda.synchronize()
        x = torch.cat(cols, -1)
        m = torch.cat(mcols, -1)
        xs.append(torch.where(m, torch.zeros_like(x), x))
        ms.append(m)
    return torch.cat(xs, 1).contiguous(), torch.cat(ms, 1).contiguous()
