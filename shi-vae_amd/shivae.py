"""Shi-VAE model with the reference's construction / forward / imputation API, running its train
step on the hand-written HIP kernels of libshivae_hip.so.

Mirrors ``src/models/shivae.py`` (ShiVAE) and ``src/models/base_missvae.py`` (BaseMissVAE):
same constructor, ``model(x, mask, beta, temp) -> loss dict``, ``feed2model``, ``compute_loss``,
``heter_nll``, ``kl_q_z``, ``kl_q_s``, ``reconstruction``, ``generate_heter_samples`` and the same
``state_dict`` keys, so reference checkpoints load and ``torch.optim.Adam(model.parameters())``
works.  PyTorch is used for device memory, streams and torch.distributed only; there is no eager
fallback: without the HIP library (or without a GPU) the step raises.

Differences that are deliberate (and documented in DESIGN.md):
  * all parameters are views into ONE flat fp32 buffer (and gradients into one flat buffer), laid
    out by the native plan so that per-attribute decoder weights are stacked for grouped GEMMs;
  * forward+backward are fused: ``model(...)`` in training mode already computes the gradients,
    ``loss['n_elbo'].backward()`` publishes them to ``p.grad``;
  * noise: by default the scans draw it in-kernel from a counter-based generator (``noise_mode='device'``:
    csrc/noise.h, Philox4x32-10, seeded from torch's CPU generator; no noise tensor exists);
    ``noise_mode='torch'`` draws tensors with torch.rand / torch.randn on the device;
    ``noise_mode='reference'`` replays the reference's CPU stream (torch.rand(B,K) then normal_(B,Z)
    per timestep, samplers.py:25,60) so that ``torch.manual_seed`` reproduces the reference exactly;
    ``set_noise(u, eps)`` injects tensors for the next call.
"""
import os

import torch
import torch.nn as nn

from . import _native as N
from . import utils

print_shivae = "ELBO:{n_elbo:.3f} | NLL:{nll_loss:.3f}  KL_Q_Z:{kld_q_z:.3f}  KL_Q_S:{kld_q_s:.3f}|  " \
               "Real:{nll_real:.3f}  Pos:{nll_pos:.3f}  Bin:{nll_bin:.3f}  Cat:{nll_cat:.3f}"
LOSSES = ["real", "pos", "bin", "cat"]
LOSS_KEYS = ["n_elbo", "nll_loss", "kld", "kld_q_z", "kld_q_s", "nll_real", "nll_pos", "nll_bin", "nll_cat"]
FIXED_STD = 0.1  # decoders.py:22


def set_device():
    """lib/aux.py:37-47."""
    return "cuda" if torch.cuda.is_available() else "cpu"


class _Holder(nn.Module):
    """Plain container; only gives parameters their reference names."""


class _AttachGrads(torch.autograd.Function):
    """Gives n_elbo a grad_fn.  The gradients were already computed by the fused step; backward
    publishes them to ``p.grad`` (scaled by the incoming gradient)."""

    @staticmethod
    def forward(ctx, hook, value, model):
        ctx.model = model
        return value.clone()

    @staticmethod
    def backward(ctx, go):
        ctx.model._publish_grads(go)
        return None, None, None


class ShiVAE(nn.Module):
    def __init__(self, h_dim, z_dim, s_dim, n_layers, types_list=None, learn_std=False, activation_layer='ReLU',
                 gain_init=1, device=None):
        super().__init__()
        assert types_list is not None  # base_missvae.py:41
        if str(activation_layer).lower() != 'relu':
            raise NotImplementedError("the HIP Shi-VAE step implements activation_layer='ReLU' (the reference default)")
        if int(n_layers) != 1:
            raise NotImplementedError("the HIP Shi-VAE step implements n_layers=1 (the reference's setting)")
        self.types_list = utils.reindex_types_list(types_list)
        self.transform_idx = utils.get_idx_transform(self.types_list)
        self.device = device or set_device()
        self.learn_std = bool(learn_std)
        self.K = 1  # IWAE / Monte-Carlo replication factors of the reference (always 1 there)
        self.M = 1
        self.x_dim = utils.get_x_dim(self.types_list)
        self.h_dim, self.z_dim, self.s_dim, self.n_layers = int(h_dim), int(z_dim), int(s_dim), 1

        self.noise_mode = 'device'
        self.max_micro_batch = None  # cap on the per-call batch (default: whatever fits the memory budget)
        self.print_losses = False   # the reference prints one line per forward (base_missvae.py:163)
        self.check_nan = True       # base_missvae.py:223-228 (one host sync per forward)
        self.workspace_limit_bytes = int(float(os.environ.get("SHIVAE_WS_GB", "0")) * 1e9) or None
        self.chunk_rows = int(os.environ.get("SHIVAE_CHUNK_ROWS", "0"))
        # rows of the decoder's activation buffers when fewer than the chunk's (shivae_plan_set_activation_rows):
        # `act_rows` fixes them next to `chunk_rows`; otherwise _grow_chunk sizes them from the observed fraction of
        # the first training batch of a shape when `narrow_activations` is on (SHIVAE_NARROW=0 turns it off)
        self.act_rows = int(os.environ.get("SHIVAE_ACT_ROWS", "0"))
        self.narrow_activations = os.environ.get("SHIVAE_NARROW", "1") != "0"
        # data parallel: start the all-reduce of the decoder's gradients (99 % of the buffer) as soon as the last
        # decoder row chunk is done, in buckets, while the reverse-time scan and the cell's weight gradients still
        # run.  Opt-in, because the caller then HAS to call allreduce_grads() (bench.py and Trainer do).
        self.overlap_allreduce = False
        self.bucket_floats = int(float(os.environ.get("SHIVAE_BUCKET_MB", "128")) * (1 << 18))
        # Small models are launch-bound (cfg1 / cfg2: ~125 launches for a 2-3 ms step): a step whose shapes and scalars
        # repeat is captured into a HIP graph on its second occurrence and replayed afterwards.  None = automatic
        # (single micro-batch, single decoder row chunk, at most `graph_max_rows` (t,b) rows), False = never.
        self.use_graphs = None if os.environ.get("SHIVAE_GRAPHS", "auto") == "auto" else os.environ["SHIVAE_GRAPHS"] not in ("0", "off")
        self.graph_max_rows = 1 << 16
        self._graphs = {}
        self._graph_seen = set()
        self._pending_reduce = []
        self._pending_noise = None
        self._plans = {}
        self._chunk_done = set()
        self._active = None
        self._workspace = None
        self._last = None

        self._attr_types = [N.TYPE_CODE[t['type']] for t in self.types_list]
        self._attr_ncls = [int(t['dim']) if t['type'] == 'cat' else 1 for t in self.types_list]
        for t in self.types_list:
            if t['type'] == 'cat':
                assert int(t['dim']) == int(t['nclass']), "categorical dim must equal nclass"
        self._layout = N.Plan(1, 1, self.h_dim, self.z_dim, self.s_dim, self._attr_types, self._attr_ncls,
                              self.learn_std)
        self.p_dim = self._layout.p_dim
        self._n_flat = self._layout.num_params
        dev = torch.device(self.device)
        self._flat = torch.zeros(self._n_flat, dtype=torch.float32, device=dev)
        self._gflat = torch.zeros(self._n_flat, dtype=torch.float32, device=dev)
        self._hook = torch.zeros(1, device=dev, requires_grad=True)
        self._slices = []  # (parameter, offset, numel, shape) in registration order
        self._build_parameters()
        self.init_weights(gain=gain_init)

    # ------------------------------------------------------------------ parameters
    def _register(self, holder, pname, tensor_id, attr=0, bias=False, row=None):
        info = self._layout.param_info(tensor_id, attr)
        assert info is not None, (pname, tensor_id, attr)
        off, r, c = info
        shape = (c,) if bias else (r, c)
        n = r * c
        p = nn.Parameter(self._flat[off:off + n].view(shape))
        holder.register_parameter(pname, p)
        self._slices.append((p, off, n, shape))

    def _linear(self, parent, name, wid, attr=0):
        h = _Holder()
        parent.add_module(name, h)
        self._register(h, "weight", wid, attr)
        self._register(h, "bias", wid + 1, attr, bias=True)
        return h

    def _build_parameters(self):
        """Same module tree / registration order as shivae.py:48-68, encoders.py:47-68, decoders.py:54-107."""
        P = N.P
        enc = _Holder(); self.add_module("encoder", enc)
        for name, ids in (("phi_x", ((0, P["PHIX0_W"]), (2, P["PHIX2_W"]))), ("phi_x_s", ((0, P["XS0_W"]), (2, P["XS2_W"]))),
                          ("enc", ((0, P["E0_W"]), (2, P["E2_W"])))):
            seq = _Holder(); enc.add_module(name, seq)
            for idx, wid in ids:
                self._linear(seq, str(idx), wid)
        self._linear(enc, "enc_mean", P["ENCMEAN_W"])
        seq = _Holder(); enc.add_module("enc_std", seq); self._linear(seq, "0", P["ENCSTD_W"])
        seq = _Holder(); self.add_module("phi_z", seq); self._linear(seq, "0", P["PHIZ_W"])
        seq = _Holder(); self.add_module("prior", seq); self._linear(seq, "0", P["PRIOR0_W"])
        self._linear(self, "prior_mean", P["PRIORMEAN_W"])
        seq = _Holder(); self.add_module("prior_std", seq); self._linear(seq, "0", P["PRIORSTD_W"])
        dec = _Holder(); self.add_module("dec", dec)
        dd = _Holder(); dec.add_module("decoder_dict", dd)
        for d, t in enumerate(self.types_list):
            v = _Holder(); dd.add_module(t['name'], v)
            trunk = _Holder(); v.add_module("dec", trunk)
            self._linear(trunk, "0", P["DEC0_W"], d)
            self._linear(trunk, "2", P["DEC2_W"], d)
            self._linear(trunk, "4", P["DEC4_W"], d)
            ty = t['type']
            if ty in ('real', 'pos'):
                self._linear(v, "dec_mean", P["HEADMEAN_W"], d)
                if self.learn_std:
                    s = _Holder(); v.add_module("dec_std", s); self._linear(s, "0", P["HEADSTD_W"], d)
            else:
                s = _Holder(); v.add_module("dec_p" if ty == 'bin' else "dec_p_cat", s)
                self._linear(s, "0", P["HEAD0_W"], d)
                self._linear(s, "2", P["HEADOUT_W"], d)
        rnn = _Holder(); self.add_module("rnn", rnn)
        self._register(rnn, "weight_ih_l0", P["RNN_WIH"])
        self._register(rnn, "weight_hh_l0", P["RNN_WHH"])
        self._register(rnn, "bias_ih_l0", P["RNN_BIH"], bias=True)
        self._register(rnn, "bias_hh_l0", P["RNN_BHH"], bias=True)

    def init_weights(self, gain=1):
        """base_missvae.py:63-85: LSTM weights orthogonal, other weights Xavier-normal, biases 0."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if 'bias' in name:
                    nn.init.constant_(p, 0.0)
                elif name.startswith('rnn.'):
                    nn.init.orthogonal_(p, gain=gain)
                else:
                    nn.init.xavier_normal_(p, gain=gain)

    def get_name(self):
        return self.__class__.__name__

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._reflatten()
        return self

    def _reflatten(self):
        """After .to()/.cuda() the parameters are separate tensors again: re-pack them into one buffer."""
        if not self._slices:
            return
        dev = self._slices[0][0].device
        if self._slices[0][0].data_ptr() == self._flat.data_ptr() + 4 * self._slices[0][1] and self._flat.device == dev:
            return
        flat = torch.zeros(self._n_flat, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off, n, shape in self._slices:
                flat[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + n].view(shape)
                p.grad = None
        self._flat = flat
        self._gflat = torch.zeros_like(flat)
        self._hook = torch.zeros(1, device=dev, requires_grad=True)
        self.device = dev.type
        self._plans, self._active, self._workspace = {}, None, None
        self._graphs, self._graph_seen = {}, set()

    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        return self._gflat

    def zero_grad(self, set_to_none=True):
        self._gflat.zero_()
        if set_to_none:
            for p, _, _, _ in self._slices:
                p.grad = None

    def _publish_grads(self, go=None):
        fresh = self._slices[0][0].grad is None
        if self._pending_reduce:
            for w in self._pending_reduce[0][1]:  # (nothing below may race with buckets in flight, fresh or not)
                w.wait()
        if go is not None and fresh:
            self._gflat.mul_(go.reshape(()))
        for p, off, n, shape in self._slices:
            g = self._gflat[off:off + n].view(shape)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)

    # ------------------------------------------------------------------ plans / workspace
    def _require_gpu(self):
        if self._flat.device.type != 'cuda':
            raise N.ShiVAENativeError("the Shi-VAE step runs only on a GPU (HIP kernels); model is on %s and there "
                                      "is no CPU fallback" % self._flat.device)
        N.lib()

    def _make_plan(self, T, B, chunk_rows=None, act_rows=None):
        key = (T, B)
        if chunk_rows is not None:
            self._plans.pop(key, None)
        if key not in self._plans:
            self._plans[key] = N.Plan(T, B, self.h_dim, self.z_dim, self.s_dim, self._attr_types, self._attr_ncls,
                                      self.learn_std, self.chunk_rows if chunk_rows is None else chunk_rows,
                                      (self.act_rows if chunk_rows is None else 0) if act_rows is None else act_rows)
        return self._plans[key]

    def _grow_chunk(self, T, B, fmax=None):
        """Give the memory a micro-batch leaves unused to the decoder row chunk (fewer, larger launches).
        Workspace is linear in the chunk rows, so two plans give the slope.  fmax: the largest observed fraction of
        any attribute in the training batch, when known: a mask-aware chunk fills its activation buffers only that
        far, so below 0.45 they are sized for 1.3 * fmax of the chunk's rows and the chunk grows by what that frees
        (at 90 % missing: 5x the rows per launch; a chunk that does not fit after all is split by the library)."""
        if self.chunk_rows or (T, B) in self._chunk_done:
            return
        self._chunk_done.add((T, B))
        budget = self._budget(getattr(self, "_slice_reserve", 0))
        p0 = self._make_plan(T, B)
        rc0, ws0 = p0.chunk_rows, p0.workspace_bytes
        if rc0 >= T * B or ws0 >= budget:
            return
        rc1 = min(T * B, 2 * rc0)
        ws1 = self._make_plan(T, B, chunk_rows=rc1).workspace_bytes
        per_row = max(1.0, (ws1 - ws0) / float(rc1 - rc0))
        rc = int(min(T * B, rc0 + 0.9 * (budget - ws0) / per_row)) // 128 * 128
        frac = None if fmax is None or not self.narrow_activations else min(1.0, 1.3 * fmax + 0.01)
        if frac is not None and frac <= 0.6 and rc < T * B:
            # workspace = fixed + (per chunk row) * Rc + (per activation row) * Ra: a third plan separates the slopes
            ra1 = max(16, int(frac * rc1))
            ws1n = self._make_plan(T, B, chunk_rows=rc1, act_rows=ra1).workspace_bytes
            per_act = max(0.0, (ws1 - ws1n) / float(rc1 - ra1))
            per_narrow = max(1.0, per_row - per_act + frac * per_act)
            rcn = int(min(T * B, 8 * max(rc, rc0), (rc0 * per_row + 0.9 * (budget - ws0)) / per_narrow)) // 128 * 128
            ran = min(rcn, (int(frac * rcn) + 127) // 128 * 128)
            if rcn > max(rc, rc0) and self._make_plan(T, B, chunk_rows=rcn, act_rows=ran).workspace_bytes <= budget:
                self._active = None
                return
        self._make_plan(T, B, chunk_rows=max(rc, rc0))
        self._active = None

    def _budget(self, reserve=0):
        """Bytes the workspace may take: what is free now plus what the current workspace holds, minus head-room --
        15 %, or `reserve` (the contiguous x / mask / u / eps slices of a split batch) + 4 GB if that is more."""
        if self.workspace_limit_bytes:
            return self.workspace_limit_bytes
        dev = self._flat.device
        free, _ = torch.cuda.mem_get_info(dev)
        # (+ what torch's caching allocator holds without using it: the driver does not count that as free)
        cached = max(0, torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
        held = self._workspace.numel() if self._workspace is not None else 0
        total = free + cached + held
        return int(total - max(0.15 * total, reserve + 4e9))

    def micro_batch(self, T, B):
        """Largest per-call batch whose workspace fits the budget (callers' B is split into these);
        ``max_micro_batch`` caps it (a multiple of the 16-sequence scan tile)."""
        bm = B if not self.max_micro_batch else min(B, max(16, int(self.max_micro_batch) // 16 * 16))
        # a split batch also needs contiguous copies of each micro-batch's x, mask, u, eps slices
        slices = lambda b: 0 if b >= B else T * b * (5 * self.x_dim + 4 * self.s_dim + 4 * self.z_dim)
        while True:
            budget = self._budget(slices(bm))
            ws = self._make_plan(T, bm).workspace_bytes
            if ws <= budget or bm <= 16:
                return bm
            self._plans.pop((T, bm), None)
            nb = max(16, int(bm * budget / ws * 0.98) // 16 * 16)
            bm = nb if nb < bm else max(16, bm - 16)

    def micro_batches(self, T, B):
        """Per-call batch sizes the step is run in: equal-sized micro-batches (one plan, no re-binding), multiples
        of the 16-sequence scan tile, each fitting the workspace budget."""
        bm = self.micro_batch(T, B)
        if bm < B:
            nmb = -(-B // bm)
            bm = min(bm, -(-(-(-B // nmb)) // 16) * 16)
        return [min(bm, B - b0) for b0 in range(0, B, bm)]

    def _activate(self, T, B):
        plan = self._make_plan(T, B)
        if self._active is plan:
            return plan
        need = plan.workspace_bytes
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        plan.bind(self._workspace, self._flat, self._gflat)
        self._active = plan
        return plan

    # ------------------------------------------------------------------ noise
    def set_noise(self, u, eps):
        """Noise for the NEXT step: u[T,B,K] in [0,1) (Gumbel), eps[T,B,Z] ~ N(0,1)."""
        self._pending_noise = (u, eps)

    def _draw_noise(self, T, B, tensors=False):
        """-> (u, eps, seed).  ``noise_mode='device'`` (default): (None, None, seed) -- the scans draw their noise
        in-kernel from a counter-based generator under `seed` (csrc/noise.h), nothing is materialised; `seed` comes
        from torch's CPU generator, so ``torch.manual_seed`` fixes the run.  ``tensors=True`` asks for the same
        streams as tensors (captured steps replay fixed kernel arguments, so they take their noise from buffers)."""
        dev = self._flat.device
        if self._pending_noise is not None:
            u, eps = self._pending_noise
            self._pending_noise = None
            u = torch.as_tensor(u, dtype=torch.float32, device=dev).contiguous()
            eps = torch.as_tensor(eps, dtype=torch.float32, device=dev).contiguous()
            assert tuple(u.shape) == (T, B, self.s_dim) and tuple(eps.shape) == (T, B, self.z_dim)
            return u, eps, 0
        if self.noise_mode == 'reference':
            u = torch.empty(T, B, self.s_dim)
            eps = torch.empty(T, B, self.z_dim)
            for t in range(T):  # samplers.py:25 then :60, per timestep, CPU generator
                u[t] = torch.rand(B, self.s_dim)
                eps[t] = torch.FloatTensor(B, self.z_dim).normal_()
            return u.to(dev), eps.to(dev), 0
        if self.noise_mode == 'torch':
            return torch.rand(T, B, self.s_dim, device=dev), torch.randn(T, B, self.z_dim, device=dev), 0
        assert self.noise_mode == 'device', "noise_mode is 'device', 'torch' or 'reference'"
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        seed = (seed + 0x9E3779B97F4A7C15 * rank) & (2 ** 64 - 1)  # (ranks under one torch seed draw different noise)
        if tensors:
            u, eps = self.noise_tensors(seed, T, B)
            return u, eps, seed
        return None, None, seed

    def noise_tensors(self, seed, T, B):
        """u[T,B,K], eps[T,B,Z] exactly as the scans draw them in-kernel under `seed` for a (T, B) batch
        (shivae_noise_fill): ``set_noise(*model.noise_tensors(seed, T, B))`` replays a ``noise_mode='device'`` step."""
        dev = self._flat.device
        u = torch.empty(T, B, self.s_dim, device=dev)
        eps = torch.empty(T, B, self.z_dim, device=dev)
        N.check(N.lib().shivae_noise_fill(int(seed), T * B, self.s_dim, self.z_dim, N.ptr(u), N.ptr(eps), N.stream()))
        return u, eps

    @staticmethod
    def _mb_seed(seed, i):
        """seed of micro-batch i of a step (rows are per micro-batch: each takes a stream of its own)"""
        return (seed + 0xD1B54A32D192ED03 * i) & (2 ** 64 - 1)

    # ------------------------------------------------------------------ the step
    def _step(self, x, mask, beta, temp, grad, keep_params=False, inv_batch=None, collect=None):
        """Run the fused step over micro-batches.  collect: list of output ids to gather ([T,B,w])."""
        self._require_gpu()
        dev = self._flat.device
        x = torch.as_tensor(x, dtype=torch.float32, device=dev)
        mask = torch.as_tensor(mask, device=dev)
        if mask.dtype != torch.bool:
            mask = mask != 0
        T, B, X = x.shape
        assert X == self.x_dim, "x has %d columns, types_list implies %d" % (X, self.x_dim)
        inv_batch = 1.0 / B if inv_batch is None else float(inv_batch)
        if grad and self._pending_reduce:
            # A second training forward before allreduce_grads() (gradient accumulation): the decoder buckets of the
            # first forward are still in flight and already hold rank-summed values.  Accumulating on top of them and
            # reducing again would scale the first forward's part by the world size, so this is refused outright.
            raise RuntimeError("ShiVAE: a training forward was run with overlap_allreduce on while the gradient "
                               "buckets of the previous forward are still pending; call allreduce_grads() after every "
                               "backward, or set model.overlap_allreduce = False to accumulate gradients over "
                               "several forwards and reduce once")
        losses = torch.zeros(16, dtype=torch.float32, device=dev)
        flags = (0 if grad else N.STEP_FORWARD_ONLY) | (N.STEP_KEEP_PARAMS if keep_params else 0)
        bm = self.micro_batches(T, B)[0]
        self._slice_reserve = 0 if bm >= B else T * bm * (5 * self.x_dim + 4 * self.s_dim + 4 * self.z_dim)
        fmax = None
        if (grad and not keep_params and self.narrow_activations and not self.chunk_rows
                and (T, min(bm, B)) not in self._chunk_done):
            # largest observed fraction of any attribute, from a sample of 16 time steps (sizes the activation buffers
            # of the mask-aware decoder; an estimate is enough, the library checks every chunk's counts)
            sub = mask[::max(1, T // 16)]
            fmax = 1.0 - float(sub.sum(dim=(0, 1), dtype=torch.float32).min()) / float(sub.shape[0] * sub.shape[1])
        self._grow_chunk(T, min(bm, B), fmax)
        graphed = self._graph_eligible(T, B, bm, collect)
        u, eps, seed = self._draw_noise(T, B, tensors=graphed)
        if graphed:
            done = self._step_graphed(x, mask, u, eps, float(beta), float(temp), inv_batch, flags)
            if done is not None:
                self._last["seed"] = seed
                return done, {}
        outs = {w: [] for w in (collect or [])}
        persum = []
        L = N.lib()
        xs = ms = us = es = None
        for b0 in range(0, B, bm):
            b1 = min(B, b0 + bm)
            plan = self._activate(T, b1 - b0)
            xs = ms = us = es = None  # (the previous micro-batch's slices go back to the allocator BEFORE the next ones are cut)
            if b0 == 0 and b1 == B:
                xs, ms, us, es = x.contiguous(), mask.contiguous(), u, eps
            else:
                xs, ms = x[:, b0:b1].contiguous(), mask[:, b0:b1].contiguous()
                if u is not None:
                    us, es = u[:, b0:b1].contiguous(), eps[:, b0:b1].contiguous()
            if u is None:  # in-kernel noise: nothing to slice, one seed per micro-batch
                N.check(L.shivae_plan_set_noise_seed(plan.h, self._mb_seed(seed, b0 // bm)))
            if grad and b1 == B and self.overlap_allreduce and self._dp_backend() == "nccl" and not self._pending_reduce:
                # last micro-batch of a data-parallel step, phase by phase (== shivae_elbo_step): the decoder's
                # gradients are final once its last row chunk is done
                st, m8 = N.stream(), ms.view(torch.uint8)
                N.check(L.shivae_scan_fwd(plan.h, N.ptr(xs), N.ptr(us), N.ptr(es), float(temp), st))
                N.check(L.shivae_prior_kl(plan.h, float(beta), inv_batch, flags, st))
                N.check(L.shivae_decoder_nll(plan.h, N.ptr(xs), N.ptr(m8), inv_batch, flags, st))
                self._start_bucket_allreduce()
                N.check(L.shivae_scan_bwd(plan.h, N.ptr(xs), N.ptr(us), N.ptr(es), float(temp), None, None, None, None, st))
                N.check(L.shivae_loss_reduce(plan.h, float(beta), inv_batch, N.ptr(losses), st))
            else:
                N.check(L.shivae_elbo_step(plan.h, N.ptr(xs), N.ptr(ms.view(torch.uint8)), N.ptr(us), N.ptr(es),
                                           float(beta), float(temp), inv_batch, flags, N.ptr(losses), N.stream()))
            for w in outs:
                o = torch.empty(T, b1 - b0, plan.output_width(w), dtype=torch.float32, device=dev)
                N.check(L.shivae_get_output(plan.h, w, N.ptr(o), N.stream()))
                outs[w].append(o)
            # per-sequence sums over t (base_missvae.py:284,318,343): B x 6 floats, kept for callers and tests
            ps = torch.empty(b1 - b0, 6, dtype=torch.float32, device=dev)
            N.check(L.shivae_get_persum(plan.h, N.ptr(ps), N.stream()))
            persum.append(ps)
        outs = {w: (v[0] if len(v) == 1 else torch.cat(v, dim=1)) for w, v in outs.items()}
        self._last = dict(u=u, eps=eps, seed=seed, persum=torch.cat(persum) if persum else None)
        return losses, outs

    # ------------------------------------------------------------------ captured steps (small, launch-bound models)
    def _graph_eligible(self, T, B, bm, collect):
        if self.use_graphs is False or collect or bm < B or T * B > self.graph_max_rows:
            return False
        if self._dp_backend() is not None and self.overlap_allreduce:
            return False  # (the overlapped all-reduce interleaves host-side collectives with the phases)
        if N.lib().shivae_profile_active():
            return False  # (the per-launch HIP events of the profiler cannot be recorded into a graph)
        plan = self._plans.get((T, B))
        # (one row chunk, and activation buffers that hold it: a narrow plan checks its counts on the host, which a
        # capture cannot do)
        return plan is None or (plan.chunk_rows >= T * B and plan.activation_rows >= plan.chunk_rows)

    def _step_graphed(self, x, mask, u, eps, beta, temp, inv_batch, flags):
        """The step through a HIP graph: the first call with a given (shape, scalars) runs eagerly (it binds the plan,
        allocates the workspace and, for training steps, lets the plan measure its observed fraction); the second
        captures shivae_elbo_step + the per-sequence sums on static buffers; later ones copy the inputs in and replay.
        -> losses tensor, or None to make the caller take the eager path."""
        T, B, _ = x.shape
        key = (T, B, flags, beta, temp, inv_batch)
        ent = self._graphs.get(key)
        if ent is None:
            if key not in self._graph_seen:
                self._graph_seen.add(key)
                return None
            if len(self._graphs) >= 8:  # (schedules move beta / temp every epoch: keep the cache small)
                self._graphs.pop(next(iter(self._graphs)))
            dev = self._flat.device
            plan = self._activate(T, B)
            ent = dict(x=torch.empty_like(x), m=torch.empty(mask.shape, dtype=torch.uint8, device=dev), u=torch.empty_like(u),
                       e=torch.empty_like(eps), losses=torch.zeros(16, device=dev), ps=torch.empty(B, 6, device=dev),
                       plan=plan, graph=torch.cuda.CUDAGraph())
            L = N.lib()
            torch.cuda.synchronize()
            try:
                with torch.cuda.graph(ent["graph"]):
                    ent["losses"].zero_()
                    N.check(L.shivae_elbo_step(plan.h, N.ptr(ent["x"]), N.ptr(ent["m"]), N.ptr(ent["u"]), N.ptr(ent["e"]), beta,
                                               temp, inv_batch, flags, N.ptr(ent["losses"]), N.stream()))
                    N.check(L.shivae_get_persum(plan.h, N.ptr(ent["ps"]), N.stream()))
            except Exception:
                self.use_graphs = False  # (capture not possible here: stay on the eager path for good)
                return None
            self._graphs[key] = ent
        if self._active is not ent["plan"]:  # another shape ran in between: the workspace has to be this plan's again
            self._graphs.pop(key)
            return None
        ent["x"].copy_(x); ent["m"].copy_(mask); ent["u"].copy_(u); ent["e"].copy_(eps)
        ent["graph"].replay()
        self._last = dict(u=u, eps=eps, persum=ent["ps"].clone())
        return ent["losses"].clone()

    def forward(self, x, mask, beta=1., temp=3, inv_batch=None):
        """base_missvae.py:144-164: dict of the 9 loss scalars (K*M == 1, so the ``repeat`` is a no-op).

        In training mode with grad enabled the gradients of n_elbo are computed in the same pass;
        ``loss['n_elbo'].backward()`` makes them visible as ``p.grad``."""
        grad = torch.is_grad_enabled() and self.training
        fresh = self._slices[0][0].grad is None
        if grad and fresh:
            self._gflat.zero_()
        losses, _ = self._step(x, mask, beta, temp, grad, inv_batch=inv_batch)
        vals = losses[:9]
        if self.check_nan or self.print_losses:
            host = vals.cpu()
            if self.check_nan and bool(torch.isnan(host).any()):
                bad = [k for k, v in zip(LOSS_KEYS, host.tolist()) if v != v]
                print("%s Loss NAN" % ", ".join(bad))  # base_missvae.py:223-228, utils.py:171-175
                raise SystemExit(1)
            if self.print_losses:
                print(print_shivae.format(**dict(zip(LOSS_KEYS, host.tolist()))))
        loss_dict = {k: vals[i] for i, k in enumerate(LOSS_KEYS)}
        if grad:
            loss_dict['n_elbo'] = _AttachGrads.apply(self._hook, vals[0], self)
        return loss_dict

    # ------------------------------------------------------------------ reference API around the step
    def _likes_from_params(self, dp):
        """base_missvae.py:111-142 decparams2likes on the raw head outputs [T,B,P]."""
        likes, off = {}, 0
        for t in self.types_list:
            ty, var = t['type'], t['name']
            if ty in ('real', 'pos'):
                mu = dp[..., off:off + 1]
                if self.learn_std:
                    std = torch.nn.functional.softplus(dp[..., off + 1:off + 2])
                    off += 2
                else:
                    std = torch.full_like(mu, FIXED_STD)
                    off += 1
                likes[var] = torch.distributions.Normal(mu, std, validate_args=False)
            elif ty == 'bin':
                likes[var] = torch.distributions.Bernoulli(logits=dp[..., off:off + 1], validate_args=False)
                off += 1
            else:
                C = int(t['dim'])
                th = torch.cat([torch.zeros_like(dp[..., :1]), dp[..., off:off + C - 1]], dim=-1)  # decoders.py:169-170
                likes[var] = torch.distributions.OneHotCategorical(logits=th, validate_args=False)
                off += C - 1
        return likes

    def _params_from_likes(self, likes, like_ref=None):
        """Raw head outputs [T,B,P] from the likelihood objects; attributes absent from `likes` (heter_nll on a subset)
        get zeros shaped like `like_ref`'s batch dimensions."""
        cols = []
        for t in self.types_list:
            ty = t['type']
            if t['name'] not in likes:
                ref = next(iter(likes.values())) if like_ref is None else like_ref
                shp = tuple((ref.loc if hasattr(ref, 'loc') else ref.logits).shape[:-1])
                n = (2 if self.learn_std else 1) if ty in ('real', 'pos') else (1 if ty == 'bin' else int(t['dim']) - 1)
                cols.append(torch.zeros(shp + (n,), device=self._flat.device))
                continue
            lk = likes[t['name']]
            if ty in ('real', 'pos'):
                cols.append(lk.loc)
                if self.learn_std:
                    s = lk.scale
                    cols.append(torch.where(s > 20, s, torch.log(torch.expm1(s.clamp(max=20)))))  # inverse softplus
            elif ty == 'bin':
                cols.append(lk.logits)
            else:
                lg = lk.logits
                cols.append((lg - lg[..., :1])[..., 1:])
        return torch.cat(cols, dim=-1).contiguous().float()

    @torch.no_grad()
    def feed2model(self, x, temp=3):
        """shivae.py:72-154: (z_tup, z_prior_tup, s_tup, likes)."""
        O = N.O
        want = [O["Z"], O["Z_MEAN"], O["Z_STD"], O["ZP_MEAN"], O["ZP_STD"], O["S"], O["S_PROBS"], O["DEC_PARAMS"]]
        x = torch.as_tensor(x, dtype=torch.float32, device=self._flat.device)
        _, o = self._step(x, torch.zeros_like(x, dtype=torch.bool), 1.0, temp, grad=False, keep_params=True, collect=want)
        z_tup = [o[O["Z"]], o[O["Z_MEAN"]], o[O["Z_STD"]]]
        z_prior_tup = [o[O["ZP_MEAN"]], o[O["ZP_STD"]]]
        s_tup = [o[O["S"]], o[O["S_PROBS"]]]
        self._last["dec_params"] = o[O["DEC_PARAMS"]]
        return z_tup, z_prior_tup, s_tup, self._likes_from_params(o[O["DEC_PARAMS"]])

    @torch.no_grad()
    def heter_nll(self, x, mask, likes, types_list=None, eval_obs=False):
        """base_missvae.py:233-285 via the HIP likelihood kernel: dict nll_<type> -> [B]."""
        self._require_gpu()
        dev = self._flat.device
        x = torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
        mask = torch.as_tensor(mask, device=dev)
        mask = (mask != 0) if mask.dtype != torch.bool else mask
        if eval_obs:
            mask = ~mask  # :278-279 evaluate on the missing entries instead
        if types_list is not None:
            # a subset of the attributes (:254 loops over the list it is given): the others count as missing everywhere
            names = {t['name'] for t in types_list}
            unknown = names - {t['name'] for t in self.types_list}
            if unknown:
                raise KeyError("heter_nll: attributes %s are not in the model's types_list" % sorted(unknown))
            if len(names) < len(self.types_list):
                mask = mask.clone()
                for t, lo in zip(self.types_list, self.transform_idx):
                    if t['name'] not in names:
                        mask[..., lo:lo + int(t['dim'])] = True
                likes = {k: v for k, v in likes.items() if k in names}
        T, B, _ = x.shape
        P = self._params_from_likes(likes)
        rowloss = torch.zeros(T * B, 6, device=dev)
        N.check(N.lib().shivae_heter_nll(self._layout.h, N.ptr(x), N.ptr(mask.contiguous().view(torch.uint8)), N.ptr(P),
                                         self.p_dim, None, T * B, 1.0, N.ptr(rowloss), N.stream()))
        s = rowloss.view(T, B, 6).sum(dim=0)
        return {"nll_" + k: s[:, i] for i, k in enumerate(LOSSES)}

    @torch.no_grad()
    def _kl_rows(self, z_tup, z_prior_tup, s_tup):
        self._require_gpu()
        dev = self._flat.device
        _, mu, sd = [torch.as_tensor(a, dtype=torch.float32, device=dev).contiguous() for a in z_tup]
        mup, sdp = [torch.as_tensor(a, dtype=torch.float32, device=dev).contiguous() for a in z_prior_tup]
        probs = torch.as_tensor(s_tup[1], dtype=torch.float32, device=dev).contiguous()
        T, B, Z = mu.shape
        K = probs.shape[-1]
        rowloss = torch.zeros(T * B, 6, device=dev)
        N.check(N.lib().shivae_kl_terms(N.ptr(mu), N.ptr(sd), N.ptr(mup), N.ptr(sdp), Z, N.ptr(probs), K, T * B, Z, K,
                                        1.0, None, None, None, None, None, N.ptr(rowloss), N.stream()))
        return rowloss.view(T, B, 6)

    def kl_q_z(self, z_tup, z_prior_tup):
        """base_missvae.py:287-318 -> [B]."""
        K = self.s_dim
        dummy = torch.full(tuple(z_tup[1].shape[:2]) + (K,), 1.0 / K, device=self._flat.device)
        return self._kl_rows(z_tup, z_prior_tup, (None, dummy))[..., 4].sum(dim=0)

    def kl_q_s(self, s_tup):
        """base_missvae.py:320-343 -> [B]."""
        T, B, _ = s_tup[1].shape
        one = torch.ones(T, B, 1, device=self._flat.device)
        return self._kl_rows((None, one, one), (one, one), s_tup)[..., 5].sum(dim=0)

    @torch.no_grad()
    def compute_loss(self, x, mask, z_tup, z_prior_tup, likes, beta=1., s_tup=None):
        """base_missvae.py:166-231 (Monte-Carlo branch, K == 1) from already computed tuples."""
        nll = self.heter_nll(x, mask, likes)
        nll_loss = sum(nll.values())
        kl = self._kl_rows(z_tup, z_prior_tup, s_tup).sum(dim=0)
        klz, kls = kl[:, 4], kl[:, 5]
        out = {'n_elbo': (nll_loss + beta * klz + beta * kls).mean(), 'nll_loss': nll_loss.mean(),
               'kld': klz.mean() + kls.mean(), 'kld_q_z': klz.mean(), 'kld_q_s': kls.mean()}
        nll_mean = {k: v.mean() for k, v in nll.items()}
        if torch.isnan(klz.sum()):
            print("KL Loss NAN")
            raise SystemExit(1)
        utils.check_nan_losses(nll_mean)
        return {**out, **nll_mean}

    @torch.no_grad()
    def reconstruction(self, x, temp=0.01):
        """base_missvae.py:345-359: ((likes, x_hat), (z_tup, s_tup))."""
        z_tup, z_prior_tup, s_tup, likes = self.feed2model(x, temp)
        return (likes, self.generate_heter_samples(likes)), (z_tup, s_tup)

    @torch.no_grad()
    def generate_heter_samples(self, likes, n_samples=100, seed=None):
        """base_missvae.py:378-404: Monte-Carlo means (round for bin/cat), concatenated on the last dim -- one HIP
        pass over the raw likelihood parameters (csrc/loss.hip mc_impute_kernel)."""
        self._require_gpu()
        P = self._params_from_likes(likes)
        T, B, _ = P.shape
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # follows torch.manual_seed
        xhat = torch.empty(T, B, self.x_dim, dtype=torch.float32, device=P.device)
        N.check(N.lib().shivae_mc_impute(self._layout.h, N.ptr(P), self.p_dim, T * B, int(n_samples), int(seed),
                                         N.ptr(xhat), N.stream()))
        return xhat

    @torch.no_grad()
    def impute_mean(self, x, temp=0.01):
        """Monte-Carlo-free imputation: real/pos -> mean, bin -> [logit>0], cat -> one-hot argmax."""
        _, _, _, likes = self.feed2model(x, temp)
        cols = []
        for t in self.types_list:
            lik, ty = likes[t['name']], t['type']
            if ty in ('real', 'pos'):
                cols.append(lik.loc)
            elif ty == 'bin':
                cols.append((lik.logits > 0).float())
            else:
                cols.append(torch.nn.functional.one_hot(lik.logits.argmax(-1), int(t['dim'])).float())
        return torch.cat(cols, dim=-1)

    @torch.no_grad()
    def sample_from_prior(self, batch_size, ts, temp=0.02):
        """shivae.py:156-214: (z_prior_tup, s_tup, likes) generated from the prior, as one HIP scan + the batched
        decoder (csrc/scan.hip scan_gen_kernel).  Noise: ``set_noise(u, eps)`` or ``noise_mode``."""
        self._require_gpu()
        dev = self._flat.device
        T, B = int(ts), int(batch_size)
        u, eps, seed = self._draw_noise(T, B)
        O, L = N.O, N.lib()
        bm = self.micro_batches(T, B)[0]
        want = [O["Z"], O["Z_MEAN"], O["Z_STD"], O["S"], O["S_PROBS"], O["DEC_PARAMS"]]
        outs = {w: [] for w in want}
        for b0 in range(0, B, bm):
            b1 = min(B, b0 + bm)
            plan = self._activate(T, b1 - b0)
            if u is None:  # in-kernel noise
                us = es = None
                N.check(L.shivae_plan_set_noise_seed(plan.h, self._mb_seed(seed, b0 // bm)))
            else:
                us, es = (u, eps) if (b0 == 0 and b1 == B) else (u[:, b0:b1].contiguous(), eps[:, b0:b1].contiguous())
            N.check(L.shivae_sample_prior(plan.h, N.ptr(us), N.ptr(es), float(temp), N.stream()))
            for w in want:
                o = torch.empty(T, b1 - b0, plan.output_width(w), dtype=torch.float32, device=dev)
                N.check(L.shivae_get_output(plan.h, w, N.ptr(o), N.stream()))
                outs[w].append(o)
        o = {w: (v[0] if len(v) == 1 else torch.cat(v, dim=1)) for w, v in outs.items()}
        z_prior_tup = [o[O["Z"]], o[O["Z_MEAN"]], o[O["Z_STD"]]]
        s_tup = [o[O["S"]], o[O["S_PROBS"]]]
        self._last = dict(u=u, eps=eps, seed=seed, persum=None, dec_params=o[O["DEC_PARAMS"]])
        return z_prior_tup, s_tup, self._likes_from_params(o[O["DEC_PARAMS"]])

    @torch.no_grad()
    def sample(self, batch_size, ts, temp=0.01):
        """base_missvae.py:361-376: ((likes, x_hat), z_prior_tup, s_tup)."""
        z_prior_tup, s_tup, likes = self.sample_from_prior(batch_size, ts, temp)
        return (likes, self.generate_heter_samples(likes)), z_prior_tup, s_tup

    # ------------------------------------------------------------------ data parallel
    def _dp_backend(self):
        """Backend name if a process group with more than one rank is up (SHIVAE_FORCE_DP=1: also with one), else None."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SHIVAE_FORCE_DP")):
            return dist.get_backend()
        return None

    def _start_bucket_allreduce(self):
        """Asynchronous RCCL all-reduce of the decoder's part of the flat gradient buffer, bucket by bucket.  RCCL's
        stream orders itself behind the kernels already enqueued on the current stream (the decoder phase), and the
        kernels enqueued after this call (reverse-time scan, cell gradients) overlap with the collective: they write
        the front of the buffer only."""
        import torch.distributed as dist
        off = self._layout.param_info(N.P["DEC0_W"], 0)[0]  # the decoder tensors are the tail of the layout
        n, b = self._n_flat, max(1, self.bucket_floats)
        self._pending_reduce = [(off, [dist.all_reduce(self._gflat[a:min(a + b, n)], op=dist.ReduceOp.SUM, async_op=True)
                                       for a in range(off, n, b)])]

    def allreduce_grads(self):
        """Sum the flat gradient buffer over the ranks (RCCL all-reduce over xGMI).  Each rank ran its
        batch shard with inv_batch = 1/B_global, so the sum IS the global-batch gradient."""
        import torch.distributed as dist
        backend = self._dp_backend()
        if backend is None:
            return
        if self._pending_reduce:  # the decoder's buckets are in flight: reduce the rest, then join them
            (off, works), = self._pending_reduce
            self._pending_reduce = []
            dist.all_reduce(self._gflat[:off], op=dist.ReduceOp.SUM)
            for w in works:
                w.wait()
        elif backend == "nccl" or not self._gflat.is_cuda:
            # RCCL orders itself behind the step's kernels on the current stream: no host wait
            dist.all_reduce(self._gflat, op=dist.ReduceOp.SUM)
        else:
            # gloo (CPU / 1-GPU rehearsals of the N>1 path): staged through the host by hand.  torch's gloo backend
            # handed a DEVICE tensor while the step's kernels are still in flight was seen to block both ranks
            # inside all_reduce for good (two ranks sharing one GPU); the blocking copy below is
            # also the wait for the step.
            host = self._gflat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self._gflat.copy_(host)
