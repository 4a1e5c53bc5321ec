"""Fused global-norm clip + Adam on the model's flat buffers (one HIP kernel each), replacing
``torch.nn.utils.clip_grad_norm_(model.parameters(), clip)`` + ``torch.optim.Adam.step()``
(trainers.py:222-224, main_hmm.py:123)."""
import torch

from . import _native as N


class FusedClipAdam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(model.parameters(), lr)`` when the model is a shivae_amd.ShiVAE.

    ``step(clip=...)`` also performs the gradient-norm clipping; ``clip_grad_norm_`` called before a
    plain ``step()`` is equally fine (then pass no clip).  ``state_dict`` has torch.optim.Adam's layout
    (per-parameter ``step``/``exp_avg``/``exp_avg_sq``), so reference checkpoints load."""

    def __init__(self, model, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, clip=None):
        self.model = model
        # (weight_decay / amsgrad / ...: torch.optim.Adam's other defaults at their no-op values, so that a checkpoint
        # written here loads into torch.optim.Adam and back)
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                                                        maximize=False, foreach=None, capturable=False,
                                                        differentiable=False, fused=None, decoupled_weight_decay=False))
        self.clip = clip
        dev = model.flat_params.device
        self._m = torch.zeros_like(model.flat_params)
        self._v = torch.zeros_like(model.flat_params)
        self._scratch = torch.zeros(1024, device=dev)
        self._sq = torch.zeros(1, device=dev)
        self._step = 0
        self.last_grad_sqnorm = self._sq

    def zero_grad(self, set_to_none=False):
        # keep p.grad aliased to the flat gradient buffer: one memset instead of one per parameter
        self.model.flat_grads.zero_()
        if self.model._slices[0][0].grad is None:
            self.model._publish_grads(None)

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        clip = self.clip if clip is None else clip
        g = self.param_groups[0]
        m = self.model
        n = m.flat_params.numel()
        L = N.lib()
        self._step += 1
        N.check(L.shivae_sqnorm(N.ptr(m.flat_grads), n, N.ptr(self._scratch), N.ptr(self._sq), N.stream()))
        N.check(L.shivae_clip_adam(N.ptr(m.flat_params), N.ptr(m.flat_grads), N.ptr(self._m), N.ptr(self._v), n,
                                   N.ptr(self._sq), float(clip or 0.0), float(g["lr"]), float(g["betas"][0]),
                                   float(g["betas"][1]), float(g["eps"]), self._step, N.stream()))

    def state_dict(self):
        state = {}
        for i, (p, off, cnt, shape) in enumerate(self.model._slices):
            state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[off:off + cnt].view(shape).clone(),
                        "exp_avg_sq": self._v[off:off + cnt].view(shape).clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(self.model._slices)))
        return {"state": state if self._step else {}, "param_groups": groups}

    def load_state_dict(self, sd):
        for i, (p, off, cnt, shape) in enumerate(self.model._slices):
            st = sd["state"].get(i)
            if st is None:
                continue
            self._m[off:off + cnt].copy_(st["exp_avg"].reshape(-1))
            self._v[off:off + cnt].copy_(st["exp_avg_sq"].reshape(-1))
            self._step = int(st["step"])
        for k in ("lr", "betas", "eps"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
