"""HeterogeneousScaler with the reference's API (``src/lib/scalers.py:27-220``) whose transform / inverse
transform run as ONE HIP pass each (``csrc/prep.hip``), fused with ``Trainer.preprocess_batch``
(``models/train.py:333-362``: B,T,D -> T,B,X, float32 / bool) and ``utils.zero_fill``.

``fit`` is the one-off host statistics pass of the reference (nanmean / nanstd per attribute, ``log(x+1)`` for
positive attributes, ``lib/utils.py:319-331``).  Categories of a ``cat`` attribute are the integers
0..nclass-1 (what ``OneHotEncoder`` yields when every class occurs in the fitted data)."""
import numpy as np
import torch

from . import _native as N
from . import utils


class HeterogeneousScaler:
    def __init__(self, types_list=None):
        assert types_list is not None
        self.types_list = utils.reindex_types_list(types_list)          # scalers.py:39
        self.transform_idx = utils.get_idx_transform(self.types_list)   # scalers.py:42
        self.x_dim = int(utils.get_x_dim(self.types_list))
        self.mean = np.zeros(len(self.types_list))
        self.std = np.ones(len(self.types_list))
        self._tables = {}

    # ---- fit (host, one-off) ------------------------------------------------------------------
    def fit(self, data, mask=None):
        """scalers.py:60-83."""
        data = np.asarray(data, dtype=np.float64)
        mask = np.isnan(data) if mask is None else np.asarray(mask, dtype=bool)
        flat = np.where(mask, np.nan, data).reshape(-1, data.shape[-1])
        for i, t in enumerate(self.types_list):
            col = flat[:, int(t['index'])]
            if t['type'] == 'pos':
                col = np.where(np.isnan(col), np.nan, np.log(np.where(np.isnan(col), 1.0, col) + 1.0))
            if t['type'] in ('real', 'pos'):
                self.mean[i] = np.nanmean(col)
                self.std[i] = np.clip(np.nanstd(col), 1e-6, 1e20)
        self._tables = {}

    # ---- device tables ------------------------------------------------------------------------
    def _device_tables(self, device):
        key = str(device)
        if key not in self._tables:
            rec = np.zeros(len(self.types_list), dtype=[('type', 'i4'), ('xoff', 'i4'), ('dim', 'i4'), ('index', 'i4'),
                                                         ('mean', 'f8'), ('std', 'f8')])
            col2attr = np.zeros(self.x_dim, dtype=np.int32)
            for i, t in enumerate(self.types_list):
                dim = int(t['dim'])
                rec[i] = (N.TYPE_CODE[t['type']], self.transform_idx[i], dim, int(t['index']), self.mean[i], self.std[i])
                col2attr[self.transform_idx[i]:self.transform_idx[i] + dim] = i
            attrs = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
            self._tables[key] = (attrs, torch.from_numpy(col2attr).to(device))
        return self._tables[key]

    @staticmethod
    def _device():
        if not torch.cuda.is_available():
            raise N.ShiVAENativeError("HeterogeneousScaler runs its transforms as HIP kernels; no GPU is visible and "
                                      "there is no CPU fallback")
        N.lib()
        return torch.device('cuda', torch.cuda.current_device())

    def transform_to_device(self, data, mask=None, transpose=True, zero_fill=True):
        """data[N,T,D] (numpy / tensor, NaN allowed), mask[N,T,D] -> device tensors x[T,N,X] f32, mask[T,N,X] bool
        (the layout the step consumes); transpose=False keeps [N,T,X]."""
        dev = self._device()
        d = torch.as_tensor(np.asarray(data) if not torch.is_tensor(data) else data)
        f64 = d.dtype == torch.float64  # numpy datasets: scale in f64 on the device, as the reference does on the host
        d = d.to(dev, torch.float64 if f64 else torch.float32).contiguous()
        n, t, D = d.shape
        assert D == len(self.types_list)
        m = None
        if mask is not None:
            m = torch.as_tensor(np.asarray(mask) if not torch.is_tensor(mask) else mask).to(dev)
            m = (m != 0).contiguous().view(torch.uint8)
        attrs, col2attr = self._device_tables(dev)
        shape = (t, n, self.x_dim) if transpose else (n, t, self.x_dim)
        x = torch.empty(shape, dtype=torch.float32, device=dev)
        mo = torch.empty(shape, dtype=torch.uint8, device=dev)
        flags = (N.SCALE_TRANSPOSE if transpose else 0) | (N.SCALE_ZERO_FILL if zero_fill else 0) | (N.SCALE_F64 if f64 else 0)
        N.check(N.lib().shivae_scale_transform(N.ptr(d), N.ptr(m), n, t, D, self.x_dim, N.ptr(attrs), N.ptr(col2attr),
                                               N.ptr(x), N.ptr(mo), flags, N.stream()))
        return x, mo.view(torch.bool)

    def transform(self, data, mask=None):
        """scalers.py:85-131: numpy in, numpy out ([N,T,X] data with NaN kept, bool mask)."""
        x, m = self.transform_to_device(data, mask, transpose=False, zero_fill=False)
        return x.cpu().numpy().astype(np.float64), m.cpu().numpy()

    def inverse_transform_device(self, data, mask=None, transpose=True):
        dev = self._device()
        x = torch.as_tensor(np.asarray(data) if not torch.is_tensor(data) else data).to(dev, torch.float32).contiguous()
        a, b, X = x.shape
        assert X == self.x_dim
        m = None
        if mask is not None:
            m = torch.as_tensor(np.asarray(mask) if not torch.is_tensor(mask) else mask).to(dev)
            m = (m != 0).contiguous().view(torch.uint8)
        attrs, _ = self._device_tables(dev)
        D = len(self.types_list)
        n, t = (b, a) if transpose else (a, b)
        out = torch.empty(n, t, D, dtype=torch.float64, device=dev)
        mo = torch.empty(n, t, D, dtype=torch.uint8, device=dev)
        N.check(N.lib().shivae_scale_inverse(N.ptr(x), N.ptr(m), n, t, D, X, N.ptr(attrs), N.ptr(out), N.ptr(mo),
                                             (N.SCALE_TRANSPOSE if transpose else 0) | N.SCALE_F64, N.stream()))
        return out, mo.view(torch.bool)

    def inverse_transform(self, data, mask=None, transpose=True):
        """scalers.py:133-183: data[A,B,X] -> (data[A,B,D], mask), axes 0 and 1 swapped if transpose."""
        d, m = self.inverse_transform_device(data, mask, transpose)
        return d.cpu().numpy(), m.cpu().numpy()

    def param_inverse_transform(self, likes, transpose=True):
        """scalers.py:185-220: de-standardised likelihood parameters as numpy (B,T,1 when transpose)."""
        out = {}
        for i, t in enumerate(self.types_list):
            lik, ty = likes[t['name']], t['type']
            tr = (lambda a: a.transpose(0, 1)) if transpose else (lambda a: a)
            if ty in ('real', 'pos'):
                mean = lik.loc * self.std[i] + self.mean[i]
                std = lik.scale * self.std[i]
                if ty == 'pos':
                    mean, std = torch.exp(mean) - 1, torch.exp(std) - 1
                out[t['name']] = [tr(mean).cpu().numpy(), tr(std).cpu().numpy()]
            else:
                out[t['name']] = tr(lik.probs).cpu().numpy()
        return out
