"""CPU restatement (numpy, float64) of the reference's HeterogeneousScaler -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/lib/scalers.py (cited per function) and lib/utils.py:291-331.  Only tests/ and
__graft_entry__.smoke() may import this module; the product path is the HIP pre/post pass
(shi-vae_amd/csrc/prep.hip) behind shi-vae_amd/scalers.py.

Pinning: StandardScaler / PositiveScaler (fit, transform, inverse_transform, std_inverse_transform) and
HeterogeneousScaler.transform / inverse_transform on real+pos+bin attributes are checked against outputs of the
reference classes (tests/golden/scaler_*.npz, made by tests/golden/make_golden_scaler.py).  The categorical
part is PARITY UNPINNED: the reference's CategoricalScaler cannot be constructed under the installed
scikit-learn (OneHotEncoder(sparse=False), scalers.py:300) and HeterogeneousScaler.fit uses np.NaN (scalers.py:78,
removed in numpy 2); it is restated from the source text (one-hot over categories 0..R-1, NaN where the raw
value is NaN, argmax back).
"""
import numpy as np


def x_offsets(types_list):
    """lib/utils.py:291-315 get_idx_transform: prefix sum of 'dim'."""
    off, out = 0, []
    for t in types_list:
        out.append(off)
        off += int(t["dim"])
    return out, off


def dist_moments(data):
    """lib/utils.py:319-331."""
    mean = np.nanmean(data)
    std = np.clip(np.nanstd(data), 1e-6, 1e20)
    return float(mean), float(std)


def fit(types_list, data, mask=None):
    """scalers.py:60-83 (+ :234-236 real, :264-269 pos): per-attribute (mean, std); (0, 1) for bin/cat."""
    data = np.asarray(data, dtype=np.float64)
    mask = np.isnan(data) if mask is None else np.asarray(mask, dtype=bool)
    flat = np.where(mask, np.nan, data).reshape(-1, data.shape[-1])
    out = []
    for i, t in enumerate(types_list):
        col = flat[:, i]
        if t["type"] == "pos":
            col = np.where(np.isnan(col), np.nan, np.log(np.where(np.isnan(col), 1.0, col) + 1.0))
            out.append(dist_moments(col))
        elif t["type"] == "real":
            out.append(dist_moments(col))
        else:
            out.append((0.0, 1.0))
    return out


def transform(types_list, moments, data, mask=None):
    """scalers.py:85-131: data[N,T,D] -> (x[N,T,X], mask[N,T,X]); NaN stays NaN (zero_fill comes later)."""
    data = np.asarray(data, dtype=np.float64)
    mask = np.isnan(data) if mask is None else np.asarray(mask, dtype=bool)
    offs, X = x_offsets(types_list)
    x = np.zeros(data.shape[:-1] + (X,))
    m = np.zeros(data.shape[:-1] + (X,), dtype=bool)
    for i, t in enumerate(types_list):
        v, dim, lo = data[..., i], int(t["dim"]), offs[i]
        mean, std = moments[i]
        if t["type"] == "real":
            x[..., lo] = (v - mean) / std                                   # :238-240
        elif t["type"] == "pos":
            with np.errstate(invalid="ignore", divide="ignore"):
                x[..., lo] = (np.log(v + 1.0) - mean) / std                 # :271-275
        elif t["type"] == "bin":
            x[..., lo] = v
        else:                                                               # :308-315
            nan = np.isnan(v)
            idx = np.where(nan, 0, v).astype(np.int64)
            oh = (idx[..., None] == np.arange(dim)).astype(np.float64)
            oh[nan] = np.nan
            x[..., lo:lo + dim] = oh
        m[..., lo:lo + dim] = mask[..., i:i + 1]                            # :117-119
    return x, m


def inverse_transform(types_list, moments, x, mask=None, transpose=True):
    """scalers.py:133-183: x[A,B,X] -> (data[A,B,D], mask[A,B,D]), axes 0,1 swapped if transpose."""
    x = np.asarray(x, dtype=np.float64)
    mask = np.isnan(x) if mask is None else np.asarray(mask, dtype=bool)
    offs, _ = x_offsets(types_list)
    d = np.zeros(x.shape[:-1] + (len(types_list),))
    m = np.zeros(x.shape[:-1] + (len(types_list),), dtype=bool)
    for i, t in enumerate(types_list):
        lo, dim = offs[i], int(t["dim"])
        mean, std = moments[i]
        if t["type"] == "real":
            d[..., i] = x[..., lo] * std + mean                            # :242-245
        elif t["type"] == "pos":
            d[..., i] = np.exp(x[..., lo] * std + mean) - 1.0               # :277-283
        elif t["type"] == "bin":
            d[..., i] = x[..., lo]
        else:                                                               # :317-323
            oh = x[..., lo:lo + dim]
            v = np.argmax(oh, axis=-1).astype(np.float64)
            v[np.isnan(oh[..., -1])] = np.nan
            d[..., i] = v
        m[..., i] = mask[..., lo]                                           # :171
    if transpose:
        d, m = d.transpose(1, 0, 2), m.transpose(1, 0, 2)
    return d, m


def std_inverse_transform(ttype, moment, std_hat):
    """scalers.py:247-250 (real), :285-289 (pos)."""
    s = np.asarray(std_hat, dtype=np.float64) * moment[1]
    return np.exp(s) - 1.0 if ttype == "pos" else s
