"""Host helpers that mirror the reference's ``lib/utils.py`` names used on the hot path."""
import copy
import csv

import numpy as np
import torch


def read_csv_types(csv_name):
    """lib/utils.py:200-212: list of dicts with string fields name,type,dim,nclass,index."""
    with open(csv_name) as f:
        return [{k: v for k, v in row.items()} for row in csv.DictReader(f, skipinitialspace=True)]


def get_x_dim(types_list):
    """lib/utils.py:229-241."""
    return int(np.sum([int(t["dim"]) for t in types_list]))


def reindex_types_list(types_list):
    """lib/utils.py:275-288."""
    out = copy.deepcopy(types_list)
    for i, t in enumerate(out):
        t.update({"index": i})
    return out


def get_idx_transform(types_list):
    """lib/utils.py:291-315: column offset of every attribute after one-hot expansion."""
    out, off = [], 0
    for t in types_list:
        out.append(off)
        off += int(t["dim"])
    return out


def zero_fill(x, mask):
    """lib/utils.py:89-100."""
    return torch.where(mask, torch.zeros_like(x), x)


def chop_sequence(T, max_length=50):
    """lib/utils.py:123-141.  Bookkeeping only in the reference (h, c are carried un-detached across
    chunks), so the fused scan ignores it; kept for API compatibility."""
    if T > max_length:
        t_range = np.arange(max_length, T, max_length)
        if T not in t_range:
            t_range = np.append(t_range, T)
        return t_range
    return [T]


def check_nan_losses(nll_heter_dict):
    """lib/utils.py:162-177: print and terminate on a NaN loss term."""
    for key, nll in nll_heter_dict.items():
        if torch.isnan(nll):
            print("{} Loss NAN".format(key))
            print(nll)
            raise SystemExit(1)


def make_types_list(n_real=0, n_pos=0, n_bin=0, cat_classes=(), interleave=False):
    """Synthetic types_list (same dict fields as data_types.csv)."""
    attrs = [("real", 1, 1)] * n_real + [("pos", 1, 1)] * n_pos + [("bin", 1, 2)] * n_bin
    attrs += [("cat", int(c), int(c)) for c in cat_classes]
    if interleave:
        buckets = {}
        for a in attrs:
            buckets.setdefault(a[0], []).append(a)
        attrs = []
        while any(buckets.values()):
            for ty in ("real", "pos", "bin", "cat"):
                if buckets.get(ty):
                    attrs.append(buckets[ty].pop(0))
    return [dict(name="var%d" % i, type=ty, dim=str(dim), nclass=str(nc), index=str(i))
            for i, (ty, dim, nc) in enumerate(attrs)]
