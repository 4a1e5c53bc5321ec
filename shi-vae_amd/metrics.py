"""Imputation-quality metrics with the reference's API (``src/lib/loss.py:18-222``, ``lib/utils.py:346-390``):
NRMSE (real, pos), error rate (bin), confusion-matrix error (cat) on the artificially missing entries, and the
burst cross-correlation.  Host-side numpy, as in the reference (evaluation runs once per epoch on de-scaled
``[N,T,D]`` arrays); plots are dropped.  scikit-learn is not used: ``mean_squared_error(squared=False)``
(``loss.py:145``) no longer exists in the installed version, RMSE and the row-normalised confusion matrix are
written out."""
import numpy as np


def artmask2burst(mask):
    """utils.py:346-390: number the runs of consecutive missing entries 1, 2, ... (0 = observed)."""
    m = np.asarray(mask).flatten().astype(bool)
    starts = m & ~np.concatenate([[False], m[:-1]])
    return np.cumsum(starts) * m


def norm_cross_corr(x, x_hat, mask):
    """loss.py:52-68."""
    a = x[mask] - x[mask].mean()
    b = x_hat[mask] - x_hat[mask].mean()
    c = np.correlate(a, b, mode="same")
    return c, c.max()


def burst_cross_correlation(x, x_hat, mask):
    """loss.py:18-49: sum over sequences and bursts of the peak cross-correlation inside the burst."""
    total = []
    for i in range(x.shape[0]):
        xi, hi, mi = x[i, ...].flatten(), x_hat[i, ...].flatten(), mask[i, ...].flatten()
        num = artmask2burst(mi)
        for burst in np.unique(num)[1:]:
            idx = np.where(num == burst)[0]
            total.append(norm_cross_corr(xi, hi, idx)[1])
    return np.sum(total)


def _rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))


def _confusion_error(y_true, y_pred):
    """sum(1 - diag(confusion_matrix(..., normalize='true'))) over the labels present in either array
    (loss.py:163-164); a row without true samples is all zeros (as sklearn does), i.e. contributes 1."""
    labels = np.unique(np.concatenate([y_true, y_pred]))
    err = 0.0
    for lab in labels:
        sel = y_true == lab
        err += 1.0 - (np.mean(y_pred[sel] == lab) if sel.any() else 0.0)
    return float(err)


def compute_avg_error(x, x_hat, mask, types_list, img_path=".", plot_imp=False):
    """loss.py:71-201 -> (error_obs, error_miss, corr_obs, corr_miss); only the *_miss dicts are filled, as in
    the reference (the observed branch is commented out there)."""
    names = [t["name"] for t in types_list]
    error_obs, error_miss = dict.fromkeys(names, 0), dict.fromkeys(names, 0)
    corr_obs, corr_miss = dict.fromkeys(names, 0), dict.fromkeys(names, 0)
    for t in types_list:
        ty, var, idx = t["type"], t["name"], int(t["index"])
        xt = np.array(x[:, :, idx:idx + 1], dtype=np.float64)
        ht = np.array(x_hat[:, :, idx:idx + 1], dtype=np.float64)
        mt = np.asarray(mask[:, :, idx:idx + 1])
        if ty == "bin":
            ht = np.round(ht)
        if (mt == 1).sum() == 0:
            continue
        sel = mt == 1
        if ty == "real":
            norm = (xt.max() - xt.min()) if xt.max() != xt.min() else 1
            err = _rmse(xt[sel], ht[sel]) / norm
        elif ty == "pos":
            xt[xt < 0] = 0
            ht[ht < 0] = 0
            if xt.max() > 1000:
                xt, ht = np.log(xt + 1), np.log(ht + 1)
            norm = (xt.max() - xt.min()) if xt.max() != xt.min() else 1
            err = _rmse(xt[sel], ht[sel]) / norm
        elif ty == "bin":
            err = float(np.mean(xt[sel] != ht[sel]))
        elif ty == "cat":
            err = _confusion_error(xt[sel], ht[sel])
        else:
            err = 0
        error_miss[var] += err
        corr_miss[var] += burst_cross_correlation(xt, ht, mt) / mt.sum()
    return error_obs, error_miss, corr_obs, corr_miss


def get_type_error(error_miss, types_list):
    """loss.py:204-222."""
    out = dict.fromkeys({t["type"] for t in types_list}, 0)
    for var, err in error_miss.items():
        out[[t["type"] for t in types_list if t["name"] == var].pop()] += err
    return out
