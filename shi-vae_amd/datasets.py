"""Datasets and the synthetic heterogeneous-HMM generator with the reference's API and on-disk formats
(SURVEY.md 8f row f3): ``src/lib/datasets.py`` (HeterDataset, HMMDatset, standard_collate, load_pickle,
load_data_pickle) and ``src/hmm_dataset/heterHMMdataset.py`` (ToyDatasetGenerator: the ``.pickle`` with keys
X, Z, trans_mat, start_prob, theta, mask_{p}_{i}, train/val/test_index and ``data_types.csv``).

Host-side numpy, as in the reference.  The reference samples its chains with ``hmmlearn`` (not installed):
the Markov chain and the emissions are sampled directly here, with the same parameters (transition matrices
:76-87, emission means on linspace(-20, 20, 20) :127-148, 0/1 Bernoulli emissions :175-187, dominant-class
categoricals :222-231, softplus of a Gaussian for positives :244-249, burst masks :292-363).  Random streams
differ from the reference's -- PARITY UNPINNED for the generator; the file format is what the loaders pin."""
import csv
import os
import pickle
import random

import numpy as np
from torch.utils import data


# ------------------------------------------------------------------------------------------ datasets.py
class HeterDataset(data.Dataset):
    """datasets.py:18-77."""

    def __init__(self, data, mask, data_full=None, mask_artificial=None, types_list=None):
        assert types_list is not None
        self.data, self.mask = data, mask
        self.mask_artificial = mask_artificial if mask_artificial is not None else mask
        self.data_full = data_full if data_full is not None else data
        self.eval_opt = False
        self.types_list = types_list
        self.idx_vars = sorted(int(t['index']) for t in types_list)

    def set_eval(self, eval_opt=False):
        self.eval_opt = eval_opt

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        seq, m = self.data[index][:, self.idx_vars], self.mask[index][:, self.idx_vars]
        if self.eval_opt:
            return seq, m, self.data_full[index][:, self.idx_vars], self.mask_artificial[index][:, self.idx_vars], index
        return seq, m, index


class HMMDatset(HeterDataset):
    """datasets.py:80-137 (the reference's spelling)."""

    def __init__(self, data, mask, data_full=None, mask_artificial=None, types_list=None, index=None, Z=None, theta=None):
        super().__init__(data, mask, data_full=data_full, mask_artificial=mask_artificial, types_list=types_list)
        self.theta, self.index = theta, index
        self.data, self.data_full = self.data[index], self.data_full[index]
        self.mask, self.mask_artificial = self.mask[index], self.mask_artificial[index]
        self.Z = Z[index] if Z is not None else None

    def __len__(self):
        return len(self.index)


def standard_collate(batch):
    """datasets.py:140-197: (data[B,T,D], mask, [data_full, mask_artificial,] batch_attributes)."""
    cols = list(zip(*batch))
    index = np.stack(cols[-1])
    arrs = [np.stack(c) for c in cols[:-1]]
    arrs[1] = arrs[1].astype(bool)
    if len(arrs) > 2:
        arrs[3] = arrs[3].astype(bool)
    B, T = arrs[0].shape[:2]
    attrs = {'dates': list(range(T)), 'sequence_lengths': B * [T], 'batch_lengths': B * [B], 'index': index}
    return (*arrs, attrs)


def load_pickle(data_path, percent_miss=0, mask=1):
    """datasets.py:200-222: X, Z, theta, mask, (train, val, test index) from a generator pickle."""
    with open(data_path, 'rb') as handle:
        d = pickle.load(handle)
    return (d["X"], d["Z"], d["theta"], d["mask_" + str(percent_miss) + "_" + str(mask)],
            (d["train_index"], d["val_index"], d["test_index"]))


def load_data_pickle(dataset, data_path, percent_miss=0, mask_i=1):
    """datasets.py:225-266."""
    return load_pickle(os.path.join(data_path, dataset, dataset + ".pickle"), percent_miss=percent_miss, mask=mask_i)


def read_types(csv_path):
    """lib/utils.py:200-212 on data_types.csv."""
    with open(csv_path) as f:
        return [dict(r) for r in csv.DictReader(f, skipinitialspace=True)]


# ------------------------------------------------------------------------------ heterHMMdataset.py
class ToyDatasetGenerator:
    """heterHMMdataset.py:49-462 without hmmlearn."""

    def __init__(self, x_real_dim=2, x_pos_dim=2, x_bin_dim=1, x_cat_dim=1, x_cat_class=(3,), n_components=4, T=48,
                 N=1000, save_root="../../data", seed=None):
        self.rng = np.random.default_rng(seed)
        self.x_real_dim, self.x_pos_dim, self.x_bin_dim, self.x_cat_dim = x_real_dim, x_pos_dim, x_bin_dim, x_cat_dim
        self.x_cat_class = list(x_cat_class)
        self.X_dim = x_real_dim + x_pos_dim + x_bin_dim + x_cat_dim
        self.n_components, self.T, self.N = n_components, T, N
        self.toy_name = "hmm_heter_{}_{}real_{}pos_{}bin_{}cat_{}_{}".format(N, x_real_dim, x_pos_dim, x_bin_dim,
                                                                             x_cat_dim, n_components, T)   # :71-73
        self.save_dir = os.path.join(save_root, self.toy_name)
        K = n_components
        if K == 3:                                                                                          # :76-80
            self.transmat = np.array([[0.9, 0.1, 0], [0., 0.9, 0.1], [0.1, 0., 0.9]])
            self.startprob = np.array([0.7, 0.2, 0.1])
        elif K == 4:                                                                                        # :82-87
            self.transmat = np.array([[0.7, 0.2, 0.0, 0.1], [0.3, 0.5, 0.2, 0.0], [0.0, 0.3, 0.5, 0.2], [0.2, 0.0, 0.2, 0.6]])
            self.startprob = np.array([0.6, 0.3, 0.1, 0.0])
        else:                                                                                               # :90-94
            self.transmat = self.rng.random((K, K))
            self.transmat /= self.transmat.sum(1, keepdims=True)
            self.startprob = self.rng.random(K)
            self.startprob /= self.startprob.sum()
        self.means_gauss = self._emission_means(x_real_dim)
        self.means_pos = self._emission_means(x_pos_dim)
        self.prob_bin = [self._crazyshuffle(np.concatenate([np.ones((K, 1)), np.zeros((K, 1))], 1)) for _ in range(x_bin_dim)]
        self.prob_cat = [self._emission_cat(c) for c in self.x_cat_class[:x_cat_dim]]
        th = [{"mean": m, "covar": 0.5 * np.tile(np.identity(x_real_dim), (K, 1, 1))} for m in self.means_gauss.T]
        th += [{"mean": m, "covar": 0.5 * np.tile(np.identity(x_pos_dim), (K, 1, 1))} for m in self.means_pos.T]
        th += [{"prob": p} for p in self.prob_bin] + [{"prob": p} for p in self.prob_cat]
        self.theta = {"var" + str(i): t for i, t in enumerate(th)}                                           # :283-284

    def _emission_means(self, dim):
        """:127-148: per state a mean drawn from its own slice of linspace(-20, 20, 20), states shuffled."""
        K, grid = self.n_components, np.linspace(-20, 20, 20)
        steps = np.round(np.linspace(0, 20, K + 1)).astype(int)
        means = np.array([[grid[self.rng.integers(steps[s], max(steps[s] + 1, steps[s + 1]))] for s in range(K)]
                          for _ in range(dim)]).T.reshape(K, dim) if dim else np.zeros((K, 0))
        self.rng.shuffle(means)
        return means

    def _crazyshuffle(self, arr):
        """:216-220: permute every row independently."""
        return np.stack([row[self.rng.permutation(arr.shape[1])] for row in arr])

    def _emission_cat(self, C):
        """:222-231: one dominant class (0.8-1.0) per state, every class dominant in some state when K >= C."""
        K = self.n_components
        p = self.rng.integers(8, 11, (K, 1)) / 10
        mat = np.concatenate([p, np.repeat(np.round((1 - p) / (C - 1), 2), C - 1, 1)], 1)
        em = mat
        for _ in range(10000):
            if len(np.unique(np.argmax(em, 1))) == min(C, K):
                break
            em = self._crazyshuffle(mat)
        return em / em.sum(1, keepdims=True)

    def generate_samples(self):
        """:233-281: one sequence X[T, D] (real | pos | bin | cat columns) and its states Z[T]."""
        K, T, rng = self.n_components, self.T, self.rng
        Z = np.zeros(T, dtype=int)
        Z[0] = rng.choice(K, p=self.startprob)
        for t in range(1, T):
            Z[t] = rng.choice(K, p=self.transmat[Z[t - 1]])
        Xr = self.means_gauss[Z] + np.sqrt(0.5) * rng.standard_normal((T, self.x_real_dim))
        Xp = np.log1p(np.exp(self.means_pos[Z] + np.sqrt(0.5) * rng.standard_normal((T, self.x_pos_dim))))   # :246
        Xb = np.stack([(rng.random(T) >= p[Z, 0]).astype(float) for p in self.prob_bin], 1) if self.prob_bin else np.zeros((T, 0))
        Xc = np.stack([(rng.random(T)[:, None] > np.cumsum(p[Z], 1)).sum(1).clip(max=p.shape[1] - 1).astype(float)
                       for p in self.prob_cat], 1) if self.prob_cat else np.zeros((T, 0))
        return np.concatenate([Xr, Xp, Xb, Xc], 1), Z

    def miss_mask(self, percent):
        """:286-297: round(percent*T) random missing steps per dimension."""
        mask = np.zeros((self.T, self.X_dim), dtype=bool)
        for d in range(self.X_dim):
            mask[self.rng.choice(self.T, int(round(percent * self.T)), replace=False), d] = True
        return mask

    def burst_missmask(self, percent):
        """:299-363: per dimension, bursts of int(U(3,10)) steps until round(T*percent) steps are masked."""
        T = self.T
        num_miss = int(np.round(T * percent))
        out = np.zeros((T, self.X_dim), dtype=bool)
        for d in range(self.X_dim):
            free, burst = list(range(T)), np.zeros(0, dtype=int)
            while free and num_miss > 0:
                t_art = int(self.rng.uniform(3, 10))
                centre = -1
                for _ in range(1000):
                    centre = free[self.rng.integers(len(free))]
                    if not ((centre + t_art / 2 > T) or (centre - t_art / 2 < 0)):
                        break
                new = np.arange(centre - t_art / 2, centre + t_art / 2).astype(int)
                burst = np.unique(np.concatenate([burst, new[(new >= 0) & (new < T)]])).astype(int)
                free = sorted(set(free) - set(burst.tolist()))
                if len(burst) >= num_miss:
                    burst = burst[:num_miss]
                    break
            out[burst, d] = True
        return out

    def generate_toydataset(self, percent_miss=(0.3, 0.5, 0.7), num_masks=10, save=True):
        """:365-441: the data dict (and, if save, <name>.pickle + data_types.csv under save_dir)."""
        X, Z = zip(*(self.generate_samples() for _ in range(self.N)))
        d = {"X": np.stack(X), "Z": np.stack(Z), "trans_mat": self.transmat, "start_prob": self.startprob}
        index = self.rng.permutation(self.N)
        tr, va = int(0.8 * self.N), int(0.1 * self.N + int(0.8 * self.N))
        d.update(train_index=index[:tr], val_index=index[tr:va], test_index=index[va:int(0.1 * self.N + va)], theta=self.theta)
        for p in percent_miss:
            for i in range(num_masks):
                d["mask_%d_%d" % (int(p * 100), i + 1)] = np.stack([self.burst_missmask(p) for _ in range(self.N)])
        if save:
            os.makedirs(self.save_dir, exist_ok=True)
            with open(os.path.join(self.save_dir, self.toy_name + ".pickle"), "wb") as h:
                pickle.dump(d, h, protocol=pickle.HIGHEST_PROTOCOL)
            self.write_types_csv(os.path.join(self.save_dir, "data_types.csv"))
        return d

    def types_rows(self):
        """:443-455: rows of data_types.csv (name, type, dim, nclass, index)."""
        ty = self.x_real_dim * ['real'] + self.x_pos_dim * ['pos'] + self.x_bin_dim * ['bin'] + self.x_cat_dim * ['cat']
        ncl = self.x_real_dim * [1] + self.x_pos_dim * [1] + self.x_bin_dim * [2] + self.x_cat_class[:self.x_cat_dim]
        dim = self.x_real_dim * [1] + self.x_pos_dim * [1] + self.x_bin_dim * [1] + self.x_cat_class[:self.x_cat_dim]
        return [["var" + str(i), t, d, n, i] for i, (t, n, d) in enumerate(zip(ty, ncl, dim))]

    def write_types_csv(self, path):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["name", "type", "dim", "nclass", "index"])
            w.writerows(self.types_rows())

    def types_list(self):
        return [dict(name=r[0], type=r[1], dim=str(r[2]), nclass=str(r[3]), index=str(r[4])) for r in self.types_rows()]
