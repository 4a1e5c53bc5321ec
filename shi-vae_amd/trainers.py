"""Train loop with the reference's API (``src/models/trainers.py`` Trainer, ``src/models/train.py``
BaseTrainer / KL_Annealing) around the HIP step.  What is kept: constructor signature, ``train``,
``train_epoch``, ``valid_epoch``, ``load_checkpoint``, ``preprocess_batch``, the KL / temperature
schedules, checkpoint dict layout ``{'state_dict','epoch','optimizer','params'}`` and the ';'-separated
loss CSVs, the imputation-error evaluation of ``valid_epoch`` (every ``args.plot_every`` epochs, result in
``trainer.last_avg_error``).  What is dropped on purpose: the matplotlib plots."""
import csv
import os

import numpy as np
import torch

from . import utils
from .optim import FusedClipAdam

shivae_losses = ['n_elbo', 'nll_loss', 'nll_real', 'nll_pos', 'nll_bin', 'nll_cat', 'kld', 'kld_q_z', 'kld_q_s']


def update_loss_dict(old_loss_dict, new_loss_dict):
    """train.py:25-41."""
    for loss in old_loss_dict.keys():
        v = new_loss_dict[loss]
        old_loss_dict[loss].append(v.detach().cpu().numpy() if torch.is_tensor(v) else v)
    return old_loss_dict


class KL_Annealing:
    """train.py:44-69: beta = (epoch/annealing_epochs)^2 clipped to [0,1]."""

    def __init__(self, annealing_epochs):
        self.annealing_epochs = annealing_epochs

    def update_beta(self, epoch):
        if self.annealing_epochs > 0 and epoch < self.annealing_epochs:
            beta = (epoch * 1. / self.annealing_epochs) ** 2
        else:
            beta = 1.
        return float(np.clip(beta, 0.0, 1.0))


class LinearTempScheduler:
    """trainers.py:23-53: tau = max(temp_end, temp_init - slope*epoch)."""

    def __init__(self, init_temp=3, end_temp=0.01, annealing_epochs=10):
        self.temp_init, self.temp_end, self.annealing_epochs = init_temp, end_temp, annealing_epochs
        self.slope = (init_temp - end_temp) / annealing_epochs

    def update_temp(self, epoch):
        return max(self.temp_init - self.slope * epoch, self.temp_end)


class ExpTempScheduler:
    """trainers.py:56-85."""

    def __init__(self, end_temp, annealing_epochs=20):
        self.temp_end, self.annealing_epochs = end_temp, annealing_epochs
        self.alpha = -np.log(end_temp) / annealing_epochs

    def update_temp(self, epoch):
        return max(float(np.exp(-self.alpha * epoch)), self.temp_end)


class EarlyStopping:
    """lib/early_stopping.py:8-46: patience counter on the validation -ELBO."""

    def __init__(self, patience=50, min_delta=0.2):
        self.patience, self.min_delta = patience, min_delta
        self.best, self.counter = None, 0

    def stop(self, val_loss):
        if self.best is None or self.best - val_loss > self.min_delta:
            self.best, self.counter = val_loss, 0
            return False
        self.counter += 1
        return self.counter >= self.patience


class Trainer:
    """trainers.py:88-338 / train.py:72-380.  ``args`` needs: l_rate, n_epochs, kl_annealing_epochs, clip,
    save_every, print_every, save_model, result_dir, model (nickname), restore (optional)."""

    def __init__(self, model, optimizer, args, scaler=None):
        self.model, self.optimizer, self.scaler, self.args = model, optimizer, scaler, args
        self.device = model.device
        self.lr, self.init_epoch, self.n_epochs = args.l_rate, 1, args.n_epochs
        self.annealing_weight = KL_Annealing(args.kl_annealing_epochs)
        self.clip = args.clip
        self.early_stopping = EarlyStopping(patience=50, min_delta=0.2)
        self.save_every, self.print_every = args.save_every, args.print_every
        self.save_model, self.result_dir = args.save_model, args.result_dir
        self.model_name = args.result_dir.split("/")[-1]
        self.train_csv = os.path.join(args.result_dir, self.model_name) + '_train.csv'
        self.val_csv = os.path.join(args.result_dir, self.model_name) + '_val.csv'
        self.loss_list = shivae_losses
        self.csv_cols = ['epoch'] + self.loss_list
        self.temp = 3
        self.temp_scheduler = LinearTempScheduler(3, 0.1, max(1, round(self.n_epochs * 0.9)))  # trainers.py:109-111
        self.print_str = "-ELBO: {n_elbo:.3f} | KLD Loss: {kld:.3f}  NLL Loss: {nll_loss:.3f} |  Real Loss {nll_real:.3f}" \
                         "  Pos Loss {nll_pos:.3f}  Binary Loss {nll_bin:.3f}  Categ Loss {nll_cat:.3f}  |" \
                         "  KL_q_z {kld_q_z:.3f}  KL_q_s {kld_q_s:.3f}"
        os.makedirs(self.result_dir, exist_ok=True)
        for path in (self.train_csv, self.val_csv):
            if not (os.path.exists(path) and getattr(args, "restore", False)):
                with open(path, "w", newline="") as f:
                    csv.writer(f, delimiter=';').writerow(self.csv_cols)

    # ---- helpers
    def init_loss_dict(self):
        return {k: [] for k in self.loss_list}, {k: [] for k in self.loss_list}

    def _append_csv(self, path, loss_dict, epoch):
        with open(path, "a", newline="") as f:
            csv.writer(f, delimiter=';').writerow([epoch] + [float(loss_dict[k]) for k in self.loss_list])

    def load_checkpoint(self, model_dict):
        """train.py:226-235."""
        self.model.load_state_dict(model_dict['state_dict'])
        self.optimizer.load_state_dict(model_dict['optimizer'])
        self.init_epoch = model_dict['epoch'] + 1

    def save(self, dict_params, epoch):
        path = os.path.join(self.result_dir, self.model_name) + "_" + str(epoch) + ".pth"
        torch.save(dict_params, path)
        print('Saved model to ' + path)

    def save_best(self, dict_params):
        path = os.path.join(self.result_dir, self.model_name) + "_best.pth"
        torch.save(dict_params, path)
        print('Saved model to ' + path)

    def _ckpt(self, epoch):
        return {'state_dict': self.model.state_dict(), 'epoch': epoch, 'optimizer': self.optimizer.state_dict(),
                'params': self.args}

    def preprocess_batch(self, data, mask):
        """train.py:333-362: scale, B,T,D -> T,B,D, float32 / bool tensors on the model's device."""
        if self.scaler is not None and hasattr(self.scaler, "transform_to_device"):
            # one HIP pass: scale + one-hot + mask replication + B,T,D -> T,B,X + zero fill (csrc/prep.hip)
            return self.scaler.transform_to_device(data, mask, transpose=True, zero_fill=True)
        if self.scaler is not None:
            data, mask = self.scaler.transform(data, mask)
        data = torch.as_tensor(np.asarray(data).transpose(1, 0, 2)).float().to(self.model.flat_params.device)
        mask = torch.as_tensor(np.asarray(mask).transpose(1, 0, 2), dtype=torch.bool).to(self.model.flat_params.device)
        return data, mask

    # ---- epochs
    def train_epoch(self, train_loader, epoch):
        """trainers.py:181-249 (the hot loop :197-226)."""
        epoch_loss, _ = self.init_loss_dict()
        beta = self.annealing_weight.update_beta(epoch)
        self.model.train()
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1 and hasattr(train_loader, "__len__"):
            # every rank must run the same number of steps: each step holds collectives
            nb = torch.tensor([len(train_loader), -len(train_loader)], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                nb = nb.to(self.model.flat_params.device)
            dist.all_reduce(nb, op=dist.ReduceOp.MIN)
            lo, hi = int(nb[0]), -int(nb[1])
            if lo != hi:
                raise RuntimeError("data-parallel train_epoch: the ranks' loaders have between %d and %d batches; "
                                   "shard the dataset so that every rank gets the same number" % (lo, hi))
        prev_overlap = self.model.overlap_allreduce
        self.model.overlap_allreduce = True  # (this loop calls allreduce_grads() after every backward)
        try:
            return self._train_batches(train_loader, epoch, epoch_loss, beta, world)
        finally:
            self.model.overlap_allreduce = prev_overlap

    def _train_batches(self, train_loader, epoch, epoch_loss, beta, world):
        import torch.distributed as dist
        for data, mask, _attrs in train_loader:
            data, mask = self.preprocess_batch(data, mask)
            data = utils.zero_fill(data, mask)
            self.optimizer.zero_grad()
            inv_batch = None
            if world > 1:
                # data parallel over the batch: every rank scales by 1/B_global (ragged last batches: the summed
                # count), so the all-reduced SUM of the flat gradients is the global-batch gradient that
                # clip_grad_norm_ sees in the reference (trainers.py:222)
                # (the count is known on the host: a host tensor for gloo -- a device tensor handed to gloo while
                # kernels are in flight is the pattern that hung two ranks on one GPU)
                cnt = torch.tensor([float(data.shape[1])], dtype=torch.float64)
                if dist.get_backend() == "nccl":
                    cnt = cnt.to(data.device)
                dist.all_reduce(cnt)
                inv_batch = 1.0 / float(cnt.item())
            loss_dict = self.model(data, mask=mask, beta=beta, temp=self.temp, inv_batch=inv_batch)
            n_elbo = loss_dict['n_elbo']
            n_elbo.backward()
            self.model.allreduce_grads()
            if world > 1:
                # the 9 logged scalars are per-rank shares of the global means: sum them; a NaN on any rank becomes
                # a NaN on all of them, so every rank takes the same branch below (none is left inside a collective)
                vals = torch.stack([loss_dict[k].detach() for k in self.loss_list])
                if dist.get_backend() != "nccl":
                    vals = vals.cpu()  # (blocking copy = the wait for the step; see above)
                dist.all_reduce(vals)
                loss_dict = {k: vals[i] for i, k in enumerate(self.loss_list)}
                n_elbo = loss_dict['n_elbo']
            if np.isnan(n_elbo.detach().cpu().numpy()):
                print('Encountered NaN, stopping training. Please check the learning_rate settings and the momentum.')
                break
            if isinstance(self.optimizer, FusedClipAdam):
                self.optimizer.step(clip=self.clip)
            else:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
                self.optimizer.step()
            epoch_loss = update_loss_dict(epoch_loss, loss_dict)
        if epoch % self.save_every == 0 and self.save_model:
            self.save(self._ckpt(epoch), epoch)
        epoch_loss = {k: float(np.mean(v)) for k, v in epoch_loss.items()}
        if epoch % self.print_every == 0:
            print(("Epoch: {}/{}\n" + self.print_str).format(epoch, self.n_epochs, **epoch_loss))
        return epoch_loss

    def valid_epoch(self, valid_loader, epoch):
        """trainers.py:251-338 (loss part :266-280; the imputation-error plots are out of scope)."""
        _, epoch_loss = self.init_loss_dict()
        beta = self.annealing_weight.update_beta(epoch)
        self.model.eval()
        with torch.no_grad():
            for batch in valid_loader:
                data, mask = self.preprocess_batch(batch[0], batch[1])
                data = utils.zero_fill(data, mask)
                epoch_loss = update_loss_dict(epoch_loss, self.model(data, mask=mask, beta=beta, temp=self.temp))
        epoch_loss = {k: float(np.mean(v)) for k, v in epoch_loss.items()}
        if epoch % self.print_every == 0:
            print(("Epoch (valid): {}/{}\n" + self.print_str).format(epoch, self.n_epochs, **epoch_loss))
        plot_every = int(getattr(self.args, "plot_every", 0) or 0)
        if plot_every > 0 and epoch % plot_every == 0:
            self.last_avg_error = self.imputation_error(valid_loader)
        return self.early_stopping.stop(epoch_loss['n_elbo']), epoch_loss

    def imputation_error(self, loader):
        """trainers.py:286-333 without the plots: reconstruct the evaluation view of the loader's dataset
        (data with the artificial mask applied), de-normalise, and score the artificially removed entries ->
        (error_miss, corr_miss) dicts per attribute (lib/loss.py:71-201).  None if the dataset has no evaluation
        view or there is no scaler."""
        from .metrics import compute_avg_error
        ds = getattr(loader, "dataset", None)
        if ds is None or not hasattr(ds, "set_eval") or self.scaler is None:
            return None
        was_training = self.model.training
        self.model.eval()
        ds.set_eval(True)
        try:
            data_hat, masks, fulls, arts = [], [], [], []
            with torch.no_grad():
                for data, mask, data_full, mask_artificial, _attrs in loader:
                    data, mask = self.preprocess_batch(data, mask)
                    data_zf = utils.zero_fill(data, mask)
                    (_likes, x_hat), _lat = self.model.reconstruction(data_zf, temp=self.temp)
                    data_hat.append(x_hat)
                    masks.append(mask)
                    fulls.append(np.asarray(data_full))
                    arts.append(np.asarray(mask_artificial))
            data_hat = torch.cat(data_hat, dim=1)                      # [T, sum B, X]  (np.hstack in the reference)
            mask = torch.cat(masks, dim=1)
            data_full, mask_artificial = np.vstack(fulls), np.vstack(arts).astype(bool)
            data_hat_denorm, mask = self.scaler.inverse_transform(data_hat, mask=mask, transpose=True)
            real_mask = mask_artificial ^ mask.astype(bool)
            data_full_zf = np.where(real_mask, np.zeros_like(data_full), data_full)
            _, error_miss, _, corr_miss = compute_avg_error(data_full_zf, data_hat_denorm, mask_artificial,
                                                            self.model.types_list, img_path=self.result_dir)
        finally:
            ds.set_eval(False)
            self.model.train(was_training)
        return error_miss, corr_miss

    def train(self, train_loader, valid_loader):
        """trainers.py:118-179."""
        n_elbo_list = []
        epoch = 0
        self.train_loss, self.val_loss = self.init_loss_dict()
        for epoch in range(self.init_epoch, self.n_epochs + 1):
            print('Epoch {}/{}'.format(epoch, self.n_epochs))
            tr = self.train_epoch(train_loader, epoch)
            self._append_csv(self.train_csv, tr, epoch)
            self.train_loss = update_loss_dict(self.train_loss, tr)
            if all(tr["n_elbo"] < np.array(n_elbo_list)) and self.save_model:
                self.save_best(self._ckpt(epoch))
            n_elbo_list.append(tr["n_elbo"])
            stop, va = self.valid_epoch(valid_loader, epoch)
            self._append_csv(self.val_csv, va, epoch)
            self.val_loss = update_loss_dict(self.val_loss, va)
            if stop:
                break
            self.model.train()
            self.temp = self.temp_scheduler.update_temp(epoch)
        self.train_loss = {k: float(np.mean(v)) for k, v in self.train_loss.items()}
        self.val_loss = {k: float(np.mean(v)) for k, v in self.val_loss.items()}
        if self.save_model:
            self.save(self._ckpt(epoch), epoch)
        return self.train_loss, self.val_loss
