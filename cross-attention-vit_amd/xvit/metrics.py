"""On-device epoch statistics for the reference's log_stats (model_cross.py:243-255 -> utils.py:18-62).

The reference computes accuracy / precision / recall / specificity / F1 / NPV with torchmetrics and reads each back with
.item(), plus an AUROC, on EVERY step (seven host syncs after a step of a few tens of milliseconds), and logs them with
on_epoch=True, i.e. as batch-size-weighted epoch means.  BinaryEpochMetrics.update() is one tiny kernel launch and never
touches the host; compute() reads the 16-double state once (per epoch) and returns the same seven epoch values under the
reference's log names, plus the pooled confusion counts.

    stats = BinaryEpochMetrics(device)
    for x, y in loader:
        logits, loss = model(x, y); ...; stats.update(logits, y)       # no sync
    print(stats.compute("train"))                                      # {'train_acc': ..., 'train_auc_roc': ..., 'train_confusion': {...}}
    stats.reset()
"""
from __future__ import annotations

import torch

from . import _lib

KEYS = ("acc", "prec", "rec", "spec", "f1", "npv", "auc_roc")     # log-name suffixes of model_cross.py:246-255
STATE = 16


class BinaryEpochMetrics:
    def __init__(self, device):
        self.state = torch.zeros(STATE, dtype=torch.float64, device=device)

    def reset(self):
        self.state.zero_()

    @torch.no_grad()
    def update(self, logits: torch.Tensor, labels: torch.Tensor):
        if not logits.is_cuda:
            raise RuntimeError("BinaryEpochMetrics: logits must live on the GPU (there is no CPU path)")
        if logits.dim() != 2 or logits.shape[1] != 2:
            raise ValueError(f"binary classification expected, got logits of shape {tuple(logits.shape)}")
        lg = logits.detach()
        if lg.dtype != torch.float32 or lg.stride(1) != 1:
            lg = lg.float().contiguous()
        lb = labels.detach()
        if lb.dtype != torch.int64 or not lb.is_contiguous():
            lb = lb.to(torch.int64).contiguous()
        st = torch.cuda.current_stream(lg.device).cuda_stream
        _lib.check(_lib.load().xvit_binary_metrics_step(lg.data_ptr(), lg.stride(0), lb.data_ptr(), lg.shape[0], 2, self.state.data_ptr(), st),
                   "xvit_binary_metrics_step")

    def compute(self, name: str = "", sync_dist: bool = True, group=None) -> dict:
        """The epoch values (ONE host sync).  With torch.distributed initialised and sync_dist=True the state is summed
        over `group` (default: the world) first: the exact batch-size-weighted mean over all ranks, the counterpart of
        the reference's sync_dist=True."""
        s = self.state
        if sync_dist and torch.distributed.is_available() and torch.distributed.is_initialized():
            s = s.clone()
            torch.distributed.all_reduce(s, group=group)
        return summarize(s.cpu(), name)


def summarize(state: torch.Tensor, name: str = "") -> dict:
    """state (16 doubles on the host) -> {f'{name}_acc': ..., ..., f'{name}_confusion': {...}}"""
    s = state.double().tolist()
    n = s[4]
    pre = f"{name}_" if name else ""
    out = {pre + k: (s[6 + i] / n if n > 0 else 0.0) for i, k in enumerate(KEYS)}
    out[pre + "confusion"] = {"tn": int(s[0]), "fp": int(s[1]), "fn": int(s[2]), "tp": int(s[3]), "samples": int(n), "steps": int(s[5])}
    return out


class EpochStatsMixin:
    """log_stats of the reference models (model_cross.py:243-255; modelv3.py has the same method): same log names and
    epoch values, accumulated on the device.  The reference computes seven torchmetrics values with seven host syncs per
    step and logs them on_epoch; here a step is one tiny launch and the epoch values are logged once, from
    on_train_epoch_end / on_validation_epoch_end (or read with epoch_stats(name))."""

    def log_stats(self, name, logits, labels):
        stats = self.__dict__.setdefault("_stats", {})
        if name not in stats or stats[name].state.device != logits.device:
            stats[name] = BinaryEpochMetrics(logits.device)
        stats[name].update(logits, labels)

    def epoch_stats(self, name, reset=True):
        """{f'{name}_acc', _prec, _rec, _spec, _f1, _npv, _auc_roc, _confusion}: one host sync (summed over the ranks)."""
        acc = self.__dict__.get("_stats", {}).get(name)
        if acc is None:
            return {}
        out = acc.compute(name)
        if reset:
            acc.reset()
        return out

    def _log_epoch_stats(self, name):
        for k, v in self.epoch_stats(name).items():
            if not k.endswith("_confusion"):
                self.log(k, v, sync_dist=False)        # epoch_stats already reduced over the ranks

    def on_train_epoch_end(self):
        self._log_epoch_stats("train")

    def on_validation_epoch_end(self):
        self._log_epoch_stats("val")

    # ---- test loop surface of both reference models (model_cross.py:294-308, modelv3.py:229-243): logits / targets on the host
    def on_test_epoch_start(self):
        self.test_logits = []
        self.test_targets = []

    def test_step(self, batch, batch_idx):
        x, labels = batch
        logits, _ = self(x, labels)
        self.test_logits.append(logits.detach().cpu())
        self.test_targets.append(labels.cpu())

    def on_test_epoch_end(self):
        self.test_logits = torch.cat(self.test_logits)
        self.test_targets = torch.cat(self.test_targets)
