"""Depth metrics with Keras ``Mean`` semantics -- same class names as the
reference ``metrics.py``; inputs are device tensors, accumulation stays on the
device (no host sync per batch).  ``state()`` / ``load_state()`` expose the
(total, count) pair that the multi-GPU eval all-gathers over RCCL.
"""
from __future__ import annotations

import torch


def masked_reduce_mean(err, gt_depth):
    """metrics.py:3-5: mean of err over gt > 1e-6 (multiply_no_nan), / max(count, 1)."""
    mask = (gt_depth > 1e-6)
    e = torch.where(mask, err, torch.zeros_like(err))
    return e.sum() / torch.clamp_min(mask.sum().to(torch.float32), 1.0)


class _Mean:
    """tf.keras.metrics.Mean: total += value, count += 1 per update."""

    def __init__(self, name):
        self.name = name
        self.total = None
        self.count = 0

    def _update(self, value):
        value = value.detach().to(torch.float32)
        self.total = value if self.total is None else self.total + value
        self.count += 1

    def result(self):
        if self.total is None:
            return torch.zeros(())
        return self.total / float(max(self.count, 1))

    def reset_state(self):
        self.total = None
        self.count = 0

    def state(self, device):
        t = self.total if self.total is not None else torch.zeros((), device=device)
        return torch.stack([t.to(device), torch.tensor(float(self.count), device=device)])

    def load_state(self, total, count):
        self.total = total
        self.count = int(count)


class RootMeanSquaredError(_Mean):
    def __init__(self, name='RMSE', **kwargs):
        super().__init__(name)

    def update_state(self, y_true, y_pred, sample_weight=None):
        self._update(torch.sqrt(masked_reduce_mean(torch.square(y_true - y_pred), y_true)))     # metrics.py:13-15


class RootMeanSquaredLogError(_Mean):
    def __init__(self, name='RMSE_log', **kwargs):
        super().__init__(name)

    def update_state(self, y_true, y_pred, sample_weight=None):
        lt = torch.log(y_true + 1e-6)
        lp = torch.log(y_pred + 1e-6)
        self._update(torch.sqrt(masked_reduce_mean(torch.square(lt - lp), lt)))      # mask on the LOG (:24-28)


class AbsRelError(_Mean):
    def __init__(self, name='AbsRel', **kwargs):
        super().__init__(name)

    def update_state(self, y_true, y_pred, sample_weight=None):
        self._update(masked_reduce_mean(torch.abs(y_true - y_pred) / (y_true + 1e-6), y_true))  # :37-39


class SqRelError(_Mean):
    def __init__(self, name='SqRel', **kwargs):
        super().__init__(name)

    def update_state(self, y_true, y_pred, sample_weight=None):
        self._update(masked_reduce_mean(torch.square(y_true - y_pred) / (y_true + 1e-6), y_true))  # :48-50


class ThresholdRelError(_Mean):
    def __init__(self, threshold, name='Delta', **kwargs):
        self.threshold = threshold
        super().__init__(name + str(threshold))

    def update_state(self, y_true, y_pred, sample_weight=None):
        thresh = torch.maximum(y_true / y_pred, y_pred / y_true)
        err = (thresh < 1.25 ** self.threshold).to(torch.float32)
        self._update(masked_reduce_mean(err, y_true))                                   # :60-63


def default_metrics():
    """The list compiled at main.py:127-130, in that order."""
    return [AbsRelError(), SqRelError(), RootMeanSquaredError(), RootMeanSquaredLogError(),
            ThresholdRelError(1), ThresholdRelError(2), ThresholdRelError(3)]
