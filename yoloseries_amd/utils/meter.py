"""Windowed meters for the training log — mirror of utils/meter.py:16-103 (AverageMeter, MeterBuffer)."""
import functools
from collections import defaultdict, deque

import numpy as np
import torch

__all__ = ["AverageMeter", "MeterBuffer"]


class AverageMeter:
    """latest / windowed median & mean / global mean of a scalar series"""

    def __init__(self, window_size=50):
        self._deque = deque(maxlen=window_size)
        self._total, self._count = 0.0, 0

    def update(self, value):
        self._deque.append(value)
        self._count += 1
        self._total += value

    @property
    def median(self):
        return np.median(np.array(list(self._deque)))

    @property
    def avg(self):
        return np.array(list(self._deque)).mean()         # nan on an empty window, like the reference

    @property
    def global_avg(self):
        return self._total / max(self._count, 1e-5)

    @property
    def latest(self):
        return self._deque[-1] if self._deque else None

    @property
    def total(self):
        return self._total

    def reset(self):
        self._deque.clear()
        self._total, self._count = 0.0, 0

    def clear(self):
        self._deque.clear()


class MeterBuffer(defaultdict):
    """dict of AverageMeters created on first use; update(**loss_dict) as train_yolov5.py:435 does"""

    def __init__(self, window_size=20):
        super().__init__(functools.partial(AverageMeter, window_size=window_size))

    def reset(self):
        for v in self.values():
            v.reset()

    def get_filtered_meter(self, filter_key="time"):
        return {k: v for k, v in self.items() if filter_key in k}

    def update(self, values=None, **kwargs):
        values = dict(values or {})
        values.update(kwargs)
        for k, v in values.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            self[k].update(v)

    def clear_meters(self):
        for v in self.values():
            v.clear()
