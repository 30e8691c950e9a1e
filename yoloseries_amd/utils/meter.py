"""Meters of the training log line.  Public surface of utils/meter.py:16-103 (`AverageMeter` with
median / avg / global_avg / latest / total, `MeterBuffer.update(**loss_dict)` as train_yolov5.py:435 calls it);
the storage is this package's own: a preallocated float64 ring for the window plus a compensated running sum for
the global mean, so an update costs one array write and no list is rebuilt per query."""
import numpy as np
import torch

__all__ = ["AverageMeter", "MeterBuffer"]


def _as_float(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return float(np.asarray(x, dtype=np.float64).reshape(-1)[0]) if np.ndim(x) else float(x)


class AverageMeter:
    """scalar series: statistics over the last `window_size` values and over everything seen"""

    def __init__(self, window_size=50):
        self.window_size = int(window_size)
        self._ring = np.zeros(max(self.window_size, 1), dtype=np.float64)
        self._seen = 0            # values since the last clear() — decides how much of the ring is live
        self._n = 0               # values since the last reset()
        self._sum = 0.0           # Kahan-compensated sum of those values
        self._comp = 0.0
        self._last = None

    def _window(self):
        k = min(self._seen, len(self._ring))
        return self._ring[:k]

    def update(self, value):
        v = _as_float(value)
        self._ring[self._seen % len(self._ring)] = v
        self._seen += 1
        self._n += 1
        y = v - self._comp
        t = self._sum + y
        self._comp = (t - self._sum) - y
        self._sum = t
        self._last = value

    @property
    def median(self):
        w = self._window()
        return float(np.median(w)) if w.size else float("nan")

    @property
    def avg(self):
        w = self._window()
        return float(w.mean()) if w.size else float("nan")      # nan on an empty window, like the reference

    @property
    def global_avg(self):
        return self._sum / max(self._n, 1e-5)

    @property
    def latest(self):
        return self._last if self._seen else None

    @property
    def total(self):
        return self._sum

    def clear(self):
        """forget the window, keep the global sums"""
        self._seen = 0

    def reset(self):
        self.clear()
        self._n, self._sum, self._comp, self._last = 0, 0.0, 0.0, None


class MeterBuffer(dict):
    """name -> AverageMeter, created when a name is first touched"""

    def __init__(self, window_size=20):
        super().__init__()
        self.window_size = window_size

    def __missing__(self, key):
        m = self[key] = AverageMeter(self.window_size)
        return m

    def update(self, values=None, **kwargs):
        for src in (values or {}), kwargs:
            for name, v in src.items():
                self[name].update(v)

    def get_filtered_meter(self, filter_key="time"):
        return {name: m for name, m in self.items() if filter_key in name}

    def reset(self):
        for m in self.values():
            m.reset()

    def clear_meters(self):
        for m in self.values():
            m.clear()
