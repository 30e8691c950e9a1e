"""Small host helpers the drivers import from `utils` — mirror of the reference's utils/common.py public names
(padding :16-28, is_parallel :31-37, maybe_mkdir :60-64, time_synchronize :67-70, clear_dir :77-82, ...).
Nothing here touches a kernel; they exist so that the reference's import blocks (train_yolov5.py:28-44,
val_yolov5.py) resolve against this package unchanged."""
import numbers
import shutil
import time
import warnings
from pathlib import Path

import numpy as np
import torch

__all__ = ["padding", "is_parallel", "catch_warnnings", "maybe_mkdir", "time_synchronize", "is_exists", "clear_dir",
           "compute_resize_scale", "compute_featuremap_shape", "check_parameters_no_used", "dummy_context"]


def padding(hw, factor=32):
    """round an image size up to the next multiple of `factor`; a scalar means a square (utils/common.py:16-28)"""
    if isinstance(hw, numbers.Real):
        hw = (hw, hw)
    assert len(hw) == 2, "input image size's format should like (h, w)"
    h, w = (int(-(-v // factor) * factor) if v % factor else v for v in hw)
    return h, w


def is_parallel(model):
    """utils/common.py:31-37"""
    return isinstance(model, (torch.nn.parallel.DataParallel, torch.nn.parallel.DistributedDataParallel))


def catch_warnnings(fn):
    def wrapper(instance):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fn(instance)
    return wrapper


def maybe_mkdir(dirname):
    Path(dirname).mkdir(parents=True, exist_ok=True)


def time_synchronize():
    """wall clock after the device has drained (utils/common.py:67-70)"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def is_exists(path_str):
    return Path(path_str).exists()


def clear_dir(dirname):
    """empty directory `dirname` (created if missing; an existing one is removed first, utils/common.py:77-82)"""
    d = Path(dirname)
    if d.exists():
        shutil.rmtree(str(d))
    d.mkdir(parents=True)


def compute_resize_scale(img, min_side, max_side):
    min_side = 800 if min_side is None else min_side
    max_side = 1300 if max_side is None else max_side
    h, w = img.shape[:2]
    scale = min(min_side / h, min_side / w)
    if scale * max(h, w) > max_side:
        scale = min(max_side / h, max_side / w)
    return scale


def compute_featuremap_shape(img_shape, pyramid_level):
    return (np.array(img_shape) - 1) // (2 ** pyramid_level) + 1


def check_parameters_no_used(model):
    unused = [n for n, p in model.named_parameters() if p.grad is None]
    if unused:
        print("=" * 100)
        print(unused)
        print("=" * 100)


class dummy_context:
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        pass

    def step(self):
        pass
