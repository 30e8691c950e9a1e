"""Host -> device prefetch on a side HIP stream (the role of dataset/data_prefetcher.py:6-106 in the reference's
training loop, train_yolov5.py:458-497): while a step computes, the next batch is already being copied; `next()`
orders the compute stream behind that copy and marks the tensors as used on it."""
import torch

__all__ = ["DataPrefetcher", "TestDataPrefetcher"]


class _Prefetcher:
    """keys in `tensor_keys` are moved to the device, every other entry of the batch dict is passed through"""
    tensor_keys = ()
    all_keys = ()

    def __init__(self, loader):
        self._it = iter(loader)
        self.stream = torch.cuda.Stream()
        self._staged = None
        self.preload()

    def preload(self):
        batch = next(self._it, None)
        if batch is None:
            self._staged = None
            return
        with torch.cuda.stream(self.stream):
            self._staged = {k: (batch[k].cuda(non_blocking=True) if k in self.tensor_keys else batch[k]) for k in self.all_keys}

    def next(self):
        current = torch.cuda.current_stream()
        current.wait_stream(self.stream)
        out = self._staged if self._staged is not None else {k: None for k in self.all_keys}
        for k in self.tensor_keys:
            if out[k] is not None:
                out[k].record_stream(current)
        self.preload()
        return out


class DataPrefetcher(_Prefetcher):
    tensor_keys = ('img', 'ann')
    all_keys = ('img', 'ann', 'resize_info', 'img_id')


class TestDataPrefetcher(_Prefetcher):
    tensor_keys = ('img',)
    all_keys = ('img', 'resize_info')
