"""Host->device prefetcher — mirror of dataset/data_prefetcher.py:6-106: the next batch is copied on a side HIP
stream while the current step computes; `next()` makes the compute stream wait for that copy."""
import torch

__all__ = ["DataPrefetcher", "TestDataPrefetcher"]


class DataPrefetcher:

    def __init__(self, loader):
        self.loader = iter(loader)
        self.stream = torch.cuda.Stream()
        self.preload()

    def preload(self):
        try:
            out = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = self.next_resize_info = self.next_img_id = None
            return
        self.next_input, self.next_target = out["img"], out["ann"]
        self.next_resize_info, self.next_img_id = out['resize_info'], out['img_id']
        with torch.cuda.stream(self.stream):
            self.next_input = self.next_input.cuda(non_blocking=True)
            self.next_target = self.next_target.cuda(non_blocking=True)

    def next(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        inp, target = self.next_input, self.next_target
        resize_info, img_id = self.next_resize_info, self.next_img_id
        if inp is not None:
            inp.record_stream(torch.cuda.current_stream())
        if target is not None:
            target.record_stream(torch.cuda.current_stream())
        self.preload()
        return {'img': inp, 'ann': target, 'resize_info': resize_info, 'img_id': img_id}


class TestDataPrefetcher:

    def __init__(self, loader):
        self.loader = iter(loader)
        self.stream = torch.cuda.Stream()
        self.preload()

    def preload(self):
        try:
            out = next(self.loader)
        except StopIteration:
            self.next_input = self.next_resize_info = None
            return
        self.next_input, self.next_resize_info = out["img"], out["resize_info"]
        with torch.cuda.stream(self.stream):
            self.next_input = self.next_input.cuda(non_blocking=True)

    def next(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        inp, info = self.next_input, self.next_resize_info
        if inp is not None:
            inp.record_stream(torch.cuda.current_stream())
        self.preload()
        return {'img': inp, 'resize_info': info}
