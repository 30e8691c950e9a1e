"""build_dataloader / build_test_dataloader — the factory names the drivers import (dataset/data_loader.py:57-88, :156-179),
returning the reference's (dataset, dataloader, prefetcher) triple.

Reading image files and augmenting them (YOLODataset, Transforms, mosaic ...) is outside the hot-path scope (SURVEY §8:
dataset tooling is out of scope, and OpenCV is absent from the image): `img_dir` is either the string "synthetic" — the
SyntheticDetectionDataset in the reference's __getitem__ format — or any object with __len__/__getitem__ yielding
(image (h,w,3) uint8, {'bboxes': (n,4) xyxy, 'classes': [n]}, id) items, i.e. a user-supplied dataset; everything
downstream of __getitem__ (fixed_imgsize_collate_fn -> letterbox -> DataPrefetcher) is the mirrored format path."""
from functools import partial

import numpy as np
import torch
from torch.utils.data import DataLoader

from .data_collater import fixed_imgsize_collate_fn, test_dataset_collate_fn
from .data_prefetcher import DataPrefetcher, TestDataPrefetcher
from .synthetic import SyntheticDetectionDataset

__all__ = ["build_dataloader", "build_val_dataloader", "build_test_dataloader"]


def _dataset_from(img_dir, input_dim, seed, length=512):
    if isinstance(img_dir, str):
        if img_dir != "synthetic":
            raise NotImplementedError(
                "image-directory datasets (YOLODataset / TestDataset of the reference) are outside this package's scope: pass "
                "'synthetic' or a dataset object yielding (img uint8 (h,w,3), {'bboxes','classes'}, id)")
        h, w = int(input_dim[0]), int(input_dim[1])
        return SyntheticDetectionDataset(length, img_hw=(max(8, int(h * 0.75) // 8 * 8), w), seed=seed or 7)
    return img_dir


def _seed_worker(worker_id):
    np.random.seed((torch.initial_seed() + worker_id) % 2 ** 31)


def build_dataloader(img_dir, lab_dir, name_path, input_dim, aug_hyp, cache_num, enable_data_aug,
                     seed, batch_size, num_workers, pin_memory, shuffle, drop_last):
    """training loader: DataLoader -> fixed_imgsize_collate_fn(dst_size=input_dim) -> DataPrefetcher on a GPU box"""
    if enable_data_aug:
        raise NotImplementedError("data augmentation (utils/data_aug.py Transforms / mosaic) is outside the hot-path scope")
    dataset = _dataset_from(img_dir, input_dim, seed)
    gen = torch.Generator().manual_seed(seed if seed else 7)
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=bool(shuffle), drop_last=bool(drop_last), num_workers=num_workers,
                        pin_memory=bool(pin_memory), generator=gen, worker_init_fn=_seed_worker,
                        collate_fn=partial(fixed_imgsize_collate_fn, dst_size=input_dim))
    prefetcher = DataPrefetcher(loader) if torch.cuda.is_available() else None
    return dataset, loader, prefetcher


build_val_dataloader = build_dataloader


class _ImagesOnly:
    """test-time view of a detection dataset: (CHW float tensor letterboxed by the collate's companion, resize info)"""

    def __init__(self, ds, input_dim):
        self.ds, self.input_dim = ds, input_dim

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        from ..utils.letterbox import letter_resize_img
        img = self.ds[i][0]
        # training=True: every item gets the full target size, so a batch stacks (test_dataset_collate_fn uses torch.stack)
        out, info = letter_resize_img(img, self.input_dim, training=True)
        return torch.from_numpy(np.ascontiguousarray(out.transpose(2, 0, 1))).float() / 255.0, info


def build_test_dataloader(img_dir, input_dim, batch_size=1, num_workers=0):
    """inference loader: images only, test_dataset_collate_fn, TestDataPrefetcher on a GPU box"""
    base = _dataset_from(img_dir, input_dim, 7, length=64)
    dataset = _ImagesOnly(base, input_dim)
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, drop_last=False, num_workers=num_workers, pin_memory=True,
                        worker_init_fn=_seed_worker, collate_fn=test_dataset_collate_fn)
    prefetcher = TestDataPrefetcher(loader) if torch.cuda.is_available() else None
    return dataset, loader, prefetcher
