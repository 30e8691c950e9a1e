"""Batch assembly — the tensor format between a dataset and the hot path, as the reference's
dataset/data_collater.py:16-82 produces it:

    'img'          (B, 3, H, W) float32 in [0, 1], channel order as loaded
    'ann'          (B, maxbox, 6) float32 rows [xmin, ymin, xmax, ymax, cls, index of the image in the batch], -1 padded
    'resize_info'  one letterbox record per image (utils/letterbox.py)
    'img_id'       the dataset's ids

`YOLOV5Loss` / `YOLOXLoss` and the evaluators of this package consume exactly this (loss/yolov5_loss.py:30-60)."""
import numpy as np
import torch

from ..utils.letterbox import letter_resize_bbox, letter_resize_img

__all__ = ['fixed_imgsize_collate_fn', 'test_dataset_collate_fn', 'normal_normalization']


def normal_normalization(img):
    """uint8 HWC image -> CHW tensor scaled to [0, 1] (float64 like the reference; the batch tensor is float32)"""
    return torch.from_numpy(img / 255.0).permute(2, 0, 1).contiguous()


def _annotation_rows(ann, info, index):
    """(n, 6) float32 rows of one image, boxes moved into the letterboxed frame"""
    n = len(ann['classes'])
    if len(ann['bboxes']) != n:
        raise AssertionError("every box needs a class")
    rows = torch.empty(n, 6)
    if n:
        rows[:, :4] = torch.from_numpy(np.asarray(letter_resize_bbox(ann['bboxes'], info), dtype=np.float64)).float()
        rows[:, 4] = torch.as_tensor([float(c) for c in ann['classes']])
        rows[:, 5] = float(index)
    return rows


def fixed_imgsize_collate_fn(data_in, dst_size):
    """data_in: sequence of (image (h,w,3) uint8, {'bboxes': (n,4) xyxy, 'classes': n}, image id); dst_size: [h, w]"""
    first = data_in[0][0]
    assert first.ndim == 3 and first.shape[-1] == 3, f"data's formate should be (h, w, 3), but got {first.shape}"
    batch = torch.zeros(len(data_in), 3, dst_size[0], dst_size[1])
    infos, rows, ids = [], [], []
    for index, (img, ann, img_id) in enumerate(data_in):
        boxed, info = letter_resize_img(img, dst_size)
        batch[index] = normal_normalization(boxed)
        infos.append(info)
        rows.append(_annotation_rows(ann, info, index))
        ids.append(img_id)
    ann_out = torch.full((len(data_in), max(len(r) for r in rows), 6), -1.0)      # -1 rows = padding
    for index, r in enumerate(rows):
        ann_out[index, :len(r)] = r
    return {'img': batch, 'ann': ann_out, 'resize_info': infos, 'img_id': ids}


def test_dataset_collate_fn(data_in):
    """items: (already letterboxed (3,h,w) tensor, letterbox record)"""
    img = torch.stack([item[0].to(torch.float32) for item in data_in])
    return {'img': img, 'resize_info': [item[1] for item in data_in]}
