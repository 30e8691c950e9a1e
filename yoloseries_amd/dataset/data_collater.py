"""Collate functions — mirror of dataset/data_collater.py:16-82.

A batch is ``{'img': (B,3,H,W) float32 in [0,1] (channel order as loaded), 'ann': (B,maxbox,6) float32
[xmin,ymin,xmax,ymax,cls,img_idx] padded with -1, 'resize_info': [letterbox dict per image], 'img_id': [...]}``:
exactly what ``YOLOV5Loss`` / the evaluators of this package consume (loss/yolov5_loss.py:30-60)."""
import numpy as np
import torch

from ..utils.letterbox import letter_resize_bbox, letter_resize_img

__all__ = ['fixed_imgsize_collate_fn', 'test_dataset_collate_fn', 'normal_normalization']


def normal_normalization(img):
    """(h,w,3) uint8 -> (3,h,w) float64 tensor in [0,1] (dataset/data_collater.py:16-17; the batch tensor it is
    written into is float32)"""
    return torch.from_numpy(img / 255.0).permute(2, 0, 1).contiguous()


def fixed_imgsize_collate_fn(data_in, dst_size):
    """:param data_in: sequence of (image (h,w,3) uint8, {'bboxes': [[xmin,ymin,xmax,ymax]...], 'classes': [...]}, image id)
    :param dst_size: [h, w] of the batch (dataset/data_collater.py:20-64)"""
    assert data_in[0][0].ndim == 3 and data_in[0][0].shape[-1] == 3, \
        f"data's formate should be (h, w, 3), but got {data_in[0][0].shape}"
    batch_size = len(data_in)
    imgs = [d[0] for d in data_in]
    anns = [d[1] for d in data_in]
    img_ids = [d[2] for d in data_in]
    imgs_out = torch.zeros(batch_size, 3, dst_size[0], dst_size[1])
    boxes_num = [len(ann['bboxes']) for ann in anns]
    # -1 marks padding rows; the last column is the image index inside the batch (used by the target assignment)
    anns_out = torch.ones(batch_size, max(boxes_num), 6) * -1
    resize_infos = []
    for b in range(batch_size):
        ann_bboxes, ann_classes = anns[b]['bboxes'], anns[b]['classes']
        assert len(ann_bboxes) == len(ann_classes)
        img, resize_info = letter_resize_img(imgs[b], dst_size)
        imgs_out[b] = normal_normalization(img)
        resize_infos.append(resize_info)
        if len(ann_classes) > 0:
            boxes = letter_resize_bbox(ann_bboxes, resize_info)
            n = len(ann_classes)
            anns_out[b, :n, :4] = torch.from_numpy(np.asarray(boxes, dtype=np.float64)).float()
            anns_out[b, :n, 4] = torch.as_tensor([float(c) for c in ann_classes])
            anns_out[b, :n, 5] = b
    return {'img': imgs_out, 'ann': anns_out, 'resize_info': resize_infos, 'img_id': img_ids}


def test_dataset_collate_fn(data_in):
    """items are (tensor (3,h,w) already letterboxed, resize_info) — dataset/data_collater.py:67-82"""
    batch_size = len(data_in)
    imgs = [d[0] for d in data_in]
    infoes = [d[1] for d in data_in]
    h, w = imgs[0].shape[1:]
    img_out = torch.ones(batch_size, 3, h, w)
    for i in range(batch_size):
        img_out[i] = imgs[i]
    return {'img': img_out, 'resize_info': list(infoes)}
