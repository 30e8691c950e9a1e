"""NMS entry points — mirror of the reference's utils/nms.py over csrc/postproc.hip."""
import numpy as np
import torch

from .. import _lib
from .._lib import check, lib

__all__ = ['gpu_nms', 'gpu_linear_soft_nms', 'gpu_exponential_soft_nms', 'numba_nms']


def _nms_one(boxes_t, scores_t, thr, inclusive):
    m = boxes_t.shape[0]
    if m == 0:
        return []
    dev = boxes_t.device
    cap = ((m + 3) // 4) * 4
    cand = torch.zeros(1, cap, 6, dtype=torch.float32, device=dev)
    cand[0, :m, :4] = boxes_t
    cand[0, :m, 4] = scores_t.reshape(-1)
    ncand = torch.tensor([m], dtype=torch.int32, device=dev)
    out = torch.empty(1, m, 6, dtype=torch.float32, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    keep = torch.empty(1, m, dtype=torch.int32, device=dev)
    L = lib()
    ws = torch.empty(L.yh_nms_ws_bytes(1, cap), dtype=torch.uint8, device=dev)
    check(L.yh_nms_batched(cand.data_ptr(), ncand.data_ptr(), 1, cap, float(thr), 0, int(inclusive), m, 0,
                           out.data_ptr(), nkeep.data_ptr(), keep.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "yh_nms_batched")
    n = int(nkeep.item())
    return keep[0, :n].tolist()


def numba_nms(boxes, scores, iou_threshold, device="cuda:0"):
    """Greedy hard-NMS with inclusive threshold, NumPy in, list of kept indices in pick order out
    (utils/nms.py:10-27).  Scores must be non-negative (the evaluator feeds sigmoid products)."""
    assert boxes.shape[0] == scores.shape[0]
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(device)
    s = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32)).to(device)
    return _nms_one(b, s, iou_threshold, True)


def _pairwise_kind(iou_type):
    """the reference's selector (utils/nms.py:43-52): name -> IoU function; 'iou' is the (N,M) matrix form"""
    from .bbox_tools import gpu_CIoU, gpu_DIoU, gpu_Giou, gpu_iou
    table = {'iou': gpu_iou, 'giou': gpu_Giou, 'diou': gpu_DIoU, 'ciou': gpu_CIoU}
    if iou_type not in table:
        raise ValueError(f'Uknown paramemter: <{iou_type}>')
    return table[iou_type]


def _one_vs_all(fn, iou_type, box1, boxes):
    """IoU of one box against all boxes as a flat (M,) tensor on the HIP kernels (yh_iou_matrix / yh_iou_pairwise)"""
    if iou_type == 'iou':
        return fn(box1, boxes).reshape(-1)
    return fn(box1.expand(boxes.shape[0], 4), boxes).reshape(-1)


def gpu_nms(boxes, scores, iou_type, iou_threshold):
    """Torch-tensor greedy NMS with exclusive threshold `iou > thr` (utils/nms.py:30-65), iou_type one of
    'iou' | 'giou' | 'diou' | 'ciou' (case-insensitive).  'iou' runs entirely in the HIP NMS kernel (one launch, pick order
    as utils/nms.py:54-64 intends — the reference itself raises IndexError there for M > 1 because the (1,M) mask of the
    matrix IoU indexes a 1-D score tensor).  The pairwise kinds follow the reference's loop: one arg-max + one pairwise
    IoU launch (yh_iou_pairwise) per kept box, like its one `.item()` per kept box."""
    assert isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor)
    assert boxes.shape[0] == scores.shape[0]
    if not boxes.is_cuda:
        raise _lib.YoloHipError("gpu_nms: tensors must live on an MI355X device")
    kind = iou_type.lower()
    fn = _pairwise_kind(kind)
    if kind == 'iou':
        return _nms_one(boxes.detach().float(), scores.detach().float(), iou_threshold, False)
    box_copy = boxes.detach().float().clone()
    score_copy = scores.detach().float().clone().reshape(-1)
    keep_index = []
    while score_copy.sum() > 0.:
        i = int(torch.argmax(score_copy).item())
        keep_index.append(i)
        score_copy[i] = 0.
        ious = _one_vs_all(fn, kind, box_copy[[i]], box_copy)
        score_copy[ious.gt(iou_threshold)] = 0.
    return keep_index


def _soft_nms(boxes, scores, iou_type, iou_threshold, thresh, decay):
    """common loop of the two soft-NMS variants (utils/nms.py:68-140): the current maximum is recorded in `processed`,
    then every box overlapping it by more than the threshold (the maximum itself included) has its score scaled by
    decay(iou).  Mirrors the reference statement for statement, quirks included: a box is never retired explicitly, it
    is picked again until its score has decayed to exactly zero, and `processed` keeps the LAST recorded value."""
    assert isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor)
    assert boxes.shape[0] == scores.shape[0]
    if not boxes.is_cuda:
        raise _lib.YoloHipError("soft-NMS: tensors must live on an MI355X device")
    fn = _pairwise_kind(iou_type)
    box_copy = boxes.detach().float().clone()
    score_copy = scores.detach().float().clone()
    flat = score_copy.reshape(-1)                       # view: (M,) and (M,1) inputs behave alike
    processed = torch.zeros_like(score_copy)
    pflat = processed.reshape(-1)
    while flat.sum() > 0.:
        i = int(torch.argmax(flat).item())
        pflat[i] = flat[i]
        ious = _one_vs_all(fn, iou_type, box_copy[[i]], box_copy)
        sel = ious.gt(iou_threshold)
        flat[sel] *= decay(ious[sel])
    keep = processed > thresh
    return keep.squeeze_()


def gpu_linear_soft_nms(boxes, scores, iou_type, iou_threshold=0.3, thresh=0.001):
    """utils/nms.py:68-102: score *= (1 - iou) for overlaps above the threshold; returns a bool mask (M,)"""
    return _soft_nms(boxes, scores, iou_type, iou_threshold, thresh, lambda u: 1. - u)


def gpu_exponential_soft_nms(boxes, scores, iou_type, iou_threshold, sigmma=0.5, thresh=0.001):
    """utils/nms.py:105-140: score *= exp(-iou^2 / sigma); returns a bool mask (M,)"""
    return _soft_nms(boxes, scores, iou_type, iou_threshold, thresh, lambda u: torch.exp(-(u ** 2) / sigmma))
