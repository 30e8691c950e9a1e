"""NMS entry points — mirror of the reference's utils/nms.py over csrc/postproc.hip."""
import numpy as np
import torch

from .. import _lib
from .._lib import check, lib

__all__ = ['gpu_nms', 'numba_nms']


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


def gpu_nms(boxes, scores, iou_type, iou_threshold):
    """Torch-tensor NMS with exclusive threshold (utils/nms.py:30-65).  Only iou_type='iou' is wired to
    the HIP kernel; the reference implementation itself raises IndexError for M > 1 (:62-63)."""
    assert isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor)
    assert boxes.shape[0] == scores.shape[0]
    if iou_type.lower() != 'iou':
        raise NotImplementedError(f"gpu_nms: iou_type '{iou_type}' is not implemented on the HIP path")
    if not boxes.is_cuda:
        raise _lib.YoloHipError("gpu_nms: tensors must live on an MI355X device")
    return _nms_one(boxes.detach().float(), scores.detach().float(), iou_threshold, False)
