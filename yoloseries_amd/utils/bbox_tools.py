"""Box utilities — mirror of the reference's utils/bbox_tools.py public surface.
IoU-family functions run on the HIP kernels of csrc/loss_v5.hip (yh_iou_matrix,
yh_iou_pairwise); the format conversions are the same one-line tensor expressions
as the reference (:87-134), evaluated on whatever device the tensor lives on."""
import numpy as np
import torch

from .. import _lib
from .._lib import check, lib

__all__ = ['xyxy2xywh', 'xyxy2xywhn', 'xywh2xyxy', 'numba_xywh2xyxy', 'gpu_iou', 'gpu_Giou', 'gpu_CIoU', 'gpu_DIoU',
           'numba_iou', 'numba_xyxy2xywh']


def _need_gpu(t, who):
    if not t.is_cuda:
        raise _lib.YoloHipError(f"{who}: tensor must live on an MI355X device (the product has no CPU path)")


def xyxy2xywh(bboxes):
    """[xmin, ymin, xmax, ymax] -> [center_x, center_y, w, h]  (utils/bbox_tools.py:87-100)"""
    new_bbox = torch.zeros_like(bboxes)
    new_bbox[..., [0, 1]] = (bboxes[..., [0, 1]] + bboxes[..., [2, 3]]) / 2
    new_bbox[..., [2, 3]] = bboxes[..., [2, 3]] - bboxes[..., [0, 1]]
    return new_bbox


def xyxy2xywhn(bboxes, img_shape):
    """utils/bbox_tools.py:103-119 (x,w / img_shape[0]; y,h / img_shape[1])"""
    assert bboxes.shape[-1] == 4, "the last dimension must equal 4"
    wh = bboxes[..., [2, 3]] - bboxes[..., [0, 1]]
    xy = (bboxes[..., [0, 1]] + bboxes[..., [2, 3]]) / 2
    out = torch.zeros_like(bboxes)
    # tensor/tensor division: torch's GPU "divide by a Python scalar" multiplies by the reciprocal,
    # which is 1 ulp off the reference's CPU result
    d0 = torch.full_like(xy[..., 0], float(img_shape[0]))
    d1 = torch.full_like(xy[..., 1], float(img_shape[1]))
    out[..., 0] = xy[..., 0] / d0
    out[..., 1] = xy[..., 1] / d1
    out[..., 2] = wh[..., 0] / d0
    out[..., 3] = wh[..., 1] / d1
    return out


def xywh2xyxy(bboxes):
    """utils/bbox_tools.py:122-134"""
    x, y, w, h = bboxes.chunk(4, -1)
    out = torch.zeros_like(bboxes)
    out[..., [0]] = x - w / 2
    out[..., [1]] = y - h / 2
    out[..., [2]] = x + w / 2
    out[..., [3]] = y + h / 2
    return out


def numba_xywh2xyxy(bboxes):
    """utils/bbox_tools.py:137-148 (NumPy in/out)"""
    out = np.zeros_like(bboxes)
    out[:, 0] = bboxes[:, 0] - bboxes[:, 2] / 2
    out[:, 1] = bboxes[:, 1] - bboxes[:, 3] / 2
    out[:, 2] = bboxes[:, 0] + bboxes[:, 2] / 2
    out[:, 3] = bboxes[:, 1] + bboxes[:, 3] / 2
    return out


def numba_xyxy2xywh(bboxes):
    """utils/bbox_tools.py:151-162 (NumPy in/out)"""
    out = np.zeros_like(bboxes)
    out[:, 0] = (bboxes[:, 0] + bboxes[:, 2]) / 2
    out[:, 1] = (bboxes[:, 1] + bboxes[:, 3]) / 2
    out[:, 2] = bboxes[:, 2] - bboxes[:, 0]
    out[:, 3] = bboxes[:, 3] - bboxes[:, 1]
    return out


def _iou_matrix(b1, b2, clamp):
    _need_gpu(b1, "iou")
    b1 = b1.detach().to(torch.float32).contiguous()
    b2 = b2.detach().to(torch.float32).contiguous()
    out = torch.empty(b1.shape[0], b2.shape[0], dtype=torch.float32, device=b1.device)
    check(lib().yh_iou_matrix(b1.data_ptr(), b1.shape[0], b2.data_ptr(), b2.shape[0], clamp, out.data_ptr(),
                              _lib.stream_ptr()), "yh_iou_matrix")
    return out


def gpu_iou(bbox1, bbox2):
    """(N,4),(M,4) xyxy -> (N,M), union clamped at 1e-9 (utils/bbox_tools.py:164-190)"""
    return _iou_matrix(bbox1, bbox2, 1e-9)


def numba_iou(bbox1, bbox2, device="cuda:0"):
    """NumPy in/out broadcast IoU without eps: 0/0 -> NaN (utils/bbox_tools.py:12-35)."""
    t1 = torch.from_numpy(np.ascontiguousarray(bbox1, dtype=np.float32)).to(device)
    t2 = torch.from_numpy(np.ascontiguousarray(bbox2, dtype=np.float32)).to(device)
    return _iou_matrix(t1, t2, 0.0).cpu().numpy()


class _PairwiseCIoU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b1, b2):
        n = b1.shape[0]
        a = b1.detach().to(torch.float32).contiguous()
        b = b2.detach().to(torch.float32).contiguous()
        out = torch.empty(n, dtype=torch.float32, device=a.device)
        grad = torch.empty(n, 4, dtype=torch.float32, device=a.device)
        check(lib().yh_iou_pairwise(2, a.data_ptr(), b.data_ptr(), n, out.data_ptr(), grad.data_ptr(), _lib.stream_ptr()),
              "yh_iou_pairwise")
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.reshape(-1, 1), None


def _pairwise(kind, b1, b2):
    _need_gpu(b1, "pairwise iou")
    assert b1.shape[-1] == b2.shape[-1] == 4 and b1.device == b2.device
    if b1.shape[0] != b2.shape[0]:
        b1 = b1.expand(b2.shape[0], 4) if b1.shape[0] == 1 else b1
    n = b1.shape[0]
    a = b1.detach().to(torch.float32).contiguous()
    b = b2.detach().to(torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    check(lib().yh_iou_pairwise(kind, a.data_ptr(), b.data_ptr(), n, out.data_ptr(), None, _lib.stream_ptr()), "yh_iou_pairwise")
    return out


def gpu_Giou(bbox1, bbox2):
    """pairwise GIoU (utils/bbox_tools.py:193-230); forward only"""
    return _pairwise(0, bbox1, bbox2)


def gpu_DIoU(bbox1, bbox2):
    """pairwise DIoU clamped to [-1,1] (utils/bbox_tools.py:233-283); forward only"""
    return _pairwise(1, bbox1, bbox2)


def gpu_CIoU(bbox1, bbox2):
    """pairwise CIoU, differentiable w.r.t. bbox1 with alpha held constant
    (utils/bbox_tools.py:286-339); returns (N,) squeezed like the reference."""
    _need_gpu(bbox1, "gpu_CIoU")
    if bbox1.shape[0] == 1 and bbox2.shape[0] > 1:          # the reference's expressions broadcast one box against many
        bbox1 = bbox1.expand(bbox2.shape[0], 4)
    return _PairwiseCIoU.apply(bbox1, bbox2).squeeze()
