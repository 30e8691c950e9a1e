"""ORACLE (test infrastructure, never imported by the product path).

CPU/NumPy float32 restatement of the reference's box utilities, /root/reference
utils/bbox_tools.py.  Pinned against golden vectors generated from the reference
itself (tools/gen_golden.py -> tests/golden/g1_boxes.npz).

Every function keeps the reference's operation order in float32 so that results are
bit-identical to the reference running on NumPy float32 / torch CPU float32.
"""
import numpy as np

F32 = np.float32


def numba_iou(bbox1, bbox2):
    """utils/bbox_tools.py:12-35 — broadcast IoU (M,N); no eps in the denominator
    (0/0 -> NaN), float32 throughout (the numba-less NumPy path, SURVEY §8a note 1)."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    ymax = np.minimum(b1[:, 3][:, None], b2[:, 3])
    xmax = np.minimum(b1[:, 2][:, None], b2[:, 2])
    ymin = np.maximum(b1[:, 1][:, None], b2[:, 1])
    xmin = np.maximum(b1[:, 0][:, None], b2[:, 0])
    w = np.maximum(F32(0.), xmax - xmin)
    h = np.maximum(F32(0.), ymax - ymin)
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (a1[:, None] + a2 - inter)).astype(F32)


def evaluator_bbox_iou(bbox1, bbox2):
    """trainer/eval_yolov5.py:237-258 == trainer/eval_yolox.py:177-199 — the evaluators' static `bbox_iou`: (N,M) IoU with NO clamp
    on the intersection sides (boxes apart on both axes get the positive product of two negative sides) and none on the union
    (0/0 -> NaN).  Pinned by tests/golden/g14_round5.npz (iou_v5, iou_yolox)."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    xmin = np.maximum(b1[:, 0][:, None], b2[:, 0])
    xmax = np.minimum(b1[:, 2][:, None], b2[:, 2])
    ymin = np.maximum(b1[:, 1][:, None], b2[:, 1])
    ymax = np.minimum(b1[:, 3][:, None], b2[:, 3])
    h = ymax - ymin
    w = xmax - xmin
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (a1[:, None] + a2 - inter)).astype(F32)


def gpu_iou(bbox1, bbox2):
    """utils/bbox_tools.py:164-190 — (N,M) IoU, union clamped at 1e-9."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    ymax = np.minimum(b1[:, None, 3], b2[None, :, 3])
    xmax = np.minimum(b1[:, None, 2], b2[None, :, 2])
    ymin = np.maximum(b1[:, None, 1], b2[None, :, 1])
    xmin = np.maximum(b1[:, None, 0], b2[None, :, 0])
    w = np.maximum(xmax - xmin, F32(0.))
    h = np.maximum(ymax - ymin, F32(0.))
    inter = w * h
    union = np.maximum(a1[:, None] + a2[None, :] - inter, F32(1e-9))
    return (inter / union).astype(F32)


def _pair_common(b1, b2, eps):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    ymax = np.minimum(b1[:, 3], b2[:, 3])
    xmax = np.minimum(b1[:, 2], b2[:, 2])
    ymin = np.maximum(b1[:, 1], b2[:, 1])
    xmin = np.maximum(b1[:, 0], b2[:, 0])
    w = np.maximum(xmax - xmin, F32(0.))
    h = np.maximum(ymax - ymin, F32(0.))
    inter = w * h
    union = a1 + a2 - inter
    iou = inter / np.maximum(union, F32(eps))
    return inter, union, iou


def gpu_giou(bbox1, bbox2):
    """utils/bbox_tools.py:193-230 — pairwise GIoU, eps 1e-6."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    _, union, iou = _pair_common(b1, b2, 1e-6)
    cx0 = np.minimum(b1[:, 0], b2[:, 0]); cx1 = np.maximum(b1[:, 2], b2[:, 2])
    cy0 = np.minimum(b1[:, 1], b2[:, 1]); cy1 = np.maximum(b1[:, 3], b2[:, 3])
    c_area = (cx1 - cx0) * (cy1 - cy0)
    return (iou - np.abs(c_area - union) / np.abs(np.maximum(c_area, F32(1e-6)))).astype(F32)


def gpu_diou(bbox1, bbox2):
    """utils/bbox_tools.py:233-283 — pairwise DIoU, eps 1e-6, clamped to [-1,1]."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    _, _, iou = _pair_common(b1, b2, 1e-6)
    cx0 = np.minimum(b1[:, 0], b2[:, 0]); cx1 = np.maximum(b1[:, 2], b2[:, 2])
    cy0 = np.minimum(b1[:, 1], b2[:, 1]); cy1 = np.maximum(b1[:, 3], b2[:, 3])
    c_hs = cy1 - cy0
    c_ws = cx1 - cx0
    c_diag = c_ws ** 2 + c_hs ** 2
    b1x = (b1[:, 2] + b1[:, 0]) / F32(2); b1y = (b1[:, 3] + b1[:, 1]) / F32(2)
    b2x = (b2[:, 2] + b2[:, 0]) / F32(2); b2y = (b2[:, 3] + b2[:, 1]) / F32(2)
    d = (b1x - b2x) ** 2 + (b1y - b2y) ** 2
    out = iou - d / np.maximum(c_diag, F32(1e-6))
    return np.clip(out, F32(-1), F32(1)).astype(F32)


def gpu_ciou(bbox1, bbox2):
    """utils/bbox_tools.py:286-339 — pairwise CIoU, eps 1e-9 clamps; alpha is a
    constant in the backward (torch.no_grad at :335-336).  float32, operation order
    of the reference; arctan in float32."""
    b1 = np.asarray(bbox1, dtype=F32)
    b2 = np.asarray(bbox2, dtype=F32)
    eps = F32(1e-9)
    w1 = b1[:, 2] - b1[:, 0]; h1 = b1[:, 3] - b1[:, 1]
    w2 = b2[:, 2] - b2[:, 0]; h2 = b2[:, 3] - b2[:, 1]
    ymax = np.minimum(b1[:, 3], b2[:, 3]); xmax = np.minimum(b1[:, 2], b2[:, 2])
    ymin = np.maximum(b1[:, 1], b2[:, 1]); xmin = np.maximum(b1[:, 0], b2[:, 0])
    iw = np.maximum(xmax - xmin, F32(0.)); ih = np.maximum(ymax - ymin, F32(0.))
    inter = iw * ih
    union = np.maximum(w1 * h1 + w2 * h2 - inter, eps)
    iou = inter / union
    c_hs = np.maximum(b1[:, 3], b2[:, 3]) - np.minimum(b1[:, 1], b2[:, 1])
    c_ws = np.maximum(b1[:, 2], b2[:, 2]) - np.minimum(b1[:, 0], b2[:, 0])
    c_diag = c_ws ** 2 + c_hs ** 2
    b1x = (b1[:, 0] + b1[:, 2]) / F32(2); b1y = (b1[:, 1] + b1[:, 3]) / F32(2)
    b2x = (b2[:, 0] + b2[:, 2]) / F32(2); b2y = (b2[:, 1] + b2[:, 3]) / F32(2)
    ctr_ws = b1x - b2x; ctr_hs = b1y - b2y
    ctr = ctr_hs ** 2 + ctr_ws ** 2
    k = F32(4 / (np.pi ** 2))
    v = k * (np.arctan(w1 / np.maximum(h1, eps)) - np.arctan(w2 / np.maximum(h2, eps))) ** 2
    alpha = v / np.maximum(F32(1) - iou + v, eps)
    c_diag = np.maximum(c_diag, eps)
    return (iou - (ctr / c_diag + v * alpha)).astype(F32)


def xyxy2xywh(b):
    """utils/bbox_tools.py:87-100"""
    b = np.asarray(b, dtype=F32)
    out = np.zeros_like(b)
    out[..., 2:4] = b[..., 2:4] - b[..., 0:2]
    out[..., 0:2] = (b[..., 0:2] + b[..., 2:4]) / F32(2)
    return out


def xyxy2xywhn(b, img_shape):
    """utils/bbox_tools.py:103-119 — x and w divided by img_shape[0], y and h by img_shape[1]."""
    b = np.asarray(b, dtype=F32)
    wh = b[..., 2:4] - b[..., 0:2]
    xy = (b[..., 0:2] + b[..., 2:4]) / F32(2)
    out = np.zeros_like(b)
    out[..., 0] = xy[..., 0] / F32(img_shape[0])
    out[..., 1] = xy[..., 1] / F32(img_shape[1])
    out[..., 2] = wh[..., 0] / F32(img_shape[0])
    out[..., 3] = wh[..., 1] / F32(img_shape[1])
    return out


def xywh2xyxy(b):
    """utils/bbox_tools.py:122-134 and numba_xywh2xyxy :137-148"""
    b = np.asarray(b, dtype=F32)
    out = np.zeros_like(b)
    out[..., 0] = b[..., 0] - b[..., 2] / F32(2)
    out[..., 1] = b[..., 1] - b[..., 3] / F32(2)
    out[..., 2] = b[..., 0] + b[..., 2] / F32(2)
    out[..., 3] = b[..., 1] + b[..., 3] / F32(2)
    return out
