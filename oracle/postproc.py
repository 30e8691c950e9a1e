"""ORACLE (test infrastructure, never imported by the product path).

CPU/NumPy float32 restatement of the reference's inference post-processing:
  utils/nms.py:10-27                  numba_nms   (greedy hard-NMS, inclusive threshold)
  utils/nms.py:30-65                  gpu_nms     (exclusive threshold, clamped IoU)
  trainer/eval_yolov5.py:182-209      do_inference decode
  trainer/eval_yolov5.py:261-317      YOLOV5Evaluator.numba_nms post-process
  trainer/eval_yolox.py:123-150,201-259  YOLOX decode / post-process
Pinned by tests/golden/g4_decode.npz, g5_nms.npz (generated from the reference).
"""
import numpy as np

from .bbox import F32, evaluator_bbox_iou, gpu_iou, numba_iou, xywh2xyxy


def numba_nms(boxes, scores, iou_threshold):
    """utils/nms.py:10-27.  Returns the kept indices in pick order.
    Domain: scores >= 0 (the evaluator feeds products of sigmoids); then
    `scores.sum() > 0`  <=>  `scores.max() > 0`."""
    boxes = np.asarray(boxes, dtype=F32)
    s = np.asarray(scores, dtype=F32).copy()
    assert boxes.shape[0] == s.shape[0]
    thr = F32(iou_threshold)
    keep = []
    while s.size and s.max() > 0:
        i = int(np.argmax(s))          # first maximum on ties
        keep.append(i)
        s[i] = 0
        iou = numba_iou(boxes[i:i + 1], boxes)[0]
        s[iou >= thr] = 0              # NaN (0/0 area) never suppresses
    return keep


def gpu_nms(boxes, scores, iou_threshold):
    """utils/nms.py:30-65 with iou_type='iou' (gpu_iou, union clamp 1e-9), suppress iff iou > thr."""
    boxes = np.asarray(boxes, dtype=F32)
    s = np.asarray(scores, dtype=F32).copy()
    thr = F32(iou_threshold)
    keep = []
    while s.size and s.max() > 0:
        i = int(np.argmax(s))
        keep.append(i)
        s[i] = 0
        iou = gpu_iou(boxes[i:i + 1], boxes)[0]
        s[iou > thr] = 0
    return keep


def sigmoid32(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def decode_v5(stage_preds, anchors, strides):
    """trainer/eval_yolov5.py:182-209.  stage_preds: list of (B, A*(5+nc), h, w) float32
    (reference NCHW convention); anchors (S,A,2) pixels.  Returns (B, sum A*h*w, 5+nc)."""
    outs = []
    for i, p in enumerate(stage_preds):
        p = np.asarray(p, dtype=F32)
        B, _, h, w = p.shape
        A = anchors.shape[1]
        s = F32(strides[i])
        cur = p.reshape(B, A, -1, h, w).transpose(0, 1, 3, 4, 2)
        cur = sigmoid32(cur)
        gy, gx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        grid = np.stack((gx, gy), axis=2).astype(F32)[None, None]
        st_anchor = (np.asarray(anchors[i], dtype=F32) / s)[None, :, None, None, :]
        out = cur.copy()
        out[..., 0:2] = (cur[..., 0:2] * F32(2) - F32(0.5) + grid) * s
        out[..., 2:4] = (cur[..., 2:4] * F32(2)) ** 2 * st_anchor * s
        outs.append(out.reshape(B, -1, cur.shape[-1]))
    return np.concatenate(outs, axis=1)


def decode_yolox(stage_preds, img_h):
    """trainer/eval_yolox.py:123-150.  stage_preds: list of (B, A, 5+nc, h, w)."""
    outs = []
    for p in stage_preds:
        p = np.asarray(p, dtype=F32)
        B, A, E, h, w = p.shape
        s = F32(img_h / h)
        cur = p.transpose(0, 1, 3, 4, 2).copy()
        gy, gx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        grid = np.stack((gx, gy), axis=2).astype(F32)[None, None]
        out = cur.copy()
        out[..., 0:2] = (cur[..., 0:2] + grid) * s
        out[..., 2:4] = np.exp(cur[..., 2:4]) * s
        out[..., 4:] = sigmoid32(cur[..., 4:])
        outs.append(out.reshape(B, -1, E))
    return np.concatenate(outs, axis=1)


def candidates_v5(x_img, conf_thr, cls_thr, multi_label=False):
    """Rows of one image (N, 5+nc) -> candidate table (M,6) [xmin,ymin,xmax,ymax,conf,cls]
    trainer/eval_yolov5.py:266-286.  Single-label branch (:280-286): obj >= conf ; cls*obj ; argmax ; > cls_thr.
    Multi-label branch (:276-279, hyp['mutil_label']): every (row, class) with cls*obj >= cls_thr is a candidate of its own,
    in row-major order (np.nonzero)."""
    x = np.asarray(x_img, dtype=F32)
    x = x[x[:, 4] >= F32(conf_thr)].copy()
    if len(x) == 0:
        return np.zeros((0, 6), F32)
    x[:, 5:] *= x[:, 4:5]
    box = xywh2xyxy(x[:, :4])
    if multi_label:
        ri, ci = (x[:, 5:] >= F32(cls_thr)).nonzero()
        return np.concatenate((box[ri], x[ri, ci + 5][:, None], ci[:, None].astype(F32)), axis=1).astype(F32)
    conf = x[:, 5:].max(axis=1)
    cls = x[:, 5:].argmax(axis=1).astype(F32)
    out = np.concatenate((box, conf[:, None], cls[:, None]), axis=1).astype(F32)
    return out[conf > F32(cls_thr)]


def candidates_yolox(x_img, conf_thr, cls_thr, multi_label=False):
    """trainer/eval_yolox.py:206-231: pre-filter obj*max(cls) >= conf ; cls*=obj ; cls_conf >= cls_thr.
    Multi-label branch (:218-221, hyp['mutil_label']): every (row, class) with cls*obj >= cls_thr is a candidate of its own,
    in row-major order (np.nonzero)."""
    x = np.asarray(x_img, dtype=F32)
    m = (x[:, 4] * x[:, 5:].max(axis=1)) >= F32(conf_thr)
    x = x[m].copy()
    if len(x) == 0:
        return np.zeros((0, 6), F32)
    x[:, 5:] *= x[:, 4:5]
    box = xywh2xyxy(x[:, :4])
    if multi_label:
        ri, ci = (x[:, 5:] >= F32(cls_thr)).nonzero()
        return np.concatenate((box[ri], x[ri, ci + 5][:, None], ci[:, None].astype(F32)), axis=1).astype(F32)
    conf = x[:, 5:].max(axis=1)
    cls = x[:, 5:].argmax(axis=1).astype(F32)
    out = np.concatenate((box, conf[:, None], cls[:, None]), axis=1).astype(F32)
    return out[conf >= F32(cls_thr)]


def nms_image(cand, iou_thr, class_aware, max_keep, merge_filter, inclusive=True):
    """trainer/eval_yolov5.py:288-316 on one candidate table. Returns (rows (n,6) or None, keep idx)."""
    x = np.asarray(cand, dtype=F32)
    M = x.shape[0]
    if M == 0:
        return None, []
    off = x[:, 5] * F32(4096) if class_aware else x[:, 5] * F32(0)
    boxes = (x[:, :4] + off[:, None]).astype(F32)     # offset added in float32 BEFORE the IoU
    scores = x[:, 4]
    keep = numba_nms(boxes, scores, iou_thr) if inclusive else gpu_nms(boxes, scores, iou_thr)
    if len(keep) > max_keep:
        keep = keep[:max_keep]
    if merge_filter and 1 < M < 3000:
        iou = numba_iou(boxes[keep], boxes)
        mask = iou > F32(iou_thr)
        keep = list(np.asarray(keep)[mask.astype(F32).sum(axis=1) > 1])
    return x[keep], [int(k) for k in keep]


def do_nms_v5(decoded, conf_thr, cls_thr, iou_thr, class_aware=True, max_keep=300, merge_filter=True, multi_label=False, yolox=False):
    """YOLOV5Evaluator.do_nms (trainer/eval_yolov5.py:94-150) with iou_type 'iou': candidates as numba_nms, greedy NMS by
    utils.gpu_nms (exclusive threshold, clamped gpu_iou) on the class-offset boxes, cap, then the survivors that MORE than one
    candidate overlaps by > threshold under the evaluator's unclamped bbox_iou (:138-146; the merged boxes of :143 go to a
    temporary that is never returned).  The reference raises IndexError for any image with a candidate (utils/nms.py:62-63 —
    tests/golden/g14_round5.npz `donms_status`): this is the loop its gpu_nms spells, as the product implements it."""
    outs = []
    for i in range(decoded.shape[0]):
        x = (candidates_yolox if yolox else candidates_v5)(decoded[i], conf_thr, cls_thr, multi_label)
        M = x.shape[0]
        if M == 0:
            outs.append(None)
            continue
        off = x[:, 5] * F32(4096) if class_aware else x[:, 5] * F32(0)
        boxes = (x[:, :4] + off[:, None]).astype(F32)
        keep = gpu_nms(boxes, x[:, 4], iou_thr)[:max_keep]
        if merge_filter and 1 < M < 3000:
            iou = evaluator_bbox_iou(boxes[keep], boxes) if keep else np.zeros((0, M), F32)
            keep = list(np.asarray(keep, dtype=np.int64)[(iou > F32(iou_thr)).astype(F32).sum(axis=1) > 1])
        outs.append(x[keep])
    return outs


def postprocess_v5(decoded, conf_thr, cls_thr, iou_thr, class_aware=True, max_keep=300, merge_filter=True, multi_label=False):
    """YOLOV5Evaluator.numba_nms (trainer/eval_yolov5.py:261-317): list per image of (n,6) or None."""
    outs = []
    for i in range(decoded.shape[0]):
        cand = candidates_v5(decoded[i], conf_thr, cls_thr, multi_label)
        rows, _ = nms_image(cand, iou_thr, class_aware, max_keep, merge_filter)
        outs.append(rows if rows is not None else None)
    return outs


def postprocess_yolox(decoded, conf_thr, cls_thr, iou_thr, class_aware=True, max_keep=300, merge_filter=True, multi_label=False):
    """YOLOXEvaluator.numba_nms (trainer/eval_yolox.py:201-259): list per image of (n,6) or None."""
    outs = []
    for i in range(decoded.shape[0]):
        cand = candidates_yolox(decoded[i], conf_thr, cls_thr, multi_label)
        rows, _ = nms_image(cand, iou_thr, class_aware, max_keep, merge_filter)
        outs.append(rows if rows is not None else None)
    return outs
