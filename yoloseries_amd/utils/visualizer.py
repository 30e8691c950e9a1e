"""cv2_save_img / cv2_save_img_plot_pred_gt — the two drawing helpers the drivers import (utils/visualizer.py:101-214):
[xmin, ymin, xmax, ymax] boxes with captions on an RGB uint8 image, written to `save_path`.  Drawing goes through a tiny
canvas with two back ends: OpenCV when it is installed (the reference's only path), Pillow otherwise (this image has no OpenCV)."""
from pathlib import Path

import numpy as np

__all__ = ["cv2_save_img", "cv2_save_img_plot_pred_gt"]

PRED_EDGE, PRED_TAG = (0, 238, 238), (200, 0, 0)
GT_EDGE, GT_TAG = (0, 255, 255), (0, 200, 0)
WHITE = (255, 255, 255)


def _cv2():
    try:
        import cv2
    except ImportError:
        return None
    return cv2 if hasattr(cv2, "imwrite") and hasattr(cv2, "rectangle") else None


class _Canvas:
    """an RGB uint8 image that boxes and captions are drawn on"""

    def __init__(self, img):
        self.cv2 = _cv2()
        arr = np.ascontiguousarray(img).astype(np.uint8)
        if self.cv2 is None:
            from PIL import Image, ImageDraw
            self.im = Image.fromarray(arr)
            self.dr = ImageDraw.Draw(self.im)
        else:
            self.arr = arr.copy()

    def box(self, p0, p1, color, fill=False):
        if self.cv2 is None:
            (x0, y0), (x1, y1) = p0, p1
            xy = [min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)]
            self.dr.rectangle(xy, fill=color if fill else None, outline=color, width=1)
        else:
            self.arr = self.cv2.rectangle(self.arr, pt1=p0, pt2=p1, color=list(color), thickness=-1 if fill else 1)

    def text(self, org, s):
        """`org` = left end of the baseline, as cv2.putText takes it"""
        if self.cv2 is None:
            self.dr.text((org[0] + 1, org[1] - 9), s, fill=WHITE)
        else:
            self.arr = self.cv2.putText(self.arr, text=s, org=org, fontFace=self.cv2.FONT_HERSHEY_SIMPLEX, fontScale=0.35,
                                        color=list(WHITE), thickness=1, lineType=self.cv2.LINE_AA)

    def array(self):
        return np.asarray(self.im).copy() if self.cv2 is None else self.arr.copy()


def _labelled_boxes(cv, boxes, captions, edge, tag, tag_dy):
    """each box outlined in `edge` with a filled caption strip 12 px high above (tag_dy < 0) or inside (tag_dy > 0) its top edge"""
    for b, cap in zip(boxes, captions):
        x0, y0, x1, y1 = round(b[0]), round(b[1]), round(b[2]), round(b[3])
        cv.box((x0, y0), (x1, y1), edge)
        cv.box((x0, y0), (x0 + int(b[2] - b[0]), y0 + tag_dy), tag, fill=True)
        cv.text((x0, y0 + (9 if tag_dy > 0 else -9)), cap)


def _write(save_path, rgb):
    Path(save_path).parent.mkdir(parents=True, exist_ok=True)
    cv2 = _cv2()
    if cv2 is not None:
        cv2.imwrite(str(save_path), np.ascontiguousarray(rgb[:, :, ::-1]))
    else:
        from PIL import Image
        Image.fromarray(rgb).save(str(save_path))


def cv2_save_img(img, bboxes, labels, scores, save_path):
    """predictions only (utils/visualizer.py:167-214); captions 'label:score' (label alone without scores)"""
    assert isinstance(img, np.ndarray)
    assert len(bboxes) == len(labels)
    has_scores = scores is not None and len(scores) == len(labels)
    caps = [f"{labels[i]}:{scores[i]:.1f}" if has_scores else f"{labels[i]}" for i in range(len(labels))]
    cv = _Canvas(img)
    _labelled_boxes(cv, bboxes, caps, PRED_EDGE, PRED_TAG, 12)
    _write(save_path, cv.array())


def cv2_save_img_plot_pred_gt(img, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, save_path):
    """predictions and ground truth on one picture (utils/visualizer.py:101-164, called by val_yolov5.py): the prediction layer
    (caption strip inside the box) and the layer that also carries the ground truth (strip above the box) are blended 0.65 : 0.35"""
    assert isinstance(img, np.ndarray)
    assert len(pred_bboxes) == len(pred_labels)
    assert len(gt_bboxes) == len(gt_labels)
    cv = _Canvas(img)
    _labelled_boxes(cv, pred_bboxes, [f"{pred_labels[i]}:{pred_scores[i]:.1f}" for i in range(len(pred_labels))], PRED_EDGE, PRED_TAG, 12)
    pred_layer = cv.array()
    _labelled_boxes(cv, gt_bboxes, [f"{g}" for g in gt_labels], GT_EDGE, GT_TAG, -12)
    both_layer = cv.array() if len(gt_bboxes) else np.ascontiguousarray(img).astype(np.uint8)
    _write(save_path, (pred_layer * 0.65 + both_layer * 0.35).astype(np.uint8))
