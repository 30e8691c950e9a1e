#!/usr/bin/env python3
"""train_yolox.py — the reference's YOLOX training driver on the MI355X hot path.

The reference's train_yolox.py is its train_yolov5.py with another model table, `YOLOXLoss(hyp)` / `YOLOXEvaluator(model, hyp)`
(no anchors), the L1 term in the log line, `yolox_*` checkpoint names, config/train_yolox.yaml and the gradient clipping
commented out (train_yolox.py:31-32, 51-56, 112-123, 192, 209, 328-329, 384-397, 594, 683, 802-808).  The same holds here:
`Training` derives from this repository's train_yolov5.Training and overrides exactly those points (the clipping stays on:
see CLIP_GRAD_NORM).  YOLOXLoss converts the
target boxes to xywh in place like the reference (loss/yolox_loss.py:70-75): every batch of the loaders is a fresh tensor.

    python train_yolox.py [--cfg config/train_yolox.yaml] [--epochs N] [--img 640] [--batch 64] [--data tensor|dataset|shapes]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import train_yolov5                                                                 # noqa: E402
from yoloseries_amd import models                                                   # noqa: E402
from yoloseries_amd.loss import YOLOXLoss                                           # noqa: E402
from yoloseries_amd.trainer import YOLOXEvaluator                                   # noqa: E402


class Training(train_yolov5.Training):
    CKPT_PREFIX = "yolox"             # train_yolox.py:594
    # train_yolox.py:328-329 comments clip_grad_norm_ out and leans on the AMP GradScaler, which skips a step whose gradients
    # overflow.  The bf16 path has no scaler (nothing overflows in bf16's range, so nothing would be skipped): the first steps
    # of a fresh YOLOX (objectness loss ~2e3 against bias learning rates warmed DOWN from 0.1) diverge without a guard.  The
    # v5 driver's clipping is kept as that guard — a documented deviation.  hyp['clip_grad_norm'] (config/train_yolox.yaml)
    # sets the bound; `null` there restores the reference's behaviour (no clipping).
    CLIP_GRAD_NORM = 10.0

    def __init__(self, hyp):
        super().__init__(None, hyp)
        if 'clip_grad_norm' in hyp:
            self.CLIP_GRAD_NORM = None if hyp['clip_grad_norm'] is None else float(hyp['clip_grad_norm'])
        if self.rank == 0:
            print("YOLOX driver: " + ("no gradient clipping (reference behaviour, train_yolox.py:328-329)" if self.CLIP_GRAD_NORM is None
                                      else f"clip_grad_norm_({self.CLIP_GRAD_NORM}) — bf16 path has no GradScaler to skip diverging steps"))

    def select_model(self):
        """train_yolox.py:112-123; YOLOXSmall is the family member inside the hot-path scope"""
        kind = self.hyp.get('model_type', 'small').lower()
        if kind != 'small':
            raise NotImplementedError(f"YOLOX model_type '{kind}': only YOLOXSmall is built on the HIP path")
        return models.YOLOXSmall(self.hyp.get('num_anchors', 1), 3, self.hyp['num_class'], self.hyp.get('weight_init_prior_prob', 0.01))

    def build_loss(self):
        return YOLOXLoss(self.hyp)                                                   # train_yolox.py:209

    def build_evaluator(self, model):
        return YOLOXEvaluator(model, self.hyp, compute_metric=True)                  # train_yolox.py:683

    def log_line(self, h):
        return (f"tot {h['tot_loss']:.3f} iou {h['iou_loss']:.3f} l1 {h['l1_loss']:.3f} cof {h['cof_loss']:.3f} cls {h['cls_loss']:.3f} "
                f"tars {h['tar_nums']}")


def main(argv=None):
    t = train_yolov5.main(argv, training_cls=Training, default_cfg=os.path.join(ROOT, "config", "train_yolox.yaml"))
    return t


if __name__ == "__main__":
    main()
