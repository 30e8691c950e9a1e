#!/usr/bin/env python3
"""val_yolox.py — the reference's YOLOX validation driver (val_yolox.py = val_yolov5.py with the YOLOX model table, the
anchor-free evaluator `YOLOXEvaluator(model, hyp, compute_metric=True)` and config/train_yolox.yaml) on the HIP path.

    python val_yolox.py --img 640 --batch 16 --val-batches 4 [--ckpt checkpoints/yolox_small_epoch_1.pth]
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import val_yolov5                                                                   # noqa: E402
from yoloseries_amd import models                                                   # noqa: E402
from yoloseries_amd.trainer import YOLOXEvaluator                                   # noqa: E402


class Training(val_yolov5.Training):
    """(the reference names the validation driver's class `Training` too)"""

    def __init__(self, hyp):
        super().__init__(None, hyp)

    def select_model(self):
        kind = self.hyp.get('model_type', 'small').lower()
        if kind != 'small':
            raise NotImplementedError(f"YOLOX model_type '{kind}': only YOLOXSmall is built on the HIP path")
        return models.YOLOXSmall(self.hyp.get('num_anchors', 1), 3, self.hyp['num_class'], self.hyp.get('weight_init_prior_prob', 0.01))

    def build_evaluator(self, model):
        return YOLOXEvaluator(model, self.hyp, compute_metric=True)


def main(argv=None):
    return val_yolov5.main(argv, training_cls=Training, default_cfg=os.path.join(ROOT, "config", "train_yolox.yaml"))


if __name__ == "__main__":
    main()
