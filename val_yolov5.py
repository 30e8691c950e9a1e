#!/usr/bin/env python3
"""Validation driver — mirror of the reference's val_yolov5.py:36-409 on the HIP path.

  * `Training(anchors, hyp).step()` runs the evaluator (decode + filter + class-aware NMS on the GPU,
    trainer/eval_yolov5.py) over the validation loader, maps predictions and ground truth back to the ORIGINAL image
    frame with the batch's `resize_info` (preds_postprocess / gt_bbox_postprocess, :140-258) and reports
    mAP@[.5:.95], mAP@.5, mean precision / recall with `mAP_v2` (:388-390).
  * weights come from a checkpoint written by train_yolov5.py (same keys as the reference's, :202-240) or stay random.
  * data: the synthetic dataset through DataLoader -> fixed_imgsize_collate_fn -> DataPrefetcher.
  Out of scope (SURVEY §8 "OUT OF SCOPE"): image dumps / plots, the auxiliary classifier, pickled box caches.

    python val_yolov5.py --img 640 --batch 16 --val-batches 4 [--ckpt checkpoints/yolov5_small_epoch_1.pth]
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from config.config import Config                                                         # noqa: E402
from train_yolov5 import PrefetchedDataset                                          # noqa: E402
from yoloseries_amd import models                                                   # noqa: E402
from yoloseries_amd.trainer import ExponentialMovingAverageModel, YOLOV5Evaluator   # noqa: E402
from yoloseries_amd.utils import mAP_v2                                             # noqa: E402
from yoloseries_amd.utils.dist import get_local_rank, get_rank                      # noqa: E402
from yoloseries_amd.utils.synth import COCO_ANCHORS                                 # noqa: E402


class Training:
    """(the reference names the validation driver's class `Training` too, val_yolov5.py:36)"""

    def __init__(self, anchors, hyp):
        self.anchors, self.hyp = anchors, hyp
        if not torch.cuda.is_available():
            raise RuntimeError("val_yolov5.py needs an MI355X device (the product path has no CPU fallback)")
        self.local_rank, self.rank = get_local_rank(), get_rank()
        torch.cuda.set_device(self.local_rank)
        self.device = f"cuda:{self.local_rank}"
        self.hyp['device'] = self.device
        self.cwd = Path('./').absolute()
        self.before_validation()

    def select_model(self):
        table = {'small': models.YOLOV5Small, 'middle': models.YOLOV5Middle, 'large': models.YOLOV5Large, 'xlarge': models.YOLOV5XLarge}
        return table[self.hyp.get('model_type', 'small').lower()](3, self.hyp['num_class'], 3)

    def before_validation(self):
        hyp = self.hyp
        hyp['input_img_size'] = [int(np.ceil(s / 32) * 32) for s in hyp['input_img_size']]
        img = hyp['input_img_size'][0]
        self.val_dataloader = PrefetchedDataset(hyp['val_batches'], hyp['batch_size'], img, hyp['num_class'], 101 + self.rank,
                                                hyp.get('num_workers', 0))
        self.model = self.select_model().to(self.device)
        self.ema_model = ExponentialMovingAverageModel(self.model) if hyp.get('do_ema', True) else None
        self.loaded_ema = False
        if hyp.get('pretrained_model_path'):
            self.load_model(hyp['pretrained_model_path'])

    def build_evaluator(self, model):
        return YOLOV5Evaluator(model, self.anchors.to(self.device), self.hyp, compute_metric=True)      # val_yolov5.py:301

    def load_model(self, path):
        """checkpoint keys of train_yolov5.py:603-629; the EMA weights are preferred when present (val_yolov5.py:297-303)"""
        state = torch.load(path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(state["model_state_dict"])
        if self.ema_model is not None and state.get("ema") is not None:
            self.ema_model.ema.load_state_dict(state["ema"])
            self.ema_model.update_num = state.get("ema_update_num", 0)
            self.loaded_ema = True

    # ---- frames: letterboxed network input <-> original image (val_yolov5.py:140-258)
    @staticmethod
    def preds_postprocess(outputs, info):
        """outputs[i]: None or (K,6) tensor [xmin,ymin,xmax,ymax,conf,cls] in the letterboxed frame -> NumPy in the original frame"""
        processed = []
        for i, pred in enumerate(outputs):
            if pred is None:
                processed.append(None)
                continue
            scale, pad_top, pad_left = info[i]['scale'], info[i]['pad_top'], info[i]['pad_left']
            org_h, org_w = info[i]['org_shape']
            pred = pred.clone()
            pred[:, [0, 2]] -= pad_left
            pred[:, [1, 3]] -= pad_top
            pred[:, [0, 1, 2, 3]] /= scale
            pred[:, [0, 2]] = pred[:, [0, 2]].clamp(1, org_w - 1)
            pred[:, [1, 3]] = pred[:, [1, 3]].clamp(1, org_h - 1)
            processed.append(pred.cpu().numpy())
        return processed

    @staticmethod
    def gt_bbox_postprocess(anns, infoes):
        ppb, ppc = [], []
        for i in range(anns.shape[0]):
            scale, pad_top, pad_left = infoes[i]['scale'], infoes[i]['pad_top'], infoes[i]['pad_left']
            ann_valid = anns[i][anns[i][:, 4] >= 0].clone()
            ann_valid[:, [0, 2]] -= pad_left
            ann_valid[:, [1, 3]] -= pad_top
            ann_valid[:, :4] /= scale
            ppb.append(ann_valid[:, :4].cpu().numpy())
            ppc.append(ann_valid[:, 4].cpu().numpy().astype('uint16'))
        return ppb, ppc

    def step(self):
        eval_model = self.ema_model.ema if (self.ema_model is not None and self.loaded_ema) else self.model
        eval_model.eval()
        validater = self.build_evaluator(eval_model)
        all_preds, all_gts = [], []
        t0 = time.time()
        n_img = 0
        for x in self.val_dataloader:
            gt_bbox, gt_cls = self.gt_bbox_postprocess(x['ann'], x['resize_info'])
            outputs = validater(x['img'])
            preds = self.preds_postprocess(outputs, x['resize_info'])
            for j in range(len(preds)):
                if preds[j] is not None and (preds[j][:, 5] >= 0).any():
                    all_preds.append(preds[j][preds[j][:, 5] >= 0])
                else:
                    all_preds.append(np.zeros((0, 6)))
                all_gts.append(np.concatenate((gt_bbox[j], gt_cls[j][:, None]), axis=1))
            n_img += len(preds)
        torch.cuda.synchronize()
        dt = time.time() - t0
        if any(len(p) for p in all_preds):
            m, m50, mp, mr = mAP_v2(all_gts, all_preds, self.cwd / "result" / "curve").get_mean_metrics()
        else:
            m = m50 = mp = mr = 0.0
        self.metrics = dict(map=m, map50=m50, precision=mp, recall=mr, n_pred=int(sum(len(p) for p in all_preds)), images=n_img,
                            img_per_s=n_img / max(dt, 1e-9))
        if self.rank == 0:
            print(f"map={m}, map50={m50}, mp={mp}, mr={mr}  ({n_img} images, {self.metrics['n_pred']} boxes, {self.metrics['img_per_s']:.1f} img/s)")
        return self.metrics


def main(argv=None, training_cls=None, default_cfg=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=default_cfg or os.path.join(ROOT, "config", "train_yolov5.yaml"))
    ap.add_argument("--img", type=int)
    ap.add_argument("--batch", type=int)
    ap.add_argument("--val-batches", type=int)
    ap.add_argument("--model-type")
    ap.add_argument("--ckpt")
    args = ap.parse_args(argv)
    hyp = Config().get_config(args.cfg)
    if args.img: hyp['input_img_size'] = [args.img, args.img]            # noqa: E701
    if args.batch: hyp['batch_size'] = args.batch                        # noqa: E701
    if args.val_batches: hyp['val_batches'] = args.val_batches           # noqa: E701
    if args.model_type: hyp['model_type'] = args.model_type              # noqa: E701
    if args.ckpt: hyp['pretrained_model_path'] = args.ckpt               # noqa: E701
    if training_cls is not None:
        v = training_cls(hyp)
    else:
        anchors = torch.from_numpy(COCO_ANCHORS.copy())
        v = Training(anchors, hyp)
    v.step()
    return v


if __name__ == "__main__":
    main()
