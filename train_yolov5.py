#!/usr/bin/env python3
"""train_yolov5.py — the reference's training driver (train_yolov5.py:49-870) on the MI355X hot path.

Same `Training` life-cycle and method names (before_train, select_model, _init_optimizer, _init_scheduler, warmup,
step, after_epoch, save_model, load_model) and the same flat `hyp` dict (config/config.py), but:
  * device is the rank's MI355X (no nvidia-smi / occupy_mem), one process per GPU under torch.distributed.run;
  * data come from a synthetic generator in the collate format of dataset/data_collater.py:20-64
    ({'img': (B,3,H,W) float 0-1, 'ann': (B,maxbox,6) padded with -1}) — no dataset code ships here;
  * model / loss / evaluator / EMA / metric are the yoloseries_amd mirrors; gradients are exchanged by
    DataParallelGrads (DDP semantics) and the optimizer is the flat-arena SGD with the reference's 3 groups.

    python train_yolov5.py [--cfg config/train_yolov5.yaml] [--epochs N] [--img 640] [--batch 64]
"""
import argparse
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from config.config import Config                                                    # noqa: E402
from yoloseries_amd import models                                                   # noqa: E402
from yoloseries_amd.loss import YOLOV5Loss                                          # noqa: E402
from yoloseries_amd.trainer import ExponentialMovingAverageModel, YOLOV5Evaluator   # noqa: E402
from yoloseries_amd.utils import FlatSGD, mAP_v2                                    # noqa: E402
from yoloseries_amd.utils.dist import (DataParallelGrads, all_reduce_norm, get_local_rank, get_rank, get_world_size,
                                       synchronize)                                # noqa: E402
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_shapes_batch, synth_targets      # noqa: E402


class SyntheticLoader:
    """len()-able iterable of collated batches (dataset/data_collater.py:20-64 format), one RNG stream per rank"""

    def __init__(self, steps, batch, img, num_class, device, seed, shapes=False):
        self.steps, self.batch, self.img, self.nc, self.device, self.seed, self.shapes = steps, batch, img, num_class, device, seed, shapes
        self.epoch = 0

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        self.epoch += 1
        for i in range(self.steps):
            if self.shapes:      # learnable task (coloured rectangles): fresh images every epoch, as an augmenting loader gives
                im, an = synth_shapes_batch(self.batch, self.img, self.nc, 4, seed=(self.seed * 1000 + i) * 997 + self.epoch)
                img, ann = torch.from_numpy(im), torch.from_numpy(an)
            else:
                img = torch.rand(self.batch, 3, self.img, self.img, generator=g)
                ann = torch.from_numpy(synth_targets(self.batch, self.img, self.nc, 20, seed=self.seed * 1000 + i))
            yield {'img': img.to(self.device, non_blocking=True), 'ann': ann.to(self.device, non_blocking=True)}


class PrefetchedDataset:
    """the reference's data path on a synthetic dataset: torch DataLoader -> fixed_imgsize_collate_fn
    (dataset/data_collater.py:20-64) -> DataPrefetcher (dataset/data_prefetcher.py:6-56, train_yolov5.py:458-497)"""

    def __init__(self, steps, batch, img, num_class, seed, workers=0):
        from functools import partial
        from torch.utils.data import DataLoader
        from yoloseries_amd.dataset import DataPrefetcher, SyntheticDetectionDataset, fixed_imgsize_collate_fn
        self._prefetcher = DataPrefetcher
        ds = SyntheticDetectionDataset(steps * batch, img_hw=(int(img * 0.75) // 8 * 8, img), num_class=num_class, seed=seed)
        self.loader = DataLoader(ds, batch_size=batch, shuffle=False, num_workers=workers, drop_last=True, pin_memory=True,
                                 collate_fn=partial(fixed_imgsize_collate_fn, dst_size=[img, img]))

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        pf = self._prefetcher(self.loader)
        for _ in range(len(self.loader)):
            x = pf.next()
            if x['img'] is None:
                return
            yield x


class Training:
    CKPT_PREFIX = "yolov5"            # checkpoint file names (train_yolov5.py:608; train_yolox.py:594 uses "yolox")
    CLIP_GRAD_NORM = 10.0             # train_yolov5.py:343-344 (train_yolox.py comments the clipping out: None there)

    def __init__(self, anchors, hyp):
        self.anchors, self.hyp = anchors, hyp
        self.local_rank, self.rank = get_local_rank(), get_rank()
        if not torch.cuda.is_available():
            raise RuntimeError("train_yolov5.py needs an MI355X device (the product path has no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.device = f"cuda:{self.local_rank}"
        self.hyp['device'] = self.device
        self.is_distributed = get_world_size() > 1
        self.cwd = Path('./').absolute()
        self.history = []
        self.before_train()

    # ------------------------------------------------------------------ set-up (train_yolov5.py:166-256)
    def select_model(self):
        table = {'small': models.YOLOV5Small, 'middle': models.YOLOV5Middle, 'large': models.YOLOV5Large, 'xlarge': models.YOLOV5XLarge}
        return table[self.hyp.get('model_type', 'small').lower()](3, self.hyp['num_class'], 3)

    def before_train(self):
        hyp = self.hyp
        hyp['input_img_size'] = [int(math.ceil(s / 32) * 32) for s in hyp['input_img_size']]          # padding(..., 32) :170
        hyp['batch_size'] = max(1, hyp['batch_size'] // get_world_size())                            # :180-181
        hyp['lr'] = hyp['basic_lr_per_img'] * hyp['batch_size']                                      # :184
        torch.manual_seed(hyp['random_seed'])
        img = hyp['input_img_size'][0]
        if hyp.get('data_source', 'tensor') == 'dataset':
            self.train_dataloader = PrefetchedDataset(hyp['steps_per_epoch'], hyp['batch_size'], img, hyp['num_class'], 1 + self.rank,
                                                      hyp.get('num_workers', 0))
            self.val_dataloader = PrefetchedDataset(hyp['val_batches'], hyp['batch_size'], img, hyp['num_class'], 101 + self.rank)
        else:
            shapes = hyp.get('data_source', 'tensor') == 'shapes'
            self.train_dataloader = SyntheticLoader(hyp['steps_per_epoch'], hyp['batch_size'], img, hyp['num_class'], self.device, 1 + self.rank, shapes)
            self.val_dataloader = SyntheticLoader(hyp['val_batches'], hyp['batch_size'], img, hyp['num_class'], self.device, 101 + self.rank, shapes)
        hyp['warmup_steps'] = max(hyp.get('warmup_epoch', 3) * len(self.train_dataloader), 1)       # :192
        self.model = self.select_model().to(self.device)
        self.dp = DataParallelGrads(self.model) if self.is_distributed else None
        self.loss_fcn = self.build_loss()
        self.ema_model = ExponentialMovingAverageModel(self.model) if hyp['do_ema'] else None
        self.optimizer = self._init_optimizer()
        self.lr_scheduler_fn = self._init_scheduler()
        self.accumulate = max(1, round(hyp['accumulate_loss_step'] / hyp['batch_size'] / get_world_size()))
        self.validate = self.build_evaluator(self.model)
        self.start_epoch = hyp.get('start_epoch', 0)
        if hyp.get('pretrained_model_path'):
            self.load_model(hyp['pretrained_model_path'])

    def build_loss(self):
        return YOLOV5Loss(self.anchors.to(self.device), self.hyp)                     # train_yolov5.py:223

    def build_evaluator(self, model):
        return YOLOV5Evaluator(model, self.anchors.to(self.device), self.hyp, compute_metric=True)   # :697

    def log_line(self, h):
        return f"tot {h['tot_loss']:.3f} box {h['iou_loss']:.3f} cof {h['cof_loss']:.3f} cls {h['cls_loss']:.3f} tars {h['tar_nums']}"

    def _init_optimizer(self):
        """3 parameter groups: BN weights | conv weights (+weight decay) | biases (train_yolov5.py:258-280)"""
        assert self.hyp['optimizer'].lower() == 'sgd', "the flat-arena optimizer implements SGD-nesterov (the shipped configuration)"
        return FlatSGD(self.model, lr=self.hyp['lr'], momentum=self.hyp['momentum'], weight_decay=self.hyp['weight_decay'], nesterov=True)

    def _init_scheduler(self):
        """per-epoch LambdaLR factors (train_yolov5.py:152-164)"""
        hyp = self.hyp
        ds, T = hyp['lr_max_ds_scale'], hyp['total_epoch']
        kind = hyp['scheduler_type'].lower()
        if kind == 'onecycle':
            return lambda e: ((1.0 - math.cos(e * math.pi / T)) / 2) * (ds - 1.0) + 1.0
        if kind == 'linear':
            return lambda e: (1 - e / max(T - 1, 1)) * (1. - ds) + ds
        return lambda e: ((1 + math.cos(e * math.pi / T)) / 2) * (1. - ds) + ds

    def warmup(self, step_in_total):
        """lr / momentum / accumulate interpolation during the first warmup_steps (train_yolov5.py:437-456)"""
        hyp = self.hyp
        if hyp['do_warmup'] and step_in_total < hyp['warmup_steps']:
            xs = [0., hyp['warmup_steps']]
            self.accumulate = max(1, np.interp(step_in_total, xs, [1, hyp['accumulate_loss_step'] / hyp['batch_size'] / get_world_size()]).round())
            for j, g in enumerate(self.optimizer.param_groups):
                g['lr'] = np.interp(step_in_total, xs, [0., g['initial_lr']] if j != 2 else [hyp['warmup_bias_max_lr'], g['initial_lr']])
                g['momentum'] = np.interp(step_in_total, xs, [hyp['warmup_momentum'], hyp['momentum']])

    # ------------------------------------------------------------------ hot loop (train_yolov5.py:295-375)
    def step(self):
        hyp = self.hyp
        step_in_total = self.start_epoch * len(self.train_dataloader)
        for epoch in range(self.start_epoch, hyp['total_epoch']):
            self.model.train()
            # LambdaLR semantics (train_yolov5.py:152-164): 'initial_lr' stays the base learning rate, the per-epoch factor only
            # scales 'lr'; the warm-up interpolates towards the unscheduled 'initial_lr' (train_yolov5.py:437-456)
            base = self.lr_scheduler_fn(epoch)
            for g in self.optimizer.param_groups:
                g['initial_lr'] = hyp['lr']
                g['lr'] = g['initial_lr'] * base
            for i, x in enumerate(self.train_dataloader):
                step_in_total += 1
                self.warmup(step_in_total)
                boundary = (i + 1) % self.accumulate == 0
                ctx = self.dp.no_sync() if (self.dp is not None and not boundary) else _Null()
                with ctx:
                    if hyp.get('prefetch_assign', False):
                        # the target assignment on a side stream beside the forward pass (it needs the targets only); measured on the
                        # YOLOv5s step it does not pay (profiles/r04_step_experiments.txt, r): off unless the configuration asks for it
                        self.loss_fcn.prefetch_assign(x['ann'])
                    stage_preds = self.model(x['img'])
                    loss_dict = self.loss_fcn(stage_preds, x['ann'])
                    loss_dict['tot_loss'].backward()
                if boundary:
                    if self.CLIP_GRAD_NORM is not None:
                        self.optimizer.clip_grad_norm_(self.CLIP_GRAD_NORM)
                    self.optimizer.step()
                    self.optimizer.zero_grad()
                    if self.ema_model is not None:
                        self.ema_model.update(self.model)
                self.history.append({k: (float(v) if not torch.is_tensor(v) else float(v.item())) for k, v in loss_dict.items()})
                if self.rank == 0 and (i % 10 == 0 or i == len(self.train_dataloader) - 1):
                    h = self.history[-1]
                    print(f"epoch {epoch + 1}/{hyp['total_epoch']} step {i + 1}/{len(self.train_dataloader)} "
                          f"{self.log_line(h)} lr {self.optimizer.param_groups[0]['lr']:.5f}", flush=True)
            self.save_model(epoch + 1, step_in_total=step_in_total, loss_dict=self.history[-1])
            if (epoch + 1) % hyp['validation_every'] == 0:
                self.after_epoch(epoch + 1)

    # ------------------------------------------------------------------ evaluation (train_yolov5.py:676-769)
    def after_epoch(self, epoch):
        if self.is_distributed:
            all_reduce_norm(self.model)          # average BN states over ranks before evaluation (:687)
        model = self.ema_model.ema if self.ema_model is not None else self.model
        was_training = model.training
        model.eval()
        self.validate.yolo = model
        all_gts, all_preds = [], []
        for x in self.val_dataloader:
            outs = self.validate(x['img'])
            ann = x['ann'].cpu().numpy()
            for b, o in enumerate(outs):
                gt = ann[b][ann[b][:, 4] >= 0][:, :5]
                all_gts.append(gt)
                all_preds.append(o.numpy() if o is not None else np.zeros((0, 6), np.float32))
        m, m50, mp, mr = mAP_v2(all_gts, all_preds).get_mean_metrics() if any(len(p) for p in all_preds) else (0., 0., 0., 0.)
        self.last_metrics = dict(map=m, map50=m50, precision=mp, recall=mr, n_pred=int(sum(len(p) for p in all_preds)))
        if self.rank == 0:
            print(f"[eval] epoch {epoch}: mAP {m:.4f} mAP50 {m50:.4f} P {mp:.4f} R {mr:.4f} ({self.last_metrics['n_pred']} boxes)", flush=True)
        model.train(was_training)
        synchronize()

    # ------------------------------------------------------------------ checkpoints (train_yolov5.py:546-629)
    def save_model(self, cur_epoch, filename=None, step_in_total=None, loss_dict=None, save_optimizer=True):
        if self.rank != 0 or cur_epoch % self.hyp['save_ckpt_every'] != 0:
            return None
        path = self.cwd / 'checkpoints' / (f'{filename}.pth' if filename else f'{self.CKPT_PREFIX}_{self.hyp["model_type"]}_epoch_{cur_epoch}.pth')
        path.parent.mkdir(parents=True, exist_ok=True)
        hyp_save = {k: v for k, v in self.hyp.items()}
        state = {"model_state_dict": self.model.state_dict(),
                 "optim_state_dict": self.optimizer.state_dict() if save_optimizer else None,
                 "optim_type": self.hyp['optimizer'], "scaler_state_dict": None,
                 "lr_scheduler_type": self.hyp['scheduler_type'], "lr_scheduler_state_dict": {"last_epoch": cur_epoch},
                 "loss": loss_dict, "epoch": cur_epoch, "step": step_in_total,
                 "ema": self.ema_model.ema.state_dict() if self.ema_model is not None else None,
                 "ema_update_num": self.ema_model.update_num if self.ema_model is not None else 0, "hyp": hyp_save}
        torch.save(state, str(path))
        self.last_ckpt = str(path)
        return str(path)

    def load_model(self, path, load_optimizer=True):
        state = torch.load(path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(state["model_state_dict"])
        if self.ema_model is not None and state.get("ema") is not None:
            self.ema_model.ema.load_state_dict(state["ema"])
            self.ema_model.update_num = state.get("ema_update_num", 0)
        if load_optimizer and state.get("optim_state_dict") is not None and state.get("optim_type") == self.hyp['optimizer']:
            self.optimizer.load_state_dict(state["optim_state_dict"])
        self.start_epoch = state.get("epoch", 0)
        return state


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def main(argv=None, training_cls=None, default_cfg=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=default_cfg or os.path.join(ROOT, "config", "train_yolov5.yaml"))
    ap.add_argument("--epochs", type=int)
    ap.add_argument("--img", type=int)
    ap.add_argument("--batch", type=int)
    ap.add_argument("--steps-per-epoch", type=int)
    ap.add_argument("--model-type")
    ap.add_argument("--data", choices=["tensor", "dataset", "shapes"], help="tensor: random-noise batches; dataset: synthetic images "
                    "through DataLoader + fixed_imgsize_collate_fn + DataPrefetcher (the reference's data path); shapes: a learnable "
                    "task (coloured rectangles, colour = class) that shows mAP rising")
    args = ap.parse_args(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(get_local_rank())
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", get_local_rank()))
    hyp = Config().get_config(args.cfg)
    if args.epochs: hyp['total_epoch'] = args.epochs                     # noqa: E701
    if args.img: hyp['input_img_size'] = [args.img, args.img]            # noqa: E701
    if args.batch: hyp['batch_size'] = args.batch; hyp['accumulate_loss_step'] = args.batch   # noqa: E701,E702
    if args.steps_per_epoch: hyp['steps_per_epoch'] = args.steps_per_epoch  # noqa: E701
    if args.model_type: hyp['model_type'] = args.model_type              # noqa: E701
    if args.data: hyp['data_source'] = args.data                         # noqa: E701
    if training_cls is not None:                                         # train_yolox.py:808: Training(hyp)
        t = training_cls(hyp)
    else:
        anchors = torch.from_numpy(COCO_ANCHORS.copy())                  # train_yolov5.py:819
        t = Training(anchors, hyp)
    t.step()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return t


if __name__ == "__main__":
    main()
