#!/usr/bin/env python3
"""Per-phase wall time of one train step with a device sync after every phase (so phases cannot overlap) next to the
free-running step time and the host's enqueue time: shows which phase bounds a workload and whether the host does.
usage: python tools/phase_times.py [train|yolox] [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss, YOLOXLoss
from yoloseries_amd.trainer import ExponentialMovingAverageModel
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
wl = sys.argv[1] if len(sys.argv) > 1 else "train"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device('cuda:0')
torch.manual_seed(0)
if wl == "yolox":
    m = models.YOLOXSmall(1, 3, 80).to(dev).train()
    hyp = dict(device=dev, num_class=80, input_img_size=[640, 640], batch_size=B, use_focal_loss=False, focal_loss_gamma=1.5,
               focal_loss_alpha=0.25, iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0, cof_loss_scale=1.0,
               class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0, num_anchors=1, iou_type="ciou", topk=13,
               center_radius=3, num_stage=3, loss_items_on_device=True)
    lossf = YOLOXLoss(hyp)
else:
    m = models.YOLOV5Small(3, 80).to(dev).train()
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, 640, B))
opt = FlatSGD(m, lr=0.000625 * B, momentum=0.937, weight_decay=1e-4, nesterov=True)
ema = ExponentialMovingAverageModel(m)
x = torch.rand(B, 3, 640, 640, device=dev)
t = torch.from_numpy(synth_targets(B, 640, 80, 20, seed=1)).to(dev)
S = torch.cuda.synchronize
def step(acc=None, sync=False):
    def mark(name, h):
        if sync: S()
        if acc is not None: acc[name] = acc.get(name, 0) + time.perf_counter() - h
    h = time.perf_counter(); y = m(x); mark('forward', h)
    h = time.perf_counter(); out = lossf(y, t.clone()); mark('loss', h)
    h = time.perf_counter(); out['tot_loss'].backward(); mark('backward', h)
    h = time.perf_counter(); opt.clip_grad_norm_(10.0); opt.step(); opt.zero_grad(); ema.update(m); mark('optimizer+ema', h)
for _ in range(6): step()
S()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
S()
free = (time.perf_counter() - t0) / n
ph, host = {}, {}
for _ in range(n): S(); step(ph, sync=True)
for _ in range(n): S(); step(host, sync=False)
S()
print(f"{wl} B={B}: free-running {free*1e3:.2f} ms/step")
print("  synced phases : " + ", ".join(f"{k} {v/n*1e3:.2f}" for k, v in ph.items()) + f"  (sum {sum(ph.values())/n*1e3:.2f})")
print("  host enqueue  : " + ", ".join(f"{k} {v/n*1e3:.2f}" for k, v in host.items()) + f"  (sum {sum(host.values())/n*1e3:.2f})")
