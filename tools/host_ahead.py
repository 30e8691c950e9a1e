#!/usr/bin/env python3
"""Is the host ahead of the GPU in the train loop?  Host time of each step() call (no synchronisation in between) against the GPU time
per step, and the host time of the step's sections.  usage: host_ahead.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.trainer import ExponentialMovingAverageModel
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
B, img = 64, 640
torch.manual_seed(0)
model = models.YOLOV5Small(3, 80).to(dev).train()
lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
opt = FlatSGD(model, lr=0.002, momentum=0.937, weight_decay=1e-4, nesterov=True)
ema = ExponentialMovingAverageModel(model)
x = torch.rand(B, 3, img, img, device=dev)
t = torch.from_numpy(synth_targets(B, img, 80, 20, seed=1)).to(dev)
sec = {}


def step():
    t0 = time.perf_counter()
    p = model(x)
    t1 = time.perf_counter()
    out = lossf(p, t)
    t2 = time.perf_counter()
    out["tot_loss"].backward()
    t3 = time.perf_counter()
    opt.clip_grad_norm_(10.0)
    opt.step()
    opt.zero_grad()
    t4 = time.perf_counter()
    if not os.environ.get("HA_NO_EMA"):
        ema.update(model)
    t5 = time.perf_counter()
    for k, v in (("forward", t1 - t0), ("loss", t2 - t1), ("backward", t3 - t2), ("optimizer", t4 - t3), ("ema", t5 - t4)):
        sec.setdefault(k, []).append(v * 1e3)


for _ in range(5):
    step()
torch.cuda.synchronize()
sec.clear()
h0 = time.perf_counter()
hs = []
for _ in range(n):
    a = time.perf_counter()
    step()
    hs.append((time.perf_counter() - a) * 1e3)
host_total = (time.perf_counter() - h0) * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - h0) * 1e3
print("host ms per step():", " ".join(f"{v:.2f}" for v in hs))
print(f"host loop {host_total:.1f} ms, with the final synchronize {total:.1f} ms ({total / n:.2f} per step)")
for k, v in sec.items():
    v = sorted(v)
    print(f"  {k:10s} median {v[len(v) // 2]:.3f} ms  max {v[-1]:.3f}")
