#!/usr/bin/env python3
"""Host time to enqueue one train step (forward, loss, backward, clip, SGD, EMA): every step starts with an empty launch queue
(device sync before it), so the host never waits for the GPU while enqueueing.  YH_EXEC=0 launches every kernel from Python
(one ctypes call each) instead of replaying the compiled command arrays (csrc/exec.hip)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.trainer import ExponentialMovingAverageModel
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
dev = torch.device('cuda:0')
for B in (64, 16):
    torch.manual_seed(0)
    m = models.YOLOV5Small(3, 80).to(dev).train()
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, 640, B))
    opt = FlatSGD(m, lr=0.01, momentum=0.937, weight_decay=1e-4, nesterov=True)
    ema = ExponentialMovingAverageModel(m)
    x = torch.rand(B, 3, 640, 640, device=dev)
    t = torch.from_numpy(synth_targets(B, 640, 80, 20, seed=1)).to(dev)
    parts = {}
    def step():
        h = time.perf_counter(); y = m(x); parts['forward'] = parts.get('forward', 0) + time.perf_counter() - h
        h = time.perf_counter(); out = lossf(y, t); parts['loss'] = parts.get('loss', 0) + time.perf_counter() - h
        h = time.perf_counter(); out['tot_loss'].backward(); parts['backward'] = parts.get('backward', 0) + time.perf_counter() - h
        h = time.perf_counter(); opt.clip_grad_norm_(10.0); opt.step(); opt.zero_grad(); ema.update(m)
        parts['optimizer+ema'] = parts.get('optimizer+ema', 0) + time.perf_counter() - h
    for _ in range(5): step()
    torch.cuda.synchronize()
    parts.clear()
    hs, n = [], 10
    t0 = time.perf_counter()
    for _ in range(n):
        torch.cuda.synchronize()
        h0 = time.perf_counter(); step(); hs.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    print(f"B={B}: host enqueue per step {sum(hs)/n*1e3:.2f} ms (min {min(hs)*1e3:.2f}): " +
          ", ".join(f"{k} {v/n*1e3:.2f}" for k, v in parts.items()))
