#!/usr/bin/env python3
"""Soak of the step's deterministic parts at the judged shape: N forward + backward passes of one YOLOv5s (or --large) on one input,
head outputs, loss and every BatchNorm weight / bias gradient compared bit for bit with the first pass (tests/test_gpu_model.py
runs five passes; this runs hundreds).  usage: step_determinism.py [passes] [small|large]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
size = sys.argv[2] if len(sys.argv) > 2 else "small"
dev = torch.device("cuda:0")
B, img = 64, 640
torch.manual_seed(0)
x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(11)).to(dev)
t = torch.from_numpy(synth_targets(B, img, 80, 12, seed=12, min_boxes=2)).to(dev)
m = (models.YOLOV5Small if size == "small" else models.YOLOV5Large)(3, 80).to(dev).train()
bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
first, bad = None, []
for r in range(n):
    for p_ in m.parameters():
        p_.grad = None
    outs = m(x)
    loss = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))(outs, t)["tot_loss"]
    loss.backward()
    torch.cuda.synchronize()
    cur = [o.detach().clone() for o in outs] + [loss.detach().clone()] + [b.weight.grad.clone() for b in bns] + [b.bias.grad.clone() for b in bns]
    if first is None:
        first = cur
    else:
        d = [i for i, (a, b) in enumerate(zip(first, cur)) if not torch.equal(a, b)]
        if d:
            bad.append((r, d[:6]))
            print(f"pass {r}: tensors {d[:6]} differ", flush=True)
print(f"{size}: {n} passes, {len(bad)} with differences")
