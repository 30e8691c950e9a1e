#!/usr/bin/env python3
"""Functional check: a few train steps + an eval forward of several model sizes on odd shapes (GPU)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
dev = torch.device('cuda:0')
cases = [("small", models.YOLOV5Small, (4, 320, 192)), ("xlarge", models.YOLOV5XLarge, (2, 128, 128)), ("xlarge", models.YOLOV5XLarge, (3, 192, 320)),
         ("middle", models.YOLOV5Middle, (3, 96, 160)), ("large", models.YOLOV5Large, (5, 128, 128))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, cls, shape in cases:
    B, H, W = shape
    torch.manual_seed(0)
    m = cls(3, 80).to(dev).train()
    hyp = bench.make_hyp(dev, H, B); hyp['input_img_size'] = [H, W]
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), hyp)
    opt = FlatSGD(m, lr=0.005, momentum=0.9)
    x = torch.rand(B, 3, H, W, device=dev)
    t = torch.from_numpy(synth_targets(B, min(H, W), 80, 20, seed=5)).to(dev)
    ls = []
    for i in range(4):
        preds = m(x)
        fin = [bool(torch.isfinite(p.float()).all()) for p in preds]
        out = lossf(preds, t); out['tot_loss'].backward()
        g = m._yh_last_flat_grad
        gfin = bool(torch.isfinite(g).all())
        opt.clip_grad_norm_(10.0); opt.step(); opt.zero_grad(); ls.append(float(out['tot_loss'].detach()))
        if i == 0:
            print("   step0 preds finite", fin, "grad finite", gfin)
    print(name, shape, [round(v, 3) for v in ls], "finite", all(np.isfinite(ls)))
