#!/usr/bin/env python3
"""Why is the CPU baseline 4x slower per image at batch 64 than at batch 4?  The oracle's train step at
batch 4 / 16 / 64 on the threads bench.py would use, split into forward / loss / backward / SGD, with the process's CPU time next
to the wall time (CPU time / wall = threads that actually ran).   usage: cpu_leg_scaling.py [batches ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle.v5loss import V5LossOracle
from oracle.v5net import V5NetOracle
from yoloseries_amd import models
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets

cores = bench._host_threads()
torch.set_num_threads(cores)
print(f"threads {cores}  affinity {len(os.sched_getaffinity(0))}  cpu_count {os.cpu_count()}", flush=True)
for B in [int(a) for a in sys.argv[1:]] or [4, 16, 64]:
    torch.manual_seed(0)
    net = V5NetOracle(models.YOLOV5Small(3, 80).state_dict(), train=True)
    lossf = V5LossOracle(COCO_ANCHORS, bench.make_hyp("cpu", 640, B))
    x = torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(0))
    t = synth_targets(B, 640, 80, 20, seed=1)
    rows = []
    for it in range(3):
        c0 = time.process_time()
        t0 = time.time(); p = net(x)
        t1 = time.time(); out = lossf(p, t)
        t2 = time.time(); out["tot_loss"].backward()
        t3 = time.time(); net.sgd_step(0.01)
        t4 = time.time()
        rows.append((t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, time.process_time() - c0))
    w, f, l, b, s, c = rows[-1]
    print(f"batch {B:3d}: {w:7.3f} s/step = {B / w:6.2f} img/s | per image: forward {1e3 * f / B:7.1f} ms  loss {1e3 * l / B:6.1f}  backward {1e3 * b / B:7.1f}  "
          f"sgd {1e3 * s / B:5.1f} | cpu/wall {c / w:5.1f}", flush=True)
