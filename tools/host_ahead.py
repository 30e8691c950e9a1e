#!/usr/bin/env python3
"""Is the host ahead of the GPU in the train loop?  Host time of each step() call (no synchronisation in between) against the GPU time
per step, the host time of the step's sections, and — OFF the profiler — the device-side gaps at the hand-overs a kernel trace shows
as holes (profiles/r05_step_trace.txt: 720 us between gather_f32_kernel and sumsq_part_kernel): an event recorded right behind the
launch in front of the hole and one right before the launch behind it; their distance on the device is the hole (0 when the host is
ahead: both are then processed back to back).  usage: host_ahead.py [steps]"""
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
from yoloseries_amd import hipk
holes = {"gather_f32 -> sumsq (backward end -> clip)": [], "sgd -> ema_advance": [], "ema -> next step's first launch": []}
_ev = {}
_orig = {k: getattr(hipk, k) for k in ("gather_f32", "sumsq", "sgd_step_dev", "ema_advance", "ema_update_dev", "pack_bf16")}


def _after(name, key):
    def f(*a, **k):
        r = _orig[name](*a, **k)
        e = torch.cuda.Event(enable_timing=True); e.record(); _ev[key] = e
        return r
    return f


def _before(name, key, hole):
    def f(*a, **k):
        if key in _ev:
            e = torch.cuda.Event(enable_timing=True); e.record()
            holes[hole].append((_ev.pop(key), e))
        return _orig[name](*a, **k)
    return f


if not os.environ.get("HA_NO_EVENTS"):
    hipk.gather_f32 = _after("gather_f32", "g")          # the last launch of backward() is the packed-gradient gather (pack.repack()'s gather has no sumsq behind it: popped below)
    hipk.sumsq = _before("sumsq", "g", "gather_f32 -> sumsq (backward end -> clip)")
    hipk.sgd_step_dev = _after("sgd_step_dev", "s")
    hipk.ema_advance = _before("ema_advance", "s", "sgd -> ema_advance")
    hipk.ema_update_dev = _after("ema_update_dev", "e")
    hipk.pack_bf16 = _before("pack_bf16", "e", "ema -> next step's first launch")


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
for v in holes.values():
    del v[:]
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
for k, prs in holes.items():
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in prs[-n:])
    if us:
        print(f"  device gap {k}: median {us[len(us) // 2]:.1f} us  max {us[-1]:.1f} us  ({len(us)} steps; includes the ~2 us of the two event packets)")
