#!/usr/bin/env python3
"""Back-to-back launch cost of a trivial kernel on one stream (the floor every dependent kernel of the step pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
dev = torch.device("cuda:0")
buf = torch.zeros(1024, dtype=torch.int32, device=dev)
big = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=dev)
for name, t in (("4 KB fill", buf), ("256 MB fill", big)):
    for _ in range(10):
        hipk.fill_zero(t)
    torch.cuda.synchronize()
    n = 2000 if t is buf else 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        hipk.fill_zero(t)
    e1.record()
    t1 = time.perf_counter()
    e1.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1000:.2f} us per launch on the GPU, host enqueue {(t1 - t0) / n * 1e6:.2f} us")
