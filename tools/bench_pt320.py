#!/usr/bin/env python3
"""conv_pt_kernel's inference form on YOLOv5x's 320-channel 1x1 layers (bottleneck conv_bn_act_1: 320 -> 320 with folded BatchNorm + SiLU)
against the other kernel families, isolated.   usage: bench_pt320.py [H=80] [batch=128] [N=320]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

H = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 320
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, 320, device=dev).to(torch.bfloat16)
w = torch.randn(N, 320, 1, 1, device=dev) / 320 ** 0.5
wp = hipk.pack_weight_fwd(w)
out = torch.zeros(B, H, H, N, dtype=torch.bfloat16, device=dev)
scale, shift = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, H, H, H, 1, 1, 0, wp, N, hipk.full(out), scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
buf = C.create_string_buffer(96)
M = B * H * H
nbytes = 2.0 * M * (320 + N)
fl = 2.0 * M * N * 320
cands = []
for algo, tk in ((1, 0), (1, 32), (2, 0), (3, 0), (3, 32), (4, 0), (13, 0)):
    d.algo, d.tile_k = algo, tk
    lib().yh_conv_kernel_name(C.byref(d), buf, 96)
    kn = buf.value.decode()
    if (algo in (2, 3, 4) and "conv_v3" not in kn) or (algo == 13 and "conv_pt" not in kn):
        continue
    cands.append((algo, tk, kn))
times = {c: [] for c in cands}
for _ in range(5):
    for c in cands:
        d.algo, d.tile_k = c[0], c[1]
        hipk.conv_launch(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hipk.conv_launch(d)
        e1.record()
        e1.synchronize()
        times[c].append(e0.elapsed_time(e1) / 5 * 1000)
for c in cands:
    us = float(np.median(times[c]))
    print(f"algo {c[0]:2d} tile_k {c[1]:2d} {c[2]:50s} {us:8.1f} us  {nbytes / us / 1e3:6.0f} GB/s  {fl / us / 1e6:5.0f} TFLOP/s")
