#!/usr/bin/env python3
"""conv_pt_kernel (yh_conv_desc.algo 13) against the ring / register-staged kernels (algo 1..4, both k-step widths) on the 1x1 layers of
the YOLOv5s / YOLOv5l training step at batch 64: forward with the BatchNorm partial sums, data gradient with the fused
BatchNorm-backward reduction, one- and two-segment inputs.  Interleaved rounds in one process, median per kernel; GB/s on the
algorithmic bytes (input once, output once, z once for the fused reduction).
usage: bench_pt.py [v5s|v5l] [rounds] [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

which = sys.argv[1] if len(sys.argv) > 1 else "v5s"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
B = int(os.environ.get("BP_BATCH", 64))
# name, map, (C0, C1, ups0), N
if which == "v5s":
    shapes = [("s2_cba12", 80, (128, 0, 0), 128), ("s2_cba3", 80, (64, 64, 0), 128), ("s3_b_cba1", 40, (128, 0, 0), 128),
              ("s3_cba12", 40, (256, 0, 0), 256), ("s3_cba3", 40, (128, 128, 0), 256), ("h2_conv", 40, (256, 0, 0), 128),
              ("h2_cba12", 80, (128, 128, 1), 128), ("h3_cba12", 40, (128, 128, 0), 256), ("det_s", 80, (128, 0, 0), 255), ("det_m", 40, (256, 0, 0), 255),
              ("s4_cba12", 20, (512, 0, 0), 512), ("s4_cba3", 20, (256, 256, 0), 512), ("h1_conv", 20, (512, 0, 0), 256), ("h1_cba12", 40, (256, 256, 1), 256)]
else:
    shapes = [("s1_cba12", 160, (128, 0, 0), 128), ("s1_cba3", 160, (64, 64, 0), 128), ("s2_b_cba1", 80, (128, 0, 0), 128),
              ("s2_cba12", 80, (256, 0, 0), 256), ("s2_cba3", 80, (128, 128, 0), 256), ("s3_b_cba1", 40, (256, 0, 0), 256),
              ("s3_cba12", 40, (512, 0, 0), 512), ("s3_cba3", 40, (256, 256, 0), 512), ("h2_conv", 40, (512, 0, 0), 256), ("h2_cba12", 80, (256, 256, 1), 256)]
if os.environ.get("BP_ONLY"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["BP_ONLY"].split(",")]


def kname(d):
    buf = C.create_string_buffer(96)
    lib().yh_conv_kernel_name(C.byref(d), buf, 96)
    return buf.value.decode()


def timeit(d):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hipk.conv_launch(d)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


for name, H, (C0, C1, ups0), N in shapes:
    M = B * H * H
    Ct = C0 + C1
    segs = [hipk.Slice(torch.randn(B, H >> ups0, H >> ups0, C0, device=dev).to(torch.bfloat16), 0, C0, ups0)]
    if C1:
        segs.append(hipk.full(torch.randn(B, H, H, C1, device=dev).to(torch.bfloat16)))
    w = torch.randn(N, Ct, 1, 1, device=dev) / Ct ** 0.5
    wp = hipk.pack_weight_fwd(w)
    Nr = (N + 7) // 8 * 8
    for mode in ("fwd", "dgrad3"):
        if mode == "dgrad3" and (N % 8 or C1 or ups0):
            continue
        out = torch.zeros(B, H, H, Nr, dtype=torch.bfloat16, device=dev)
        if mode == "fwd":
            d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, H, H, H, 1, 1, 0, wp, N, hipk.full(out))
            if N % 8:
                bias = torch.zeros(N, device=dev)
                d.bias = bias.data_ptr()
            else:
                stats = torch.zeros(8192, 2, wp.shape[0], device=dev)
                d.stats = stats.data_ptr()
            nbytes = 2.0 * (sum(B * (H >> s.ups) ** 2 * s.C for s in segs) + M * Nr)
        else:
            d = hipk.conv_desc(segs, hipk.YH_CONV_DGRAD, B, H, H, H, H, 1, 1, 0, wp, N, hipk.full(out))
            z = torch.randn(B, H, H, N, device=dev).to(torch.bfloat16)
            ws = torch.cat([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)])
            slab = torch.zeros(16384, 2, N, device=dev)
            d.bnr_z, d.bnr_ldz, d.bnr_C, d.bnr_ws, d.bnr_part = z.data_ptr(), N, N, ws.data_ptr(), slab.data_ptr()
            nbytes = 2.0 * (M * Ct + 2 * M * Nr)
        cands = []
        for algo, tk in ((1, 0), (1, 32), (2, 0), (3, 0), (3, 32), (4, 0), (4, 32), (13, 0)):
            d.algo, d.tile_k, d.grid_cap = algo, tk, 0
            kn = kname(d)
            if (algo in (2, 3, 4) and "conv_v3" not in kn) or (algo == 13 and "conv_pt" not in kn):
                continue
            cands.append((algo, tk, kn))
        times = {c: [] for c in cands}
        for c in cands:
            d.algo, d.tile_k = c[0], c[1]
            hipk.conv_launch(d)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for c in cands:
                d.algo, d.tile_k = c[0], c[1]
                times[c].append(timeit(d))
        med = {c: float(np.median(v)) for c, v in times.items()}
        old = min((v, c) for c, v in med.items() if c[0] != 13)
        new = [v for c, v in med.items() if c[0] == 13]
        fl = 2.0 * M * N * Ct
        line = f"{mode:6s} {name:10s} {H:3d}^2 {C0}+{C1}{'u' if ups0 else ''} -> {N:3d}: best old {old[0]:7.1f} us ({nbytes / old[0] / 1e3:6.0f} GB/s, {fl / old[0] / 1e6:5.0f} TF/s) {old[1][2][:44]:44s}"
        if new:
            line += f" | pt {new[0]:7.1f} us ({nbytes / new[0] / 1e3:6.0f} GB/s)  x{old[0] / new[0]:.2f}"
        print(line, flush=True)
