#!/usr/bin/env python3
"""Kernel families of yh_conv_igemm side by side on representative layer shapes: register-staged conv_v2 (algo 1) against the
LDS-DMA ring kernel conv_v3 with its 256x128 / 128x128 / 128x64 tiles (algo 2..4).
usage: bench_algos.py [v5s|v5l|v5x1280] [fwd|dgrad|dgrad3|eval] [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

which = sys.argv[1] if len(sys.argv) > 1 else "v5s"
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
if which == "v5s":
    B = 64
    shapes = [("s1_conv", 320, 32, 64, 3, 2), ("s1_b_3x3", 160, 32, 32, 3, 1), ("s1_cba12", 160, 64, 64, 1, 1), ("s2_conv", 160, 64, 128, 3, 2), ("s2_b_3x3", 80, 64, 64, 3, 1),
              ("s2_cba12", 80, 128, 128, 1, 1), ("s3_conv", 80, 128, 256, 3, 2), ("s3_b_3x3", 40, 128, 128, 3, 1),
              ("s3_b_1x1", 40, 128, 128, 1, 1), ("s3_cba12", 40, 256, 256, 1, 1), ("s4_conv", 40, 256, 512, 3, 2),
              ("s4_b_3x3", 20, 256, 256, 3, 1), ("s4_cba3", 20, 512, 512, 1, 1), ("spp_cba2", 20, 1024, 512, 1, 1)]
elif which == "v5l":
    B = 64
    shapes = [("s1_conv", 320, 64, 128, 3, 2), ("s1_b_3x3", 160, 64, 64, 3, 1), ("s2_conv", 160, 128, 256, 3, 2),
              ("s2_b_3x3", 80, 128, 128, 3, 1), ("s3_conv", 80, 256, 512, 3, 2), ("s3_b_3x3", 40, 256, 256, 3, 1),
              ("s3_cba12", 40, 512, 512, 1, 1), ("s4_conv", 40, 512, 1024, 3, 2), ("s4_b_3x3", 20, 512, 512, 3, 1),
              ("s4_cba3", 20, 1024, 1024, 1, 1)]
else:
    B = 16
    shapes = [("s1_conv", 640, 80, 160, 3, 2), ("s1_b_3x3", 320, 80, 80, 3, 1), ("s1_b_1x1", 320, 80, 80, 1, 1),
              ("s1_cba12", 320, 160, 160, 1, 1), ("s2_conv", 320, 160, 320, 3, 2), ("s2_b_3x3", 160, 160, 160, 3, 1),
              ("s2_b_1x1", 160, 160, 160, 1, 1), ("s2_cba12", 160, 320, 320, 1, 1), ("s3_conv", 160, 320, 640, 3, 2),
              ("s3_b_3x3", 80, 320, 320, 3, 1), ("s3_cba12", 80, 640, 640, 1, 1), ("s4_conv", 80, 640, 1280, 3, 2),
              ("s4_b_3x3", 40, 640, 640, 3, 1)]

B = int(os.environ.get("BA_BATCH", B))                       # BA_BATCH / BA_ONLY: another batch size / only these shapes
if os.environ.get("BA_ONLY"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["BA_ONLY"].split(",")]


def kname(d):
    buf = C.create_string_buffer(96)
    lib().yh_conv_kernel_name(C.byref(d), buf, 96)
    return buf.value.decode()


for name, H, Cin, Cout, k, s in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    M = B * Ho * Ho
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    if mode in ("fwd", "eval"):
        out = torch.zeros(B, Ho, Ho, Cout, dtype=torch.bfloat16, device=dev)
        wp = hipk.pack_weight_fwd(w)
        if mode == "eval":
            sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
            res = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16) if os.environ.get("BA_RES") else None      # BA_RES=1: with a residual
            d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Ho, H, H, k, s, p, wp, Cout, hipk.full(out), scale=sc, shift=sh,
                               act=hipk.YH_ACT_SILU, res=hipk.full(res) if res is not None else None)
        else:
            d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Ho, H, H, k, s, p, wp, Cout, hipk.full(out))
            stats = torch.zeros(4096, 2, wp.shape[0], device=dev)
            d.stats = stats.data_ptr()
    else:
        gy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
        gx = torch.zeros(B, H, H, Cin, dtype=torch.bfloat16, device=dev)
        wd = hipk.pack_weight_dgrad(w)
        d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, H, Ho, Ho, k, s, p, wd, Cin, hipk.full(gx))
        if mode == "dgrad3":             # with the fused BatchNorm-backward reduction of the producer layer (EPI 3)
            z = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
            ws = torch.cat([torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)])
            slab = torch.zeros(8192, 2, Cin, device=dev)
            d.bnr_z, d.bnr_ldz, d.bnr_C, d.bnr_ws, d.bnr_part = z.data_ptr(), Cin, Cin, ws.data_ptr(), slab.data_ptr()
    fl = 2.0 * M * Cout * Cin * k * k
    res = []
    for algo in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14):
        d.algo = algo
        d.tile_k = int(os.environ.get("BA_TILE_K", 0)) if algo >= 7 else 0
        d.tile_n = int(os.environ.get("BA_TILE_N", 0)) if algo >= 7 else 0
        kn = kname(d)
        if (algo in (2, 3, 4) and "conv_v3" not in kn) or (algo == 5 and "conv_halo_kernel" not in kn) or (algo == 6 and "conv_halo160" not in kn) or \
                (algo == 7 and "conv_dg2" not in kn) or (algo == 8 and "conv_p3" not in kn) or (algo == 9 and "conv_h80" not in kn) or (algo == 10 and "conv_pw" not in kn) or \
                (algo == 13 and "conv_pt" not in kn) or (algo == 14 and "conv_v3_kernel<256, 256" not in kn):
            res.append("      -")
            continue
        for _ in range(2):
            hipk.conv_launch(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hipk.conv_launch(d)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f"{fl / ms / 1e9:7.0f}")
    print(f"{mode:5s} {name:10s} {H:4d}^2 {Cin:4d}->{Cout:4d} k{k}s{s}  TFLOP/s  v2 {res[0]} | v3-256x128 {res[1]} | v3-128x128 {res[2]} | v3-128x64 {res[3]} | halo {res[4]} | halo160 {res[5]} | dg2 {res[6]} | p3 {res[7]} | h80 {res[8]} | pw {res[9]} | pt {res[10]} | v3-256x256 {res[11]}", flush=True)
    del x, w
