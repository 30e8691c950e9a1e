#!/usr/bin/env python3
"""Sweep the split-M factor of yh_conv_wgrad over representative YOLOv5s layer shapes (B=64, 640x640).
usage: sweep_wgrad.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B = 64
#        name              H    Cin  Cout k  s
shapes = [("focus",         320, 16,  32,  3, 1),
          ("s1_conv",       320, 32,  64,  3, 2),
          ("s1_b_3x3",      160, 32,  32,  3, 1),
          ("s1_cba12",      160, 64,  64,  1, 1),
          ("s2_conv",       160, 64,  128, 3, 2),
          ("s2_b_3x3",      80,  64,  64,  3, 1),
          ("s2_cba12",      80,  128, 128, 1, 1),
          ("s3_conv",       80,  128, 256, 3, 2),
          ("s3_b_3x3",      40,  128, 128, 3, 1),
          ("s3_b_1x1",      40,  128, 128, 1, 1),
          ("s3_cba12",      40,  256, 256, 1, 1),
          ("s4_conv",       40,  256, 512, 3, 2),
          ("s4_b_3x3",      20,  256, 256, 3, 1),
          ("s4_cba3",       20,  512, 512, 1, 1),
          ("spp_cba2",      20,  1024, 512, 1, 1),
          ("h2_cba12_seg",  80,  128, 128, 1, 1), ("s3_cba3_seg", 40, 128, 256, 1, 1), ("s4_cba12_seg", 20, 256, 512, 1, 1),
          ("s4_b_1x1",      20,  256, 256, 1, 1)]
if os.environ.get("WG_ONLY"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["WG_ONLY"].split(",")]
L = lib()
for name, H, Cin, Cout, k, s in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    M = B * Ho * Ho
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    gy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(Cout, k * k * Cin, device=dev)
    ntile = L.yh_conv_wgrad_tiles(Cout, k * k * Cin)
    fl = 2.0 * M * Cout * Cin * k * k
    by = 2.0 * (B * H * H * Cin + M * Cout)
    res = []
    for tot in (128, 256, 512, 768, 1024, 1536, 3072):
        splits = max(1, min((M + 255) // 256, (tot + ntile - 1) // ntile))
        if os.environ.get("WG_TK") == "40":
            splits = tot                     # patch form: `splits` caps the persistent blocks
        d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, Ho, Ho, H, H, k, s, p, dw, splits)
        d.tile_k = int(os.environ.get("WG_TK", "0"))
        for _ in range(2):
            hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hipk.wgrad_launch(d)
        e1.record()
        torch.cuda.synchronize()
        res.append((tot, splits, e0.elapsed_time(e1) / iters * 1000))
    best = min(res, key=lambda r: r[2])
    print(f"{name:10s} M={M:8d} N={Cout:4d} K={k*k*Cin:5d} tiles={ntile:3d} | " + " ".join(f"{t}:{us:6.1f}" for t, _, us in res) +
          f" | best {best[0]} {fl/best[2]/1e6:6.1f} TF/s {by/best[2]/1e3:6.0f} GB/s", flush=True)
