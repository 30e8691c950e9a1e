#!/usr/bin/env python3
"""Sweep tile_n / grid_cap of yh_conv_igemm over representative YOLOv5s layer shapes (B=64, 640x640).
usage: sweep_conv.py [fwd|dgrad] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
B = 64
shapes = [("focus",         320, 16,  32,  3, 1),
          ("s1_conv",       320, 32,  64,  3, 2),
          ("s1_b_3x3",      160, 32,  32,  3, 1),
          ("s1_b_1x1",      160, 32,  32,  1, 1),
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
          ("det_small",     80,  128, 255, 1, 1)]
for name, H, Cin, Cout, k, s in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    M = B * Ho * Ho
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    gy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
    if mode == "fwd":
        out = torch.zeros(B, Ho, Ho, (Cout + 7) // 8 * 8, dtype=torch.bfloat16, device=dev)
        wp = hipk.pack_weight_fwd(w)
        d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Ho, H, H, k, s, p, wp, Cout, hipk.full(out))
        stats = torch.zeros(4096, 2, wp.shape[0], device=dev)
        if name != "det_small":
            d.stats = stats.data_ptr()
        nout = Cout
    else:
        if name == "focus":
            continue
        gx = torch.zeros(B, H, H, Cin, dtype=torch.bfloat16, device=dev)
        wd = hipk.pack_weight_dgrad(w)
        d = hipk.conv_desc([hipk.full(gy[..., :Cout] if Cout % 8 == 0 else torch.zeros(B, Ho, Ho, (Cout + 7) // 8 * 8, dtype=torch.bfloat16, device=dev))],
                           hipk.YH_CONV_DGRAD, B, H, H, Ho, Ho, k, s, p, wd, Cin, hipk.full(gx))
        nout = Cin
    fl = 2.0 * M * Cout * Cin * k * k
    res = []
    for tn in (32, 64, 128):
        if tn > 32 and tn >= 2 * ((nout + 31) // 32 * 32):
            continue
        for capmul in (1, 2):
            d.tile_n, d.grid_cap = tn, 0
            base = hipk.conv_stat_blocks(d)
            d.grid_cap = base * capmul
            if capmul == 2 and hipk.conv_stat_blocks(d) == base:
                continue
            if hipk.conv_stat_blocks(d) > 4096:
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
            res.append((tn, capmul, e0.elapsed_time(e1) / iters * 1000))
    best = min(res, key=lambda r: r[2])
    print(f"{mode} {name:10s} M={M:8d} Cin={Cin:4d} Cout={Cout:4d} k{k}s{s} | " + " ".join(f"{t}x{c}:{us:6.1f}" for t, c, us in res) +
          f" | best {best[0]}x{best[1]} {fl/best[2]/1e6:6.1f} TF/s", flush=True)
