#!/usr/bin/env python3
"""Weight-gradient forms on the K-heavy layer shapes of YOLOv5s / v5l (B=64, 640x640): the im2col tilings of conv_wgrad_kernel
(tile_k 0 / 32 / 35 / 128, best split count) against conv_wgs_kernel (tile_k 129: wave-private tiles + stream-K).
Interleaved rounds in ONE process (median of the rounds).   usage: bench_wgs.py [rounds] [iters]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
B = int(os.environ.get("WG_B", "64"))
#        name              H    Cin  Cout k  s
shapes = [("s3_b_3x3",      40,  128, 128, 3, 1), ("s4_b_3x3",      20,  256, 256, 3, 1),
          ("s3_conv",       80,  128, 256, 3, 2), ("s4_conv",       40,  256, 512, 3, 2),
          ("s3_cba12",      40,  256, 256, 1, 1), ("s4_cba3",       20,  512, 512, 1, 1), ("spp_cba2", 20, 1024, 512, 1, 1),
          ("s3_b_1x1",      40,  128, 128, 1, 1),
          ("s2_b_3x3",      80,  64,  64,  3, 1), ("s2_conv",       160, 64,  128, 3, 2), ("s1_conv", 320, 32, 64, 3, 2),
          ("l1_b_3x3",      160, 64,  64,  3, 1), ("l1_conv",       320, 64,  128, 3, 2),
          ("l2_b_3x3",      80,  128, 128, 3, 1), ("l3_b_3x3",      40,  256, 256, 3, 1), ("l4_b_3x3", 20, 512, 512, 3, 1),
          ("l4_conv",       40,  512, 1024, 3, 2), ("l3_cba3",      40,  512, 512, 1, 1), ("l4_cba3", 20, 1024, 1024, 1, 1)]
if os.environ.get("WG_ONLY"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["WG_ONLY"].split(",")]
L = lib()


def timed(d):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hipk.wgrad_launch(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for name, H, Cin, Cout, k, s in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    M = B * Ho * Ho
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    gy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(Cout, k * k * Cin, device=dev)
    fl = 2.0 * M * Cout * Cin * k * k
    cands = []
    for tk in (0, 32, 35, 128):
        if tk == 128 and not (128 <= k * k * Cin <= 384):
            continue
        if tk in (32, 35) and not L.yh_conv_wgrad_kernel_name(Cout, k * k * Cin).decode().startswith("conv_wgrad_kernel<4, 2, 1, 2, 64"):
            continue
        nt = L.yh_conv_wgrad_tiles2(Cout, k * k * Cin, tk)
        for tot in (256, 512, 768, 1024):
            sp = max(1, min((M + 255) // 256, (tot + nt - 1) // nt))
            cands.append((tk, sp))
    d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, Ho, Ho, H, H, k, s, p, dw, 1)
    d.tile_k = 129
    T = L.yh_conv_wgrad_wave_tiles(C.byref(d))
    if T > 0:
        gs = sorted({256} | ({T * (256 // T)} if T <= 256 else set()) | {int(g) for g in os.environ.get("WG_G", "").split(",") if g})
        cands += [(129, g) for g in gs]
    cands = sorted(set(cands))
    res = {c: [] for c in cands}
    for c in cands:                          # warm-up
        d.tile_k, d.splits = c
        hipk.wgrad_launch(d)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for c in cands:
            d.tile_k, d.splits = c
            res[c].append(timed(d))
    med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
    old = min((c for c in cands if c[0] != 129), key=lambda c: med[c])
    line = f"{name:10s} M={M:7d} N={Cout:4d} K={k*k*Cin:5d} T={T:3d} | old best tk={old[0]:3d} sp={old[1]:3d} {med[old]:7.1f} us {fl/med[old]/1e6:6.1f} TF/s"
    for c in cands:
        if c[0] == 129:
            line += f" | wgs G={c[1]:3d} {med[c]:7.1f} us {fl/med[c]/1e6:6.1f} TF/s"
    print(line, flush=True)
