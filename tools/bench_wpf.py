#!/usr/bin/env python3
"""Forward / data-gradient conv families on the K-heavy stride-1 layer shapes (B = 64 train, YOLOv5s / v5l; and YOLOv5x-like inference
widths that are multiples of 128): best of algo 1..8 against conv_wpf_kernel (algo 11).  Interleaved rounds in one process.
usage: bench_wpf.py [rounds] [iters]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
B = int(os.environ.get("WG_B", "64"))
#        name          H    Cin  Cout k
shapes = [("s3_b_3x3",  40,  128, 128, 3), ("s4_b_3x3",  20,  256, 256, 3),
          ("l2_b_3x3",  80,  128, 128, 3), ("l3_b_3x3",  40,  256, 256, 3), ("l4_b_3x3", 20, 512, 512, 3),
          ("l3_cba3",   40,  512, 512, 1), ("l2_cba3",   80,  256, 256, 1), ("s3_cba3", 40, 256, 256, 1)]
if os.environ.get("WG_ONLY"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["WG_ONLY"].split(",")]
L = lib()


def kname(d):
    buf = C.create_string_buffer(96)
    L.yh_conv_kernel_name(C.byref(d), buf, 96)
    return buf.value.decode()


def timed(d):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hipk.conv_launch(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for name, H, Cin, Cout, k in shapes:
    p = k // 2
    M = B * H * H
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (k * k * Cin) ** 0.5).to(torch.bfloat16).float()
    out = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, H, H, H, k, 1, p, wp, Cout, hipk.full(out))
    fl = 2.0 * M * Cout * Cin * k * k
    res = {}
    cands = []
    stats = torch.zeros(2048, 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
    for algo in (1, 2, 3, 4, 5, 8, 11):
        d.algo = algo
        kn = kname(d)
        if algo in (2, 3, 4) and "conv_v3" not in kn: continue
        if algo == 5 and "conv_halo_kernel" not in kn: continue
        if algo == 8 and "conv_p3" not in kn: continue
        if algo == 11 and "conv_wpf" not in kn: continue
        cands.append((algo, kn))
    for algo, kn in cands:
        d.algo = algo
        hipk.conv_launch(d)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for algo, kn in cands:
            d.algo = algo
            res.setdefault(algo, []).append(timed(d))
    med = {a: sorted(v)[len(v) // 2] for a, v in res.items()}
    old = min((a for a in med if a != 11), key=lambda a: med[a])
    line = f"{name:10s} M={M:7d} C={Cin:4d} N={Cout:4d} k={k} | best old algo {old} {dict(cands)[old][:34]:34s} {med[old]:7.1f} us {fl/med[old]/1e6:6.1f} TF/s"
    if 11 in med:
        line += f" | wpf {med[11]:7.1f} us {fl/med[11]/1e6:6.1f} TF/s"
    print(line, flush=True)
