#!/usr/bin/env python3
"""Micro-benchmark of one conv shape through the C ABI (used with rocprofv3 --pmc).
usage: bench_conv.py B H W Cin Cout k s [mode: fwd|dgrad|wgrad] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk

B, H, W, Cin, Cout, k, s = (int(v) for v in sys.argv[1:8])
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
p = k // 2
dev = torch.device("cuda:0")
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5)
gy = torch.randn(B, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
if mode == "fwd":
    out = torch.zeros(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, k, s, p, wp, Cout, hipk.full(out))
    nblk = hipk.conv_stat_blocks(d)
    stats = torch.zeros(nblk, 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
    run = lambda: hipk.conv_launch(d)
elif mode == "dgrad":
    gx = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
    wd = hipk.pack_weight_dgrad(w)
    d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, k, s, p, wd, Cin, hipk.full(gx))
    run = lambda: hipk.conv_launch(d)
else:
    dw = torch.zeros(Cout, k * k * Cin, device=dev)
    M = B * Ho * Wo
    d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, Ho, Wo, H, W, k, s, p, dw, max(1, min((M + 255) // 256, 256)))
    run = lambda: hipk.wgrad_launch(d)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
by = 2.0 * (B * H * W * Cin + B * Ho * Wo * Cout)
print(f"{mode} B{B} {H}x{W} {Cin}->{Cout} k{k}s{s}: {ms*1000:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s(act)")
