#!/usr/bin/env python3
"""YOLOv5x stage-1 downsampling layer (ConvBnAct(80, 160, 3, 2) at 640 x 640, inference epilogue): conv_c80_kernel (algo 12) against
every other eligible kernel family, isolated.   usage: bench_c80.py [batch] [iters]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
H = int(os.environ.get("C80_H", "640"))
stride = 2
Ho = (H + 2 - 3) // stride + 1
x = torch.randn(B, H, H, 80, device=dev).to(torch.bfloat16)
w = (torch.randn(160, 80, 3, 3, device=dev) / 27).to(torch.bfloat16).float()
wp = hipk.pack_weight_fwd(w)
out = torch.zeros(B, Ho, Ho, 160, dtype=torch.bfloat16, device=dev)
scale = torch.rand(160, device=dev) + 0.5
shift = torch.randn(160, device=dev)
d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Ho, H, H, 3, stride, 1, wp, 160, hipk.full(out), scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
fl = 2.0 * B * Ho * Ho * 160 * 720
buf = C.create_string_buffer(96)
ref = None
for algo, tk in ((1, 0), (2, 0), (2, 32), (3, 0), (3, 32), (4, 0), (4, 32), (12, 0)):
    d.algo, d.tile_k = algo, tk
    lib().yh_conv_kernel_name(C.byref(d), buf, 96)
    kn = buf.value.decode()
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.float().clone()
    err = (out.float() - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hipk.conv_launch(d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"algo {algo:2d} tile_k {tk:2d}  {kn:52s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s   max |diff to algo 1| {err:.4f}", flush=True)
