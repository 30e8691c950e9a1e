#!/usr/bin/env python3
"""Do a weight gradient (side stream) and a BatchNorm+SiLU backward pass (main stream) overlap on the chip?
Times each alone and both together (two streams), for the weight-gradient forms old (best im2col tiling) and new (tile_k 129).
usage: overlap_probe.py   (YH_EW_BPC = blocks per CU of the elementwise pass)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()
B, H, Cin, Cout, k = 64, 40, 256, 256, 3
M = B * H * H
x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
gy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
dw = torch.zeros(Cout, k * k * Cin, device=dev)
d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, H, H, H, H, k, 1, 1, dw, 1)
# elementwise operand: a stage-2-like layer (80 x 80 x 128, B=64)
Mb, Cb = 64 * 80 * 80, 128
ga = torch.randn(Mb, Cb, device=dev).to(torch.bfloat16)
y = torch.randn(Mb, Cb, device=dev).to(torch.bfloat16)
gz = torch.empty_like(ga)
ws = torch.rand(4 * Cb, device=dev) + 0.5
gamma = torch.rand(Cb, device=dev) + 0.5
coef = torch.randn(2 * Cb, device=dev) * 0.01
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def bn(n):
    for _ in range(n):
        hipk.check(L.yh_bn_silu_bwd_apply(ga.data_ptr(), Cb, y.data_ptr(), Cb, ws.data_ptr(), gamma.data_ptr(), coef.data_ptr(), Cb, Mb,
                                          gz.data_ptr(), Cb, None, 0, 0, C.c_void_p(main.cuda_stream)), "bn")


def wg(n):
    for _ in range(n):
        hipk.check(L.yh_conv_wgrad(C.byref(d), C.c_void_p(side.cuda_stream)), "wg")


def timeit(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    side.wait_event(e0)
    fn()
    ev = torch.cuda.Event()
    ev.record(side)
    main.wait_event(ev)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000


N = 10
for tk, sp in ((35, 15), (129, 252)):
    d.tile_k, d.splits = tk, sp
    for _ in range(2):
        tw = timeit(lambda: wg(N)) / N
        tb = timeit(lambda: bn(N)) / N
        both = timeit(lambda: (wg(N), bn(N))) / N
    print(f"tile_k {tk:3d}: wgrad alone {tw:6.1f} us, bn pass alone {tb:6.1f} us ({3*Mb*Cb*2/tb/1e6:.2f} TB/s), both streams {both:6.1f} us (sum {tw+tb:6.1f})", flush=True)
