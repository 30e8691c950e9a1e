#!/usr/bin/env python3
"""Random-shape check of conv_wgs_kernel (yh_wgrad_desc.tile_k 129) against torch on the GPU:
random maps / strides / channel counts in multiples of 32 / channel slices of wider NaN-filled buffers / workgroup counts
(stream-K segment boundaries anywhere) / one or two segments.   usage: fuzz_wgs.py [cases] [seed]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
L = lib()
bad = 0


def nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


for case in range(n):
    k = int(rng.choice([1, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    p = k // 2
    B = int(rng.randint(1, 5))
    Ho, Wo = int(rng.randint(2, 30)), int(rng.randint(2, 30))
    if (B * Ho * Wo) % 32:                        # eligibility: pixel count a multiple of 32 -> pad the batch to fit
        B = 32 // np.gcd(32, Ho * Wo) * int(rng.randint(1, 3))
    H, W = (Ho * s if s == 2 else Ho), (Wo * s if s == 2 else Wo)
    if B * Ho * Wo > 60000 or Ho * Wo < 16:
        continue
    nseg = int(rng.choice([1, 1, 2]))
    Cs = [32 * int(rng.randint(1, 9)) for _ in range(nseg)]
    ups = [int(rng.randint(0, 2)) if (H % 2 == 0 and W % 2 == 0 and s == 1) else 0 for _ in range(nseg)]
    Cout = 8 * int(rng.randint(8, 40))
    Ctot = sum(Cs)
    ldg = Cout + 8 * int(rng.randint(0, 3))
    gyb = torch.full((B, Ho, Wo, ldg), float("nan"), dtype=torch.bfloat16, device=dev)
    gyb[..., :(Cout + 7) // 8 * 8] = 0
    gyb[..., :Cout] = torch.randn(B, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
    segs, xs = [], []
    for C_, u in zip(Cs, ups):
        lead = 8 * int(rng.randint(0, 3))
        xb = torch.full((B, H >> u, W >> u, lead + C_ + 8), float("nan"), dtype=torch.bfloat16, device=dev)
        xb[..., lead:lead + C_] = torch.randn(B, H >> u, W >> u, C_, device=dev).to(torch.bfloat16)
        segs.append(hipk.Slice(xb, lead, C_, ups=u))
        x = nchw(xb[..., lead:lead + C_])
        xs.append(F.interpolate(x, scale_factor=2, mode="nearest") if u else x)
    dw = torch.full((Cout, k * k * Ctot), 0.25, device=dev)
    G = int(rng.choice([1, 3, 8, 61, 192, 256]))
    d = hipk.wgrad_desc(hipk.Slice(gyb, 0, ldg), Cout, segs[0], 0, Ctot, B, Ho, Wo, H, W, k, s, p, dw, G)
    d.tile_k = 129
    merged = False
    T = L.yh_conv_wgrad_wave_tiles(C.byref(d))
    if T <= 0:
        print(f"case {case}: not eligible (skipped)")
        continue
    hipk.wgrad_launch(d)
    if nseg == 2 and not merged:
        d2 = hipk.wgrad_desc(hipk.Slice(gyb, 0, ldg), Cout, segs[1], Cs[0], Ctot, B, Ho, Wo, H, W, k, s, p, dw, G)
        d2.tile_k = 129
        assert L.yh_conv_wgrad_wave_tiles(C.byref(d2)) > 0
        hipk.wgrad_launch(d2)
    torch.cuda.synchronize()
    w = torch.zeros(Cout, Ctot, k, k, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(torch.cat(xs, 1), w, stride=s, padding=p), w, nchw(gyb[..., :Cout]))
    ref = ref.permute(0, 2, 3, 1).reshape(Cout, -1)
    got = dw - 0.25
    err = (got - ref).abs().max().item()
    tol = 3e-3 * max(1.0, ref.abs().max().item())
    ok = err <= tol and not torch.isnan(dw).any()
    bad += 0 if ok else 1
    print(f"wgs case {case}: B{B} {Ho}x{Wo} k{k} s{s} C{Cs} ups{ups} N{Cout} G{G} T{T}{' merged' if merged else ''}: max err {err:.5f} (tol {tol:.5f}) {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
