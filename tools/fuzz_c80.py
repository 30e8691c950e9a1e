#!/usr/bin/env python3
"""Random-shape check of conv_c80_kernel (yh_conv_desc.algo 12) against torch on the GPU: random batch / map sizes (odd ones, maps
smaller than a tile, tiles that start inside an output row) / stride 1 or 2 / input and output as channel slices of wider
NaN- / constant-filled buffers / persistent grids of any size.   usage: fuzz_c80.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from yoloseries_amd import hipk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for case in range(n):
    B, H, W = int(rng.randint(1, 6)), int(rng.randint(3, 70)), int(rng.randint(3, 90))
    s = int(rng.choice([1, 2]))
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    padl, padr = 8 * int(rng.randint(0, 3)), 8 * int(rng.randint(0, 3))
    xbuf = torch.full((B, H, W, 80 + padl + padr), float("nan"), dtype=torch.bfloat16, device=dev)
    x = torch.randn(B, H, W, 80, device=dev).to(torch.bfloat16)
    xbuf[..., padl:padl + 80] = x
    w = (torch.randn(160, 80, 3, 3, device=dev) / 27).to(torch.bfloat16).float()
    wp = hipk.pack_weight_fwd(w)
    ol, orr = 8 * int(rng.randint(0, 3)), 8 * int(rng.randint(0, 3))
    obuf = torch.full((B, Ho, Wo, 160 + ol + orr), 5.0, dtype=torch.bfloat16, device=dev)
    scale, shift, bias = torch.rand(160, device=dev) + 0.5, torch.randn(160, device=dev), torch.randn(160, device=dev)
    act = bool(rng.randint(0, 2))
    d = hipk.conv_desc([hipk.Slice(xbuf, padl, 80)], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, 3, s, 1, wp, 160, hipk.Slice(obuf, ol, 160),
                       bias=bias, scale=scale, shift=shift, act=hipk.YH_ACT_SILU if act else hipk.YH_ACT_NONE)
    d.algo = 12
    d.grid_cap = int(rng.choice([0, 1, 2, 3, 7, 64]))
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    ref = (F.conv2d(x.float().permute(0, 3, 1, 2), w, None, stride=s, padding=1).permute(0, 2, 3, 1) + bias) * scale + shift
    if act:
        ref = F.silu(ref)
    got = obuf[..., ol:ol + 160].float()
    err = (got - ref).abs().max().item()
    tol = 2e-2 * ref.abs().max().item() + 1e-2
    clean = bool((obuf[..., :ol] == 5.0).all() and (obuf[..., ol + 160:] == 5.0).all()) and bool(torch.isfinite(got).all())
    ok = err <= tol and clean
    bad += not ok
    if not ok or case % 20 == 0:
        print(f"case {case}: B{B} {H}x{W} s{s} cap{d.grid_cap} act{int(act)}: max err {err:.4f} (tol {tol:.4f}) {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
