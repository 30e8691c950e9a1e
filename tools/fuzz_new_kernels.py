#!/usr/bin/env python3
"""Random-shape check of the round-3 inference kernels (conv_h80_kernel, conv_pw_kernel) against torch on the GPU.
usage: fuzz_new_kernels.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from yoloseries_amd import hipk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for case in range(n):
    kind = "h80" if case % 2 == 0 else "pw"
    B, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 70)), int(rng.randint(1, 70))
    if kind == "h80":
        Cin, k, p, algo = 80, 3, 1, 9
    else:
        Cin, k, p, algo = int(rng.choice([80, 160, 320])), 1, 0, 10
    Cout = 80 * int(rng.randint(1, 4))
    lead, trail = 8 * int(rng.randint(0, 3)), 8 * int(rng.randint(0, 3))
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    xbuf = torch.full((B, H, W, lead + Cin + trail), float("nan"), dtype=torch.bfloat16, device=dev)
    xbuf[..., lead:lead + Cin] = x
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float()
    wp = hipk.pack_weight_fwd(w)
    scale, shift = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    ref = F.silu(F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=p).permute(0, 2, 3, 1) * scale + shift)
    obuf = torch.full((B, H, W, Cout + 16), 3.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([hipk.Slice(xbuf, lead, Cin)], hipk.YH_CONV_FWD, B, H, W, H, W, k, 1, p, wp, Cout, hipk.Slice(obuf, 8, Cout),
                       scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
    d.algo = algo
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    out = obuf[..., 8:8 + Cout].float()
    err = (out - ref).abs().max().item()
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    ok = err <= tol and bool((obuf[..., :8] == 3.0).all()) and bool((obuf[..., 8 + Cout:] == 3.0).all()) and not torch.isnan(out).any()
    if not ok:
        bad += 1
    print(f"{kind} B{B} {H}x{W} {Cin}->{Cout} lead{lead} trail{trail}: max err {err:.4f} (tol {tol:.4f}) {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
