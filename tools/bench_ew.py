#!/usr/bin/env python3
"""HBM-rate check of the BN/SiLU passes on layer-sized tensors against plain torch streaming ops.
usage: bench_ew.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, M, C in (("focus", 64 * 320 * 320, 32), ("s1", 64 * 160 * 160, 64), ("s2", 64 * 80 * 80, 128), ("s3", 64 * 40 * 40, 256), ("s4", 64 * 20 * 20, 512)):
    z = torch.randn(M, C, device=dev).to(torch.bfloat16)
    g = torch.randn(M, C, device=dev).to(torch.bfloat16)
    o = torch.empty_like(z)
    ws = torch.randn(4 * C, device=dev).abs() + 0.5
    gamma = torch.ones(C, device=dev)
    coef = torch.zeros(2 * C, device=dev)
    nblk = hipk.ew_blocks(M)
    part = torch.zeros(nblk, 2, C, device=dev)
    nb = M * C * 2
    r = {}
    r["torch copy (r+w)"] = (timeit(lambda: o.copy_(z)), 2 * nb)
    r["torch sum fp32acc (r)"] = (timeit(lambda: torch.sum(z, dtype=torch.float32)), nb)
    r["torch add (2r+w)"] = (timeit(lambda: torch.add(z, g, out=o)), 3 * nb)
    r["bn_silu_apply (r+w)"] = (timeit(lambda: hipk.bn_silu_apply(hipk.full(z), ws, M, hipk.full(o))), 2 * nb)
    r["bwd_reduce (2r)"] = (timeit(lambda: hipk.bn_silu_bwd_reduce(hipk.full(g), hipk.full(z), ws, M, part)), 2 * nb)
    r["bwd_apply (2r+w)"] = (timeit(lambda: hipk.bn_silu_bwd_apply(hipk.full(g), hipk.full(z), ws, gamma, coef, M, hipk.full(o))), 3 * nb)
    print(f"{name:6s} M={M:8d} C={C:4d} {nb/1e6:7.1f} MB/tensor | " + " | ".join(f"{k}: {ms*1000:6.1f}us {b/ms/1e9:5.2f}TB/s" for k, (ms, b) in r.items()), flush=True)
