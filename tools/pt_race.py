#!/usr/bin/env python3
"""Race screen of conv_pt_kernel at the judged shapes: every conv11 entry of the shipped table (or YH_PT_KEYS=<substring>), REPS launches
each into a NaN-filled output, compared with fp32 torch; prints the launches with elements out of tolerance.
usage: pt_race.py [reps]   (YH_LIBRARY selects a timing / debug build)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import YH_CONV_DGRAD, check, lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = lib()
t = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yoloseries_amd", "tune_defaults.json")))
keys = [k for k in sorted(t) if k.startswith("conv11:") and k.split(":")[1] in ("fwd", "dgrad") and os.environ.get("YH_PT_KEYS", "") in k]
bad_total = 0
for ki, key in enumerate(keys):
    f = [int(x) for x in key.split(":")[-1].split(",")]
    (mode, B, Ho, Wo, Hi, Wi, k, stride, pad, N, nseg, C0, ld0s, ups0, C1, ups1, ldo, nsplit, accumulate, stats, res, act, bias, scale, bnr, acc_rows) = f
    tile_k, grid_cap, algo = t[key]
    if algo != 13 or bnr or stats or accumulate:
        continue
    g = torch.Generator(device=dev).manual_seed(100 + ki)
    segC, segups = [C0, C1][:nseg], [ups0, ups1][:nseg]
    segs, xs = [], []
    for si in range(nseg):
        h, w_ = Hi >> segups[si], Wi >> segups[si]
        ld = ld0s if si == 0 else segC[si]
        buf = torch.randn(B, h, w_, ld, generator=g, device=dev).to(torch.bfloat16)
        segs.append(hipk.Slice(buf, 0, segC[si], segups[si])); xs.append(buf)
    Ctot = sum(segC)
    if mode == YH_CONV_DGRAD:
        Nk, Cin = C0, N
        w = (torch.randn(Nk, Cin, k, k, device=dev, generator=g) / (Nk * k * k) ** 0.5).to(torch.bfloat16).float()
        wp = hipk.pack_weight_dgrad(w)
        ref = torch.nn.grad.conv2d_input((B, N, Ho, Wo), w, xs[0][..., :C0].float().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    else:
        w = (torch.randn(N, Ctot, k, k, device=dev, generator=g) / (Ctot * k * k) ** 0.5).to(torch.bfloat16).float()
        wp = hipk.pack_weight_fwd(w)
        parts = []
        for si in range(nseg):
            x = xs[si][..., :segC[si]].float().permute(0, 3, 1, 2)
            parts.append(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest") if segups[si] else x)
        ref = torch.nn.functional.conv2d(torch.cat(parts, 1), w, None, stride=stride, padding=pad).permute(0, 2, 3, 1)
    ref = ref.contiguous()
    nbad = []
    for r in range(reps):
        out = torch.full((B, Ho, Wo, ldo), float("nan"), dtype=torch.bfloat16, device=dev)
        d = hipk.conv_desc(segs, mode, B, Ho, Wo, Hi, Wi, k, stride, pad, wp, N, hipk.Slice(out, 0, N))
        d.tile_k, d.grid_cap, d.algo = tile_k, grid_cap, algo
        # something else between the launches, as in a step: a fp32 elementwise pass and a matmul leave other bytes in LDS / caches
        junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 512, device=dev)
        check(L.yh_conv_igemm(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), key)
        err = (out[..., :N].float() - ref).abs()
        bad = ~(err <= 4e-2 + 1e-2 * ref.abs())
        n = int(bad.sum())
        if n:
            b2 = bad.reshape(-1, N)
            rows = torch.nonzero(b2.any(1)).flatten()
            cols = torch.nonzero(b2.any(0)).flatten()
            vals = out[..., :N].reshape(-1, N)[rows[0]][cols[:8]].float().tolist()
            nbad.append((r, n, rows[:6].tolist(), int(rows.numel()), cols.tolist()[:20], [f"{v:.3g}" for v in vals]))
    bad_total += len(nbad)
    name = C.create_string_buffer(96); L.yh_conv_kernel_name(C.byref(d), name, 96)
    print(f"{name.value.decode():28s} {key.split(':')[1]:5s} {Ho}x{Wo} C{Ctot} N{N}: {len(nbad)} of {reps} launches wrong", nbad[:3], flush=True)
print("launches with wrong elements:", bad_total)
