#!/usr/bin/env python3
"""Weight gradients of one C3 block of YOLOv5s / YOLOv5l (B = 64, 640 x 640) as separate conv_wgs_kernel launches (the engine's
default: 192 workgroups under 60 GFLOP, else 256) against ONE yh_conv_wgrad_group launch on 192 / 224 / 256 workgroups.
Interleaved rounds in one process, median.   usage: bench_wgs_group.py [rounds] [iters]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
B = int(os.environ.get("WG_B", "64"))
L = lib()
#  (H, Cin, Cout, k, s) per launch (a concat input = one launch per segment)
groups = {
    "v5s head_stage4 block (20^2)": [(20, 256, 512, 1, 1)] * 2 + [(20, 256, 256, 3, 1), (20, 256, 256, 1, 1)] + [(20, 256, 512, 1, 1)] * 2,
    "v5s head_stage3 block (40^2)": [(40, 128, 256, 1, 1)] * 2 + [(40, 128, 128, 3, 1), (40, 128, 128, 1, 1)] + [(40, 128, 256, 1, 1)] * 2,
    "v5s backbone_stage3 half (40^2)": [(40, 128, 256, 1, 1)] * 2 + [(40, 128, 128, 3, 1), (40, 128, 128, 1, 1)] * 3,
    "v5s stage2 block (80^2)": [(80, 64, 128, 1, 1)] * 2 + [(80, 64, 64, 3, 1), (80, 64, 64, 1, 1)] + [(80, 128, 128, 1, 1)],
    "v5s detect heads": [(80, 128, 255, 1, 1), (40, 256, 255, 1, 1), (20, 512, 255, 1, 1)],
    "v5l stage4 block (20^2)": [(20, 512, 1024, 1, 1)] * 2 + [(20, 512, 512, 3, 1), (20, 512, 512, 1, 1)] + [(20, 512, 1024, 1, 1)] * 2,
    "v5l stage3 part (40^2)": [(40, 256, 512, 1, 1)] * 2 + [(40, 256, 256, 3, 1), (40, 256, 256, 1, 1)] * 3,
}
if os.environ.get("WG_ONLY"):
    groups = {k: v for k, v in groups.items() if any(w in k for w in os.environ["WG_ONLY"].split(","))}


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for gname, specs in groups.items():
    descs, keep, flops, nbytes = [], [], 0.0, 0.0
    for (H, Cin, Cout, k, s) in specs:
        p = k // 2
        Ho = (H + 2 * p - k) // s + 1
        M = B * Ho * Ho
        x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
        ldg = (Cout + 7) // 8 * 8
        gy = torch.randn(B, Ho, Ho, ldg, device=dev).to(torch.bfloat16)
        dw = torch.zeros(Cout, k * k * Cin, device=dev)
        d = hipk.wgrad_desc(hipk.Slice(gy, 0, ldg), Cout, hipk.full(x), 0, Cin, B, Ho, Ho, H, H, k, s, p, dw, 1)
        d.tile_k = 129
        assert L.yh_conv_wgrad_wave_tiles(C.byref(d)) > 0
        fl = 2.0 * M * Cout * Cin * k * k
        d.splits = 256 if fl >= 60e9 else 192
        flops += fl
        nbytes += 2.0 * M * (ldg + Cin)
        descs.append(d)
        keep.append((x, gy, dw))
    arr = hipk.wgrad_group_array(descs)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def separate():
        for d in descs:
            L.yh_conv_wgrad(C.byref(d), st)

    forms = {"separate": separate}
    for wg in (192, 224, 256):
        forms[f"group{wg}"] = (lambda wg=wg: L.yh_conv_wgrad_group(arr, len(descs), wg, st))
    res = {k: [] for k in forms}
    for f in forms.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in forms.items():
            res[k].append(timed(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{gname:34s} {len(descs)} launches {flops/1e9:6.1f} GFLOP {nbytes/1e6:6.0f} MB | " +
          " | ".join(f"{k} {v:7.1f} us {flops/v/1e6:5.0f} TF/s {nbytes/v/1e3:5.0f} GB/s" for k, v in med.items()), flush=True)
