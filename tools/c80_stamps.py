#!/usr/bin/env python3
"""Where conv_c80_kernel's time goes: shader-clock stamps of every wave's third tile (yh_c80_set_stamps), YOLOv5x stage-1
downsampling layer.  Columns are medians over all waves, in shader cycles relative to the tile's first stamp."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
H, Ho = 640, 320
x = torch.randn(B, H, H, 80, device=dev).to(torch.bfloat16)
wp = hipk.pack_weight_fwd((torch.randn(160, 80, 3, 3, device=dev) / 27).to(torch.bfloat16).float())
out = torch.zeros(B, Ho, Ho, 160, dtype=torch.bfloat16, device=dev)
scale, shift = torch.rand(160, device=dev) + 0.5, torch.randn(160, device=dev)
d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Ho, H, H, 3, 2, 1, wp, 160, hipk.full(out), scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
d.algo = 12
hipk.conv_launch(d); torch.cuda.synchronize()
st = torch.zeros(256 * 4 * 16, dtype=torch.int64, device=dev)
L = lib().lib if hasattr(lib(), "lib") else lib()
L.yh_c80_set_stamps.argtypes = [C.c_void_p]; L.yh_c80_set_stamps.restype = None
L.yh_c80_set_stamps(st.data_ptr())
hipk.conv_launch(d); torch.cuda.synchronize()
L.yh_c80_set_stamps(None)
s = st.view(256, 4, 16).cpu()
names = ["tile top", "s0 own DMA landed", "s0 barrier passed", "s1 top (stage 0 done)", "s1 own DMA landed", "s1 barrier passed", "s8 barrier passed",
         "stage 8 done", "epilogue barrier passed", "epilogue done"]
rel = (s - s[:, :, 0:1]).float()
for i, n in enumerate(names):
    v = rel[:, :, i].flatten()
    print(f"  {n:28s} median {v.median().item():9.0f}   min {v.min().item():9.0f}   max {v.max().item():9.0f}")
