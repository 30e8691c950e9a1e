#!/usr/bin/env python3
"""Phase times inside conv_pt_kernel (yh_conv_desc.algo 13) from its shader-clock stamps (yh_pt_set_stamps): per wave of every
workgroup, the group's 5th round: wait for the tile's transfers -> barrier 1 -> fragment reads + MFMA loop -> staging written -> barrier 2 ->
next transfers / operand loads issued (+ wait for this tile's operands) -> epilogue math + stores issued.   usage: pt_stamps.py H C0 C1 N [fwd|dgrad3]   (batch 64; YH_PT_ABL applies)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

dev = torch.device("cuda:0")
B = int(os.environ.get("BP_BATCH", "64"))
H, C0, C1, N = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"
L = lib()
raw = C.CDLL(__import__("yoloseries_amd._lib", fromlist=["LIB_PATH"]).LIB_PATH)
raw.yh_pt_set_stamps.argtypes = [C.c_void_p]
raw.yh_pt_set_stamps.restype = None
segs = [hipk.full(torch.randn(B, H, H, C0, device=dev).to(torch.bfloat16))]
if C1:
    segs.append(hipk.full(torch.randn(B, H, H, C1, device=dev).to(torch.bfloat16)))
w = torch.randn(N, C0 + C1, 1, 1, device=dev) / (C0 + C1) ** 0.5
wp = hipk.pack_weight_fwd(w)
out = torch.zeros(B, H, H, N, dtype=torch.bfloat16, device=dev)
d = hipk.conv_desc(segs, hipk.YH_CONV_FWD if mode == "fwd" else hipk.YH_CONV_DGRAD, B, H, H, H, H, 1, 1, 0, wp, N, hipk.full(out))
if mode == "fwd":
    stats = torch.zeros(8192, 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
else:
    z = torch.randn(B, H, H, N, device=dev).to(torch.bfloat16)
    ws = torch.cat([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)])
    slab = torch.zeros(16384, 2, N, device=dev)
    d.bnr_z, d.bnr_ldz, d.bnr_C, d.bnr_ws, d.bnr_part = z.data_ptr(), N, N, ws.data_ptr(), slab.data_ptr()
d.algo = 13
for _ in range(3):
    hipk.conv_launch(d)
torch.cuda.synchronize()
G = 4096
st = torch.zeros(G * 8 * 8, dtype=torch.int64, device=dev)
raw.yh_pt_set_stamps(C.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
hipk.conv_launch(d)
e1.record()
torch.cuda.synchronize()
raw.yh_pt_set_stamps(None)
t = st.cpu().reshape(G, 8, 8).double()
ok = t[..., 0] > 0
names = ["wait for transfers", "barrier 1", "fragments + MFMA loop", "staging write", "barrier 2", "issue transfers / loads, wait", "epilogue math + stores"]
print(f"{mode} {H}^2 {C0}+{C1} -> {N}: kernel {e0.elapsed_time(e1) * 1000:.1f} us, {int(ok.sum())} waves stamped (100 MHz ticks x {'?'}: s_memtime)")
for i, nm in enumerate(names):
    dlt = (t[..., i + 1] - t[..., i])[ok]
    print(f"  {nm:26s} median {dlt.median().item():8.0f}  p10 {dlt.quantile(0.1).item():8.0f}  p90 {dlt.quantile(0.9).item():8.0f}")
tot = (t[..., 7] - t[..., 0])[ok]
print(f"  tile total median {tot.median().item():.0f} p90 {tot.quantile(0.9).item():.0f}")
