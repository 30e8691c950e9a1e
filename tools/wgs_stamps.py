#!/usr/bin/env python3
"""Phase times inside conv_wgs_kernel (tile_k 129) from its shader-clock stamps (yh_wgs_set_stamps): per wave of every workgroup,
first segment: start -> first stage landed -> main loop done -> all waves at the barrier -> LDS combine done -> atomics issued ->
atomics drained.   usage: wgs_stamps.py name H Cin Cout k s [G]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

dev = torch.device("cuda:0")
B = int(os.environ.get("WG_B", "64"))
name, H, Cin, Cout, k, s = sys.argv[1], *[int(v) for v in sys.argv[2:7]]
L = lib()
L.yh_wgs_set_stamps.argtypes = [C.c_void_p]
L.yh_wgs_set_stamps.restype = None
p = k // 2
Ho = (H + 2 * p - k) // s + 1
M = B * Ho * Ho
x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
gy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
dw = torch.zeros(Cout, k * k * Cin, device=dev)
d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, Ho, Ho, H, H, k, s, p, dw, 1)
d.tile_k = 129
T = L.yh_conv_wgrad_wave_tiles(C.byref(d))
G = int(sys.argv[7]) if len(sys.argv) > 7 else (T * (256 // T) if T <= 256 else 256)
d.splits = G
for _ in range(3):
    hipk.wgrad_launch(d)
torch.cuda.synchronize()
st = torch.zeros(G * 4 * 8, dtype=torch.int64, device=dev)
L.yh_wgs_set_stamps(C.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
hipk.wgrad_launch(d)
e1.record()
torch.cuda.synchronize()
L.yh_wgs_set_stamps(None)
t = st.cpu().reshape(G, 4, 8).double()
ok = t[..., 0] > 0
names = ["first stage landed", "main loop", "wait at barrier", "LDS combine", "atomics issued", "atomics drained"]
print(f"{name}: M={M} N={Cout} K={k*k*Cin} T={T} G={G}  kernel {e0.elapsed_time(e1)*1000:.1f} us; steps per wave ~{M/16*T/G/4:.0f}")
t0 = t[..., 0][ok].min()
for i, nm in enumerate(names):
    dlt = (t[..., i + 1] - t[..., i])[ok]
    print(f"  {nm:20s} median {dlt.median().item():9.0f}  p10 {dlt.quantile(0.1).item():9.0f}  p90 {dlt.quantile(0.9).item():9.0f}  max {dlt.max().item():9.0f} ticks")
tot = (t[..., 6] - t[..., 0])[ok]
print(f"  total per wave median {tot.median().item():.0f}, max {tot.max().item():.0f};  last end - first start {(t[..., 6][ok].max() - t0).item():.0f} ticks;"
      f" start spread {(t[..., 0][ok].max() - t0).item():.0f}")
