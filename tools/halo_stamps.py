#!/usr/bin/env python3
"""Where a tile of conv_halo160_kernel (yh_conv_desc.algo 6) spends its cycles, from the cycle sums it records when a stamp buffer is
set (yh_halo_set_stamps): per wave of every workgroup, the workgroup's 2nd tile — waits + barriers of the tap steps, DMA issue,
fragment reads + MFMAs, the activation, the four staging / store phases.  Inference epilogue (folded BN + SiLU), isolated launch.
usage: halo_stamps.py [C=160] [H=160] [batch=128] [N=C]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib, LIB_PATH

Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 160
H = int(sys.argv[2]) if len(sys.argv) > 2 else 160
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
N = int(sys.argv[4]) if len(sys.argv) > 4 else Cin
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
w = torch.randn(N, Cin, 3, 3, device=dev) / (9 * Cin) ** 0.5
wp = hipk.pack_weight_fwd(w)
out = torch.zeros(B, H, H, N, dtype=torch.bfloat16, device=dev)
scale, shift = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
res = torch.randn(B, H, H, N, device=dev).to(torch.bfloat16) if os.environ.get("HS_RES") == "1" else None      # the bottlenecks' shortcut
d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, H, H, H, 3, 1, 1, wp, N, hipk.full(out), scale=scale, shift=shift, act=hipk.YH_ACT_SILU,
                   **({"res": hipk.full(res)} if res is not None else {}))
raw = C.CDLL(LIB_PATH)
if not hasattr(raw, "yh_halo_set_stamps"):
    raw = None
else:
    raw.yh_halo_set_stamps.argtypes = [C.c_void_p]
    raw.yh_halo_set_stamps.restype = None
buf = C.create_string_buffer(96)
fl = 2.0 * B * H * H * N * 9 * Cin
for algo in (6, 5, 3):
    d.algo = algo
    if lib().yh_conv_kernel_name(C.byref(d), buf, 96):
        continue
    for _ in range(2):
        hipk.conv_launch(d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hipk.conv_launch(d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"algo {algo}: {buf.value.decode():44s} {ms:7.3f} ms  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
if raw is None:
    sys.exit(0)
d.algo = 6
G = 2048
st = torch.zeros(G * 8 * 8, dtype=torch.int64, device=dev)
raw.yh_halo_set_stamps(C.c_void_p(st.data_ptr()))
hipk.conv_launch(d)
torch.cuda.synchronize()
raw.yh_halo_set_stamps(None)
t = st.cpu().reshape(G, 8, 8).double()
ok = t[..., 7] > 0
names = ["sum over steps: wait + barrier", "sum over steps: DMA issue", "sum over steps: address + fragments + MFMA", "k loop in all", "activation (all waves)",
         "staging + stores (4 phases)", "tile total"]
nsteps = ((Cin + 63) // 64) * 9
print(f"{int(ok.sum())} waves stamped; {nsteps} tap steps per tile; 40 MFMAs of 16 cycles per full step and wave, two waves per SIMD")
for i, nm in enumerate(names):
    v = t[..., i][ok]
    print(f"  {nm:44s} median {v.median().item():9.0f}  p10 {v.quantile(0.1).item():9.0f}  p90 {v.quantile(0.9).item():9.0f}")
