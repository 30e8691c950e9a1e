#!/usr/bin/env python3
"""Per-phase cycle breakdown of the conv kernel's main loop (block 0, its four waves) from the timing build
(`make -C yoloseries_amd/csrc stamps`; run with YH_LIBRARY=yoloseries_amd/libyolohip_stamps.so).
usage: conv_stamps.py B H W Cin Cout k s [fwd|dgrad]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import lib

B, H, W, Cin, Cout, k, s = (int(v) for v in sys.argv[1:8])
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
p = k // 2
dev = torch.device("cuda:0")
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
gy = torch.randn(B, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
if mode == "fwd":
    out = torch.zeros(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, k, s, p, wp, Cout, hipk.full(out))
    stats = torch.zeros(hipk.conv_stat_blocks(d), 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
else:
    gx = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
    wd = hipk.pack_weight_dgrad(w)
    d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, k, s, p, wd, Cin, hipk.full(gx))
for _ in range(3):
    hipk.conv_launch(d)
torch.cuda.synchronize()
L = lib()
buf = (ctypes.c_longlong * 32)()

fn = ctypes.CDLL(os.environ["YH_LIBRARY"]).yh_debug_read_stamps
fn.restype = ctypes.c_int
assert fn(buf) == 0
names = ["load issue", "frag+mfma", "vmwait+store", "barrier", "prologue", "epilogue", "k-steps", "total"]
print(f"{mode} B{B} {H}x{W} {Cin}->{Cout} k{k}s{s}  (cycles of s_memtime, block 0)")
for wv in range(4):
    v = [buf[wv * 8 + i] for i in range(8)]
    n = max(v[6], 1)
    print(f" wave {wv}: total {v[7]:9d} | per k-step: " + "  ".join(f"{names[i]} {v[i] / n:7.1f}" for i in range(4)) +
          f" | k-steps {v[6]}  prologue/tile-sum {v[4]}  epilogue-sum {v[5]}")
