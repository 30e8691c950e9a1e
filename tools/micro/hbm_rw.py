#!/usr/bin/env python3
"""HBM bandwidth by direction on this GPU, with torch's own kernels on 4 GiB tensors (far beyond the 256 MB Infinity Cache):
write only (fill_), read only (sum), one byte written per byte read (copy_), two read per one written (add into a third)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 30                                  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
c = torch.empty(n, dtype=torch.float32, device=dev)
a.fill_(1.0); b.fill_(2.0); c.fill_(0.0)
torch.cuda.synchronize()


def timed(fn, nbytes, label, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:52s} {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s")


timed(lambda: a.fill_(3.0), 4 * n, "write only            (fill_, 4 GiB)")
timed(lambda: a.sum(), 4 * n, "read only             (sum, 4 GiB)")
timed(lambda: b.copy_(a), 8 * n, "1 read : 1 write      (copy_, 4 + 4 GiB)")
timed(lambda: torch.add(a, b, out=c), 12 * n, "2 read : 1 write      (add out=, 8 + 4 GiB)")
