#!/usr/bin/env python3
"""The loss section of the last full train step of a rocprofv3 kernel trace: every kernel from the first loss kernel (v5_* / yolox_*)
to the last one, with queue, start offset and duration.   usage: loss_section.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
# last full step: from the second-to-last input_s2d / pack kernel
starts = [i for i, r in enumerate(rows) if "input_s2d" in r["Kernel_Name"]]
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
idx = [i for i, r in enumerate(step) if name(r).startswith(("v5_", "yolox_"))]
a, b = max(idx[0] - 3, 0), min(idx[-1] + 4, len(step))
t0 = step[a]["s"]
qs = sorted({r["Queue_Id"] for r in step})
for r in step[a:b]:
    print(f"  q{qs.index(r['Queue_Id'])}  +{(r['s'] - t0) / 1e3:8.1f} us  {(r['e'] - r['s']) / 1e3:7.1f} us  {name(r)[:90]}")
print(f"  section {(step[b - 1]['e'] - t0) / 1e3:.1f} us")
