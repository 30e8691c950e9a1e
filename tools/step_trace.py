#!/usr/bin/env python3
"""Summary of one train step from a rocprofv3 --kernel-trace CSV (bench.py, two streams): span, busy time per queue, the gaps of the main
queue, the slow finalize launches and the kernels of the step's last 0.8 ms.  The profiler slows the host down: gaps at the very end of a step
(optimizer / EMA section) are host time that an un-profiled run hides (tools/host_ahead.py).
usage: step_trace.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if 'input_s2d' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = step[0]['s']
short = lambda k: k.replace('(anonymous namespace)::', '').replace('void ', '')[:64]   # noqa: E731
print(f"last full step: {(step[-1]['e'] - t0) / 1e6:.3f} ms, {len(step)} kernels")
qs = collections.Counter(r['Queue_Id'] for r in step)
mainq = qs.most_common(1)[0][0]
for q, n in qs.items():
    print(f"  queue {q}{' (main)' if q == mainq else ' (weight gradients)'}: {n} kernels, busy {sum(r['e'] - r['s'] for r in step if r['Queue_Id'] == q) / 1e6:.3f} ms")
main = [r for r in step if r['Queue_Id'] == mainq]
gaps = [((n['s'] - p['e']) / 1e3, (p['s'] - t0) / 1e6, short(p['Kernel_Name']), short(n['Kernel_Name'])) for p, n in zip(main, main[1:]) if n['s'] - p['e'] > 3000]
print(f"main-queue gaps > 3 us: {len(gaps)}, {sum(g[0] for g in gaps) / 1e3:.3f} ms")
for g in sorted(gaps, reverse=True)[:8]:
    print(f"  {g[0]:7.1f} us at {g[1]:6.2f} ms  {g[2]} -> {g[3]}")
fin = [((r['e'] - r['s']) / 1e3, (r['s'] - t0) / 1e6, short(r['Kernel_Name'])) for r in step if 'finalize' in r['Kernel_Name']]
d = sorted(f[0] for f in fin)
print(f"finalize launches: {len(fin)}, {sum(d) / 1e3:.3f} ms, median {d[len(d) // 2]:.1f} us; over 12 us: {sum(1 for x in d if x > 12)} ({sum(x for x in d if x > 12) / 1e3:.3f} ms)")
for f in sorted(fin, reverse=True)[:6]:
    print(f"  {f[0]:7.1f} us at {f[1]:6.2f} ms  {f[2]}")
print("end of the step:")
for r in step:
    if (step[-1]['e'] - r['s']) / 1e6 < 0.8:
        print(f"  {(r['s'] - t0) / 1e6:7.3f} ms {(r['e'] - r['s']) / 1e3:7.1f} us q{r['Queue_Id']} {short(r['Kernel_Name'])}")
