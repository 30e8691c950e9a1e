#!/usr/bin/env python3
"""Launches of the last full train step that are NOT kernels of libyolohip.so (torch fills / copies / elementwise kernels, runtime
copy kernels), from a rocprofv3 --kernel-trace CSV of bench.py: name, count, total time, and the library kernel each one follows —
enough to find the line of host code that issues it.
usage: foreign_launches.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if 'input_s2d' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
ours = lambda k: '(anonymous namespace)::' in k        # noqa: E731  every kernel of the library lives in an anonymous namespace
short = lambda k: k.replace('(anonymous namespace)::', '').replace('void ', '')[:90]   # noqa: E731
foreign = [(i, r) for i, r in enumerate(step) if not ours(r['Kernel_Name'])]
tot = sum(r['e'] - r['s'] for _, r in foreign)
print(f"last full step: {len(step)} launches, {len(foreign)} foreign, {tot / 1e3:.1f} us of foreign kernel time")
agg = collections.OrderedDict()
for i, r in foreign:
    prev = next((short(step[j]['Kernel_Name']) for j in range(i - 1, -1, -1) if ours(step[j]['Kernel_Name'])), '-')
    key = (short(r['Kernel_Name']), prev[:48])
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += r['e'] - r['s']
for (name, prev), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:4d} x {ns / 1e3 / n:7.1f} us  {name}\n           after {prev}")
