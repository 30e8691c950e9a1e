#!/usr/bin/env python3
"""How the two queues of a train step share the chip, from a rocprofv3 --kernel-trace CSV: time with only the main queue busy,
only the weight-gradient queue, both, neither; per kernel family on each queue: launches, time, and the share of that time during
which the OTHER queue had a kernel in flight.   usage: step_overlap.py <kernel_trace.csv>"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if 'input_s2d' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0, t1 = step[0]['s'], max(r['e'] for r in step)
qs = collections.Counter(r['Queue_Id'] for r in step)
mainq = qs.most_common(1)[0][0]
ev = []
for r in step:
    q = 0 if r['Queue_Id'] == mainq else 1
    ev.append((r['s'], 1, q)); ev.append((r['e'], -1, q))
ev.sort()
cnt = [0, 0]; last = t0; acc = collections.Counter()
for t, d, q in ev:
    key = ('main' if cnt[0] else '') + ('+side' if cnt[1] else '')
    acc[key or 'idle'] += t - last
    last = t
    cnt[q] += d
print(f"step {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels: " + ", ".join(f"{k} {v / 1e6:.3f} ms" for k, v in sorted(acc.items())))
short = lambda k: k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]   # noqa: E731
other = {0: sorted((r['s'], r['e']) for r in step if r['Queue_Id'] != mainq), 1: sorted((r['s'], r['e']) for r in step if r['Queue_Id'] == mainq)}


def overlap(s, e, ivs):
    tot = 0
    for a, b in ivs:
        if b <= s:
            continue
        if a >= e:
            break
        tot += min(e, b) - max(s, a)
    return tot


for q, nm in ((0, 'main queue'), (1, 'weight-gradient queue')):
    fam = collections.defaultdict(lambda: [0, 0, 0])
    for r in step:
        if (0 if r['Queue_Id'] == mainq else 1) != q:
            continue
        f = fam[short(r['Kernel_Name'])]
        f[0] += 1; f[1] += r['e'] - r['s']; f[2] += overlap(r['s'], r['e'], other[q])
    tot = sum(f[1] for f in fam.values())
    print(f"{nm}: busy {tot / 1e6:.3f} ms")
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {f[1] / 1e6:7.3f} ms  {f[0]:4d} x  {100 * f[2] / max(f[1], 1):5.1f} % beside the other queue  {k}")
