#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [steps profiled] [workload tag]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) streaming reads
at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores and float atomics.  Infinity-cache hits are
counted as traffic.  Output: average bytes per launch for every kernel (short name as bench.py reports it)."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):          # cut the argument list: first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return name[:i]
    return name


def collect(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, (0.0, 0))
        ws, wn = write.get(k, (0.0, 0))
        fb = 2.0 * 1024.0 * fs / max(fn, 1)          # gfx950 correction: x2
        wb = 1024.0 * ws / max(wn, 1)
        out[k] = {"launches": int(max(fn, wn)), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb)}
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yoloseries_amd", "libyolohip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
    # framework fills at buffer-allocation time and runtime copies are not part of a step
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in out.items() if not k.startswith(("at::native", "__amd_rocclr")))
    with open(sys.argv[3], "w") as f:
        json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950), unit 1024 B; "
                            "lib_sha16 = sha256 of the libyolohip.so the passes ran with (bench.py drops the numbers when it differs)",
                   "lib_sha16": sha, "workload": (sys.argv[5] if len(sys.argv) > 5 else "train:small:64:640"),
                   "steps_profiled": steps, "hbm_bytes_all_launches": total,
                   "hbm_bytes_per_step": (total / steps if steps else None), "kernels": out}, f, indent=1)
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k[:60]:60s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB write {v['write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
