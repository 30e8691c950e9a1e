#!/usr/bin/env python3
"""Ceiling statement of a workload from its per-launch table (profiles/rNN_layers_*.txt, written by `YH_BENCH_LAYERS=N bench.py`):
per launch the time its ALGORITHMIC work needs on the achievable roofs — max(bytes / 5.5 TB/s, FLOP / 1.2 PFLOP/s): the HBM rate the
streaming passes of this repo reach, and (as an upper bound for this design) a little under the best plain-HIP bf16 GEMM measured on this
chip (MI355X_MICROARCH.md / cdna_hip_programming.md: 1.32-1.47 PFLOP/s on random data) — summed, against the measured kernel time; and the
conv-only lower bound of SURVEY.md section 8d (elements x 2 B x 3 passes at the 8 TB/s spec).
usage: ceiling.py <layers file> <batch> [img/s measured]"""
import re
import sys

HBM, MFMA = 5.5e12, 1.2e15
path, batch = sys.argv[1], int(sys.argv[2])
rows = []
for ln in open(path):
    m = re.match(r"#\s+([\d.]+) ms/step\s+(.*\S)\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s", ln)
    if m:
        ms, tf, gbs = float(m.group(1)), float(m.group(3)), float(m.group(4))
        rows.append((ms, tf * 1e12 * ms * 1e-3, gbs * 1e9 * ms * 1e-3, m.group(2).strip()))
tot = sum(r[0] for r in rows)
bound = sum(max(r[2] / HBM, r[1] / MFMA) for r in rows) * 1e3
hb = sum(r[2] / HBM for r in rows if r[2] / HBM >= r[1] / MFMA) * 1e3
nb = sum(1 for r in rows if r[2] / HBM >= r[1] / MFMA)
byt = sum(r[2] for r in rows)
fl = sum(r[1] for r in rows)
print(f"{len(rows)} launches, {tot:.2f} ms of kernels back to back; algorithmic {byt / 1e9:.1f} GB, {fl / 1e12:.2f} TFLOP per step")
print(f"sum over launches of max(bytes / 5.5 TB/s, FLOP / 1.2 PF) = {bound:.2f} ms  ({100 * bound / tot:.0f} % of the kernel time) -> {batch / bound * 1e3:.0f} img/s"
      f"  [{nb} launches ({hb:.2f} ms) on the HBM side]")
print(f"all bytes at 8 TB/s spec: {byt / 8e12 * 1e3:.2f} ms -> {batch / (byt / 8e12):.0f} img/s; all FLOP at 2.5 PF: {fl / 2.5e15 * 1e3:.2f} ms")
if len(sys.argv) > 3:
    ips = float(sys.argv[3])
    print(f"measured {ips:.0f} img/s = {batch / ips * 1e3:.2f} ms per step = {100 * bound / (batch / ips * 1e3):.0f} % of that ceiling's step rate")
