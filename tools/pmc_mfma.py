#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass of the bench command
(SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; no trace domains beside it).

usage: pmc_mfma.py <counter_collection.csv> <out.json> [steps profiled] [workload tag]

Units (/opt/skills/guides/MI355X_MICROARCH.md, PMC slots and cycle constants): SQ_INSTS_VALU_MFMA_MOPS_BF16 counts executed
multiply / add operations / 512; SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix unit is busy, summed over the SIMDs
(32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the dispatch's busy time in cycles summed over the 8 XCDs.
mfma_util = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the fraction of the chip's matrix-unit cycles the kernel used,
at the clock the chip actually held (so it is not deflated by DVFS, unlike TFLOP/s against the 2.5 PF spec)."""
import csv, hashlib, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short

SIMDS = 256 * 4


def main():
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(sys.argv[1], newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    out = {}
    for k, c in acc.items():
        n = max(len(disp[k]), 1)
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[k] = {"launches": n, "mfma_flops_per_launch": round(512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / n),
                  "mfma_busy_cycles_per_launch": round(busy / n), "gpu_cycles_per_launch": round(cyc / n),
                  "mfma_util": round(busy / (SIMDS * cyc), 4) if cyc > 0 else None}
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yoloseries_amd", "libyolohip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    mine = {k: v for k, v in out.items() if not k.startswith(("at::native", "__amd_rocclr"))}
    busy_all = sum(v["mfma_busy_cycles_per_launch"] * v["launches"] for v in mine.values())
    cyc_all = sum(v["gpu_cycles_per_launch"] * v["launches"] for v in mine.values())
    with open(sys.argv[2], "w") as f:
        json.dump({"_note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, kernels back to back); "
                            "mfma_util = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
                   "lib_sha16": sha, "workload": (sys.argv[4] if len(sys.argv) > 4 else "train:small:64:640"), "steps_profiled": steps,
                   "mfma_util_all_kernels": round(busy_all / (SIMDS * cyc_all), 4) if cyc_all else None,
                   "mfma_flops_per_step": (sum(v["mfma_flops_per_launch"] * v["launches"] for v in mine.values()) / steps if steps else None),
                   "kernels": out}, f, indent=1)
    for k, v in sorted(mine.items(), key=lambda kv: -kv[1]["gpu_cycles_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:64]:64s} n={v['launches']:5d} mfma_util {v['mfma_util']}  flops/launch {v['mfma_flops_per_launch'] / 1e9:8.2f} G")


if __name__ == "__main__":
    main()
