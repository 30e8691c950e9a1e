#!/bin/bash
# Re-time the layer shapes of the shipped launch-parameter table that a NEW kernel family could take against all their other
# candidates, and merge the result.  usage: refresh_tune_family.sh <family>
#   wide   the 256 x 256 tile of conv_v3_kernel (yh_conv_desc.algo 14): N a multiple of 256, whole 64-channel blocks in every segment
#   p3s2   conv_p3_kernel at stride 2 (algo 8): 3x3 / stride-2 forward layers of the training programs with <= 128 channels in and out
#   v3w8   the 8-wave tiles of conv_v3_kernel (algo 2 / 14) after a change to their main loop: every shape with N > 64 and 32-channel blocks
#   allconv  every forward / data-gradient / inference entry
# The BASELINE workloads are built once with those entries removed from the table and a local cache of their own; every other
# entry is kept as shipped.  Run on an MI355X from the repo root; writes gpurun_out/tune/tune_defaults.json.
set -e
export REFRESH_FAMILY=${1:-wide}
OUT=gpurun_out/tune
mkdir -p $OUT
cp yoloseries_amd/tune_defaults.json $OUT/shipped_before.json
python3 - <<'PY'
import json
t = json.load(open("yoloseries_amd/tune_defaults.json"))
import os
def family_ok(k):
    p = k.split(":")
    if not p[0].startswith("conv") or len(p) != 3:
        return False
    f = [int(x) for x in p[2].split(",")]
    mode, KH, stride, N, nseg, C0, C1 = f[0], f[6], f[7], f[9], f[10], f[11], f[14]
    if os.environ["REFRESH_FAMILY"] == "allconv":
        return True
    if os.environ["REFRESH_FAMILY"] == "v3w8":        # shapes the 8-wave tiles of the ring kernel can take (K-heavy, N > 64)
        return N > 64 and C0 % 32 == 0 and (nseg == 1 or C1 % 32 == 0)
    if os.environ["REFRESH_FAMILY"] == "p3s2":
        return p[1] == "fwd" and mode == 0 and KH == 3 and stride == 2 and nseg == 1 and C0 % 32 == 0 and C0 <= 128 and N <= 128
    return N % 256 == 0 and C0 % 64 == 0 and (nseg == 1 or C1 % 64 == 0)
drop = [k for k in t if family_ok(k)]
json.dump({k: v for k, v in t.items() if k not in drop}, open("yoloseries_amd/tune_defaults.json", "w"), indent=0, sort_keys=True)
print(f"{len(drop)} of {len(t)} entries to re-time")
PY
export YH_TUNE_CACHE=$PWD/$OUT/wide_local.json YH_TUNE_ITERS=${YH_TUNE_ITERS:-12}
rm -f $YH_TUNE_CACHE
NOB="--no-cpu-baseline --no-roofline"
restore() { cp $OUT/shipped_before.json yoloseries_amd/tune_defaults.json; }
trap restore EXIT
python3 bench.py --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --workload yolox --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --model large --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --model middle --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model xlarge --img 1280 --batch 32 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model small --img 640 --batch 64 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 - <<'PY'
import json, os
kept = json.load(open("yoloseries_amd/tune_defaults.json"))
before = json.load(open("gpurun_out/tune/shipped_before.json"))
local = json.load(open(os.environ["YH_TUNE_CACHE"])) if os.path.exists(os.environ["YH_TUNE_CACHE"]) else {}
new = {k: v for k, v in local.items() if k in before and k not in kept}
lost = [k for k in before if k not in kept and k not in new]
for k in lost:                  # a shape no workload above builds any more keeps its old entry
    new[k] = before[k]
fam_algo = 8 if os.environ["REFRESH_FAMILY"] == "p3s2" else 14
n14 = sum(1 for v in new.values() if len(v) == 3 and v[2] == fam_algo)
for k, v in sorted(new.items()):
    if v != before[k]:
        print("  ", k, before[k], "->", v)
kept.update(new)
json.dump(kept, open("gpurun_out/tune/tune_defaults.json", "w"), indent=0, sort_keys=True)
print(f"re-timed {len(new) - len(lost)} entries ({len(lost)} kept as shipped: not built by the workloads), {n14} take the new family (algo {fam_algo}); {len(kept)} entries in all")
PY
