#!/bin/bash
# Time the layer shapes the shipped launch-parameter table (yoloseries_amd/tune_defaults.json) has no entry for — after a new
# key version was introduced for a class of layers (engine.KEY_*) — and merge them into the table.  The workloads of
# tools/make_tune_defaults.sh are built once with a local cache of their own; only the missing shapes are timed.
# Run on an MI355X from the repo root; writes gpurun_out/tune/tune_defaults.json.
set -e
OUT=gpurun_out/tune
mkdir -p $OUT
export YH_TUNE_CACHE=$PWD/$OUT/missing_local.json YH_TUNE_ITERS=${YH_TUNE_ITERS:-12}
rm -f $YH_TUNE_CACHE
NOB="--no-cpu-baseline --no-roofline"
python3 bench.py --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --workload yolox --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --model large --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --model middle --steps 3 --warmup 2 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model xlarge --img 1280 --batch 32 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 bench.py --workload infer --model small --img 640 --batch 64 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
python3 - <<'PY'
import json, os
from yoloseries_amd import engine
shipped = json.load(open("yoloseries_amd/tune_defaults.json"))
local = json.load(open(os.environ["YH_TUNE_CACHE"])) if os.path.exists(os.environ["YH_TUNE_CACHE"]) else {}
keep = {k: v for k, v in shipped.items() if k.split(":", 1)[0] in engine.TUNE_KEY_VERSIONS}
new = {k: v for k, v in local.items() if k.split(":", 1)[0] in engine.TUNE_KEY_VERSIONS and k not in keep}
for k, v in sorted(new.items()):
    print("  +", k, v)
keep.update(new)
json.dump(keep, open("gpurun_out/tune/tune_defaults.json", "w"), indent=0, sort_keys=True)
print(f"shipped {len(shipped)} -> kept {len(keep) - len(new)} + timed {len(new)} = {len(keep)} entries")
PY
