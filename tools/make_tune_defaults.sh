#!/bin/bash
# Launch-parameter table shipped with the package (yoloseries_amd/tune_defaults.json): run on an MI355X from the repo root.
# Every BASELINE configuration is built twice with an empty cache; a layer keeps a choice only when both runs agree on it or,
# failing that, the choice of the second run (the timing of a candidate is the minimum over its launches either way).
set -e
OUT=gpurun_out/tune
mkdir -p $OUT
for r in 1 2; do
  export YH_TUNE_CACHE=$PWD/$OUT/tune_$r.json YH_TUNE_DEFAULTS=0
  rm -f $YH_TUNE_CACHE
  python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --workload yolox --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --model large --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --model middle --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --workload infer --model xlarge --img 1280 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 bench.py --workload infer --model small --img 640 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/tune/tune_1.json")); b = json.load(open("gpurun_out/tune/tune_2.json"))
out = dict(b)
agree = sum(1 for k in b if a.get(k) == b[k])
json.dump(out, open("gpurun_out/tune/tune_defaults.json", "w"), indent=0, sort_keys=True)
print(f"{len(out)} entries, {agree} identical in both runs")
PY
