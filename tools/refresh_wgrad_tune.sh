#!/bin/bash
# Re-time only the weight-gradient entries of the shipped launch-parameter table (yoloseries_amd/tune_defaults.json) after the
# weight-gradient kernels changed (engine.KEY_WGRAD bumped): the conv entries of the shipped table are kept, the workloads of
# tools/make_tune_defaults.sh are built once (their weight gradients are timed on this MI355X), and the result is merged.
# Run on an MI355X from the repo root; writes gpurun_out/tune/tune_defaults.json.
set -e
OUT=gpurun_out/tune
mkdir -p $OUT
export YH_TUNE_CACHE=$PWD/$OUT/wg_local.json YH_TUNE_ITERS=${YH_TUNE_ITERS:-12}
rm -f $YH_TUNE_CACHE
cp yoloseries_amd/tune_defaults.json $OUT/shipped_before.json
# the tracked table is stripped of its weight-gradient entries only while the workloads below run: whatever happens, it comes back
trap 'cp $OUT/shipped_before.json yoloseries_amd/tune_defaults.json' EXIT
python3 - <<'PY'
import json
t = json.load(open("yoloseries_amd/tune_defaults.json"))
json.dump({k: v for k, v in t.items() if not k.startswith("wgrad")}, open("yoloseries_amd/tune_defaults.json", "w"), indent=0, sort_keys=True)
PY
python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 bench.py --workload yolox --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 bench.py --model large --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 bench.py --model middle --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 - <<'PY'
import json, os
from yoloseries_amd import engine
shipped = json.load(open("yoloseries_amd/tune_defaults.json"))
local = json.load(open(os.environ["YH_TUNE_CACHE"]))
keep = {k: v for k, v in shipped.items() if k.split(":", 1)[0] in engine.TUNE_KEY_VERSIONS}
new = {k: v for k, v in local.items() if k.split(":", 1)[0] in engine.TUNE_KEY_VERSIONS}
keep.update(new)
json.dump(keep, open("gpurun_out/tune/tune_defaults.json", "w"), indent=0, sort_keys=True)
print(f"shipped {len(shipped)} -> kept {len(keep) - len(new)} + timed {len(new)} = {len(keep)} entries;",
      {p: sum(1 for k in keep if k.startswith(p + ':')) for p in sorted({k.split(':', 1)[0] for k in keep})})
PY
