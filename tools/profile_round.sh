#!/bin/bash
# rocprofv3 evidence of one round (run on the GPU box from the repo root): kernel-time summaries of the judged bench line with the
# kernels back to back (YH_BWD_STREAMS=0) and in the timed two-stream configuration, the two HBM-traffic PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate runs, no trace domains beside them -> tools/pmc_traffic.py) and the MFMA pass (SQ_INSTS_VALU_MFMA_MOPS_BF16
# SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> tools/pmc_mfma.py), for the judged line and — traffic + MFMA only — for the other
# BASELINE configurations at one GPU.  The JSON summaries land in profiles/ (stamped with the library hash and the workload).
# usage: tools/profile_round.sh <tag e.g. r03> [quick]
set -e
TAG=${1:-r03}
QUICK=${2:-}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
NOB="--no-cpu-baseline --no-roofline"

pmc_passes() {   # <name suffix> <workload tag> <steps> <bench args...>
  local SUF=$1 WL=$2 ST=$3; shift 3
  YH_BWD_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf$SUF -- python3 bench.py "$@" --steps $ST --warmup 1 $NOB > $OUT/pf$SUF.log 2>&1
  YH_BWD_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw$SUF -- python3 bench.py "$@" --steps $ST --warmup 1 $NOB > $OUT/pw$SUF.log 2>&1
  YH_BWD_STREAMS=0 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pm$SUF -- python3 bench.py "$@" --steps $ST --warmup 1 $NOB > $OUT/pm$SUF.log 2>&1
  local F=$(find $OUT/pf$SUF -name "*counter_collection.csv" | head -1) W=$(find $OUT/pw$SUF -name "*counter_collection.csv" | head -1) M=$(find $OUT/pm$SUF -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_traffic.py "$F" "$W" profiles/pmc_traffic$SUF.json $((ST + 1)) $WL > $OUT/pmc_top$SUF.txt
  python3 tools/pmc_mfma.py "$M" profiles/pmc_mfma$SUF.json $((ST + 1)) $WL > $OUT/pmc_mfma_top$SUF.txt
  rm -rf $OUT/pf$SUF $OUT/pw$SUF $OUT/pm$SUF
}

python3 bench.py --steps 5 --warmup 2 $NOB > $OUT/bench_pre.json 2> $OUT/bench_pre.err     # times the layer shapes the shipped table lacks
YH_BWD_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 10 --warmup 3 $NOB > $OUT/serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64 -- python3 bench.py --steps 10 --warmup 3 $NOB > $OUT/b64.log 2>&1
S=$(find $OUT/serial -name "*kernel_stats.csv" | head -1); cp "$S" profiles/${TAG}_kernel_stats_serial.csv
S=$(find $OUT/b64 -name "*kernel_stats.csv" | head -1); cp "$S" profiles/${TAG}_kernel_stats_bench_b64.csv
T=$(find $OUT/b64 -name "*kernel_trace.csv" | head -1)
python3 tools/step_trace.py "$T" > profiles/${TAG}_step_trace.txt
python3 tools/step_overlap.py "$T" > profiles/${TAG}_step_overlap.txt
python3 tools/foreign_launches.py "$T" > profiles/${TAG}_foreign_launches.txt
rm -rf $OUT/serial $OUT/b64
pmc_passes "" train:small:64:640 3
python3 bench.py --steps 20 --warmup 5 > profiles/${TAG}_bench_default.json 2> $OUT/bench_default.err
YH_BENCH_LAYERS=400 python3 bench.py --steps 10 --no-cpu-baseline > /dev/null 2> profiles/${TAG}_layers_v5s_train_b64.txt
# the data-parallel path's own cost at one GPU: communicator, bucket hooks and finishers kept for a single rank (utils/dist.py)
YH_FORCE_DP=1 python3 bench.py --steps 20 --warmup 5 $NOB > profiles/${TAG}_bench_force_dp.json 2> $OUT/bench_force_dp.err
if [ -z "$QUICK" ]; then
  # the other BASELINE configs at one GPU: YOLOXs train, YOLOv5l train, YOLOv5x inference at 1280^2 (batch 128, per-layer table)
  python3 bench.py --workload yolox --steps 3 --warmup 2 $NOB > /dev/null 2>&1
  pmc_passes _yolox_small_64_640 yolox:small:64:640 2 --workload yolox
  python3 bench.py --workload yolox --no-cpu-baseline > profiles/${TAG}_bench_yolox.json 2> $OUT/bench_yolox.err
  python3 bench.py --model large --steps 3 --warmup 2 $NOB > /dev/null 2>&1
  pmc_passes _train_large_64_640 train:large:64:640 2 --model large
  python3 bench.py --model large --no-cpu-baseline > profiles/${TAG}_bench_v5l.json 2> $OUT/bench_v5l.err
  YH_FORCE_DP=1 python3 bench.py --model large $NOB > profiles/${TAG}_bench_v5l_force_dp.json 2> $OUT/bench_v5l_force_dp.err
  YH_BENCH_LAYERS=600 python3 bench.py --model large --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> profiles/${TAG}_layers_v5l_train_b64.txt
  python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 2 --warmup 1 $NOB > /dev/null 2>&1
  pmc_passes _infer_xlarge_128_1280 infer:xlarge:128:1280 1 --workload infer --model xlarge --img 1280 --batch 128
  YH_BENCH_LAYERS=200 python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 6 --warmup 3 --no-cpu-baseline > profiles/${TAG}_bench_infer_v5x_1280_b128.json 2> profiles/${TAG}_layers_infer_v5x_1280_b128.txt
fi
# gpurun merges only gpurun_out/ back: everything for profiles/ travels in $OUT/profiles (copy it over profiles/ afterwards)
mkdir -p $OUT/profiles
cp profiles/pmc_traffic*.json profiles/pmc_mfma*.json profiles/${TAG}_* $OUT/profiles/ 2>/dev/null || true
ls -la $OUT $OUT/profiles
