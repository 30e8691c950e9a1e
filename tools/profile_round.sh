#!/bin/bash
# rocprofv3 evidence of one round (run on the GPU box from the repo root): kernel-time summaries of the judged bench line with the
# kernels back to back (YH_BWD_STREAMS=0) and in the timed two-stream configuration, and the two PMC passes (FETCH_SIZE / WRITE_SIZE,
# separate runs, no trace domains beside them) that tools/pmc_traffic.py turns into profiles/pmc_traffic.json.
# usage: tools/profile_round.sh <tag e.g. r02>
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_pre.json 2> $OUT/bench_pre.err     # fills the tune cache
YH_BWD_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/b64.log 2>&1
YH_BWD_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.log 2>&1
YH_BWD_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_write.log 2>&1
S=$(find $OUT/serial -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/${TAG}_kernel_stats_serial.csv
S=$(find $OUT/b64 -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/${TAG}_kernel_stats_bench_b64.csv
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $OUT/pmc_traffic.json 4 > $OUT/pmc_top.txt
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json       # the bench lines below report `traffic` from the passes just taken (same library)
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
# the other BASELINE configs at one GPU (bench lines only): YOLOXs train, YOLOv5l train, YOLOv5x inference at 1280^2 (batch 128, per-layer table)
python3 bench.py --workload yolox --no-cpu-baseline > $OUT/${TAG}_bench_yolox.json 2> $OUT/bench_yolox.err
python3 bench.py --model large --no-cpu-baseline > $OUT/${TAG}_bench_v5l.json 2> $OUT/bench_v5l.err
YH_BENCH_LAYERS=200 python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_infer_v5x_1280_b128.json 2> $OUT/${TAG}_layers_infer_v5x_1280_b128.txt
# keep the merged-back directory small: the raw traces stay on the box
rm -rf $OUT/serial $OUT/b64 $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
