#!/bin/bash
# What the data-parallel path costs on ONE GPU (YH_FORCE_DP=1 keeps the communicator, the bucket hooks and the finishers for a single
# rank): bench lines with and without it, without the stream probe (yoloseries_amd/streams.py), with the collectives on torch's own
# stream, with one bucket, with the exchange after the backward, and a kernel trace of the DP step.
# Run on an MI355X from the repo root; writes gpurun_out/dp/.
set -e
OUT=gpurun_out/dp
mkdir -p $OUT
export TMPDIR=/tmp
NOB="--no-cpu-baseline --no-roofline"
M=${1:-small}
run() { local tag=$1; shift; env "$@" python3 bench.py --model $M --steps 20 --warmup 5 $NOB 2> $OUT/$tag.err | tail -1 > $OUT/$tag.json; python3 -c "
import json,sys; j=json.load(open('$OUT/$tag.json')); print('$tag', j['value'], j['ms_per_step'])"; }
run plain YH_X=0
run dp YH_FORCE_DP=1
run dp_noprobe YH_FORCE_DP=1 YH_STREAM_PROBE=0
run dp_torch_stream YH_FORCE_DP=1 YH_DP_COMM_STREAM=0
run dp_b1 YH_FORCE_DP=1 YH_DP_BUCKETS=1
run dp_after YH_FORCE_DP=1 YH_DP_OVERLAP=0
run dp_serial YH_FORCE_DP=1 YH_BWD_STREAMS=0
run plain_serial YH_BWD_STREAMS=0
YH_FORCE_DP=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --model $M --steps 8 --warmup 3 $NOB > $OUT/trace.log 2>&1
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/step_trace.py "$T" > $OUT/step_trace.txt
python3 tools/step_overlap.py "$T" > $OUT/step_overlap.txt
python3 tools/foreign_launches.py "$T" > $OUT/foreign.txt
rm -rf $OUT/trace
