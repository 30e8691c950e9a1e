#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q > $O/test_conv.log 2>&1; echo "conv tests rc $?" | tee $O/test.rc
tail -2 $O/test_conv.log
( time bash tools/refresh_tune_family.sh v3w8 ) > $O/refresh.log 2>&1; echo "refresh rc $?" | tee $O/refresh.rc
tail -6 $O/refresh.log
cp gpurun_out/tune/tune_defaults.json $O/tune_new.json
cp gpurun_out/tune/shipped_before.json $O/tune_old.json
ab() {
  local label=$1; shift
  for i in 1 2 3; do
    for tb in old new; do
      cp $O/tune_$tb.json yoloseries_amd/tune_defaults.json
      v=$(python3 bench.py "$@" --no-cpu-baseline --no-roofline 2>>$O/ab.err | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")
      echo "$label table=$tb -> $v" | tee -a $O/ab_table.txt
    done
  done
}
ab v5s --steps 30 --warmup 8
ab v5l --model large --steps 12 --warmup 4
ab yolox --workload yolox --steps 20 --warmup 5
ab v5x_infer --workload infer --model xlarge --img 1280 --batch 128 --steps 5 --warmup 2
cp $O/tune_new.json yoloseries_amd/tune_defaults.json
python -m pytest tests/test_gpu_tune_table.py -x -q > $O/test_table.log 2>&1; echo "table tests (new table) rc $?" | tee -a $O/test.rc
tail -2 $O/test_table.log
