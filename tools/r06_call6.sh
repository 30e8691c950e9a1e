#!/bin/bash
# round 6, GPU call 6: re-time the stride-2 forward layers conv_p3_kernel<..., 2> can take; table walk + race screen; A/B old / new table (v5s, yolox)
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
bash tools/refresh_tune_family.sh p3s2 > $O/refresh.log 2>&1; echo "refresh rc $?" | tee $O/refresh.rc
tail -12 $O/refresh.log
cp gpurun_out/tune/tune_defaults.json $O/tune_new.json
cp gpurun_out/tune/shipped_before.json $O/tune_old.json
cp $O/tune_new.json yoloseries_amd/tune_defaults.json
python -m pytest tests/test_gpu_tune_table.py -x -q > $O/test_table.log 2>&1; echo "table tests rc $?" | tee $O/test.rc
tail -3 $O/test_table.log
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
ab yolox --workload yolox --steps 20 --warmup 5
cp $O/tune_new.json yoloseries_amd/tune_defaults.json
