#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "v3_wide" > $O/test_wide.log 2>&1; echo "test rc $?" | tee $O/test.rc
for m in dgrad3; do python tools/bench_algos.py v5l $m 10 >> $O/algos_v5l.txt 2>&1; done
BA_ONLY=s3_conv,s3_cba12,s4_conv,s4_b_3x3,s4_cba3,spp_cba2 python tools/bench_algos.py v5s dgrad3 20 >> $O/algos_v5s.txt 2>&1
cat $O/algos_v5l.txt $O/algos_v5s.txt | cut -c1-250
bash tools/refresh_tune_family.sh wide > $O/refresh.log 2>&1; echo "refresh rc $?" | tee $O/refresh.rc
tail -25 $O/refresh.log
cp gpurun_out/tune/tune_defaults.json $O/tune_new.json
cp gpurun_out/tune/shipped_before.json $O/tune_old.json
cp $O/tune_new.json yoloseries_amd/tune_defaults.json
python -m pytest tests/test_gpu_tune_table.py -x -q > $O/test_table.log 2>&1; echo "table tests rc $?" | tee -a $O/test.rc
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
ab v5l --model large --steps 12 --warmup 4
cp $O/tune_new.json yoloseries_amd/tune_defaults.json
