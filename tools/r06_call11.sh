#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "v3_wide" > $O/test_wide.log 2>&1; echo "test rc $?" | tee $O/test.rc
for dbg in 4096 0; do
  for m in fwd dgrad dgrad3 eval; do echo "YH_CONV_DBG=$dbg $m" >> $O/algos.txt; YH_CONV_DBG=$dbg BA_ONLY=s2_conv,s3_conv,s3_b_3x3,s4_conv,s4_b_3x3,s4_cba3 python tools/bench_algos.py v5l $m 20 2>&1 | grep -v amdgpu | sed 's/.*k\([13]\)s\([12]\)  TFLOP.*v3-256x256/k\1s\2 v3-256x256/' >> $O/algos.txt; done
done
cat $O/algos.txt
for i in 1 2 3; do tools/sweep_env.sh $O/ab_split_v5l.txt "--model large --steps 12 --warmup 4" "YH_CONV_DBG=4096" "YH_CONV_DBG=0"; done
for i in 1 2; do tools/sweep_env.sh $O/ab_split_v5x.txt "--workload infer --model xlarge --img 1280 --batch 128 --steps 5 --warmup 2" "YH_CONV_DBG=4096" "YH_CONV_DBG=0"; done
