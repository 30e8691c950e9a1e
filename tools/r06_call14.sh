#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "algos or halo" > $O/test.log 2>&1; echo "test rc $?" | tee $O/test.rc
tail -2 $O/test.log
for dbg in 4096 0 4096 0; do
  for m in fwd dgrad3; do echo "DBG=$dbg $m" >> $O/algos.txt
    YH_CONV_DBG=$dbg BA_ONLY=s2_b_3x3,s3_b_3x3,s4_b_3x3 python tools/bench_algos.py v5l $m 20 2>&1 | grep -v amdgpu | sed 's/TFLOP\/s  v2 .*| halo \(.*\) | halo160.*/ halo \1/' >> $O/algos.txt
    YH_CONV_DBG=$dbg BA_ONLY=s3_b_3x3,s4_b_3x3 python tools/bench_algos.py v5s $m 20 2>&1 | grep -v amdgpu | sed 's/TFLOP\/s  v2 .*| halo \(.*\) | halo160.*/ halo \1/' >> $O/algos.txt
  done
done
cat $O/algos.txt
for i in 1 2 3; do tools/sweep_env.sh $O/ab_halo_v5s.txt "--steps 30 --warmup 8" "YH_CONV_DBG=4096" "YH_CONV_DBG=0"; done
for i in 1 2; do tools/sweep_env.sh $O/ab_halo_yolox.txt "--workload yolox --steps 20 --warmup 5" "YH_CONV_DBG=4096" "YH_CONV_DBG=0"; done
for i in 1 2; do tools/sweep_env.sh $O/ab_halo_v5l.txt "--model large --steps 12 --warmup 4" "YH_CONV_DBG=4096" "YH_CONV_DBG=0"; done
