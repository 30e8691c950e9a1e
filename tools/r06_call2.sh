#!/bin/bash
# round 6, GPU call 2: the deepened race screen + the data-parallel step at the judged shape; row-order A/B of the BN+SiLU passes (v5s);
# static wave priority A/B of the 8-wave MFMA kernels (v5l)
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests/test_gpu_dist.py tests/test_gpu_tune_table.py -x -q -k "launch_to_launch or judged_shape_ or two_ranks" ) > $O/gputest_sel.log 2>&1; echo "gputest rc $?" | tee $O/gputest.rc
tail -4 $O/gputest_sel.log
for i in 1 2 3; do
  tools/sweep_env.sh $O/ab_rev.txt "--steps 30 --warmup 8" "YH_EW_REV=0" "YH_EW_REV=1" "YH_EW_REV=2" "YH_EW_REV=3"
done
for i in 1 2; do
  tools/sweep_env.sh $O/ab_prio_v5l.txt "--model large --steps 12 --warmup 4" "YH_CONV_DBG=0" "YH_CONV_DBG=4096" "YH_CONV_DBG=8192" "YH_CONV_DBG=12288"
done
tools/sweep_env.sh $O/ab_prio_v5s.txt "--steps 30 --warmup 8" "YH_CONV_DBG=0" "YH_CONV_DBG=12288" "YH_CONV_DBG=0" "YH_CONV_DBG=12288"
cat $O/ab_rev.txt $O/ab_prio_v5l.txt $O/ab_prio_v5s.txt
