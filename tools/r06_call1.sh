#!/bin/bash
# round 6, GPU call 1: the suite after the refactors, store-policy A/B of the BN+SiLU passes, hipGraph replay, host-ahead / hole events
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "gputest rc $?" | tee $O/gputest.rc
tail -3 $O/gputest.log
for i in 1 2 3; do
  tools/sweep_env.sh $O/ab_store.txt "--steps 30 --warmup 8" "YH_EW_STORE=0" "YH_EW_STORE=1" "YH_EW_STORE=2"
done
tools/sweep_env.sh $O/ab_graph.txt "--steps 30 --warmup 8" "YH_GRAPH=0" "YH_GRAPH=1" "YH_GRAPH=0" "YH_GRAPH=1"
python tools/host_ahead.py 20 > $O/host_ahead.txt 2>&1
HA_NO_EVENTS=1 python tools/host_ahead.py 20 > $O/host_ahead_noev.txt 2>&1
python tools/cpu_leg_scaling.py 4 16 64 > $O/cpu_leg_scaling.txt 2>&1
cat $O/ab_store.txt $O/ab_graph.txt $O/host_ahead.txt $O/cpu_leg_scaling.txt
