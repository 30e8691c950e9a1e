#!/bin/bash
# the GPU suite N times on one box (flakiness screen before the round ends)
export TMPDIR=/tmp
O=gpurun_out/soak; mkdir -p $O
for i in $(seq 1 ${1:-2}); do
  ( time python -m pytest tests -m gpu -x -q ) > $O/run$i.log 2>&1; echo "run $i rc $?"; tail -2 $O/run$i.log | head -1
done
