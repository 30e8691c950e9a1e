#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06l; mkdir -p $O
YH_CONV_DBG=8192 python -m pytest tests/test_gpu_conv.py -x -q -k "algos or v3_wide or beyond_two_gib" > $O/test.log 2>&1; echo "test(8192) rc $?" | tee $O/test.rc
YH_CONV_DBG=24576 python -m pytest tests/test_gpu_conv.py -x -q -k "algos" >> $O/test.log 2>&1; echo "test(24576) rc $?" | tee -a $O/test.rc
tail -3 $O/test.log
for dbg in 0 8192 24576; do
  for m in fwd dgrad3 eval; do echo "DBG=$dbg $m v5l" >> $O/algos.txt; YH_CONV_DBG=$dbg BA_ONLY=s1_conv,s2_conv,s2_b_3x3,s3_conv,s3_b_3x3,s3_cba12,s4_conv,s4_cba3 python tools/bench_algos.py v5l $m 20 2>&1 | grep -v amdgpu | sed 's/TFLOP\/s  v2 .*| v3-256x128 \(.*\) | v3-128x128 \(.*\) | v3-128x64.*/ v3-256x128 \1 (128x128 \2)/' >> $O/algos.txt; done
  echo "DBG=$dbg eval v5x" >> $O/algos.txt; YH_CONV_DBG=$dbg BA_ONLY=s1_conv,s2_conv,s2_cba12,s3_conv,s3_cba12,s4_conv,s4_b_3x3 python tools/bench_algos.py v5x1280 eval 20 2>&1 | grep -v amdgpu | sed 's/TFLOP\/s  v2 .*| v3-256x128 \(.*\) | v3-128x128 \(.*\) | v3-128x64.*/ v3-256x128 \1 (128x128 \2)/' >> $O/algos.txt
done
cat $O/algos.txt
