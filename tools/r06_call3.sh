#!/bin/bash
# round 6, GPU call 3: the 256 x 256 tile of conv_v3_kernel (algo 14): parity tests, then kernel families side by side
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "v3_wide" > $O/test_wide.log 2>&1; echo "test rc $?" | tee $O/test.rc
tail -5 $O/test_wide.log
for m in fwd dgrad dgrad3 eval; do python tools/bench_algos.py v5l $m 10 >> $O/algos_v5l.txt 2>&1; done
BA_ONLY=s3_cba12,s4_conv,s4_b_3x3,s3_conv python tools/bench_algos.py v5x1280 eval 10 >> $O/algos_v5x.txt 2>&1
BA_ONLY=s3_conv,s3_cba12,s4_conv,s4_b_3x3,s4_cba3,spp_cba2 python tools/bench_algos.py v5s fwd 20 >> $O/algos_v5s.txt 2>&1
BA_ONLY=s3_conv,s3_cba12,s4_conv,s4_b_3x3,s4_cba3,spp_cba2 python tools/bench_algos.py v5s dgrad3 20 >> $O/algos_v5s.txt 2>&1
cat $O/algos_v5l.txt $O/algos_v5x.txt $O/algos_v5s.txt | cut -c1-260
