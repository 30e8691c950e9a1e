#!/bin/bash
# round 6, GPU call 5: conv_p3_kernel at stride 2 — parity tests, then the kernel families side by side on the downsampling layers
export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "patch3" > $O/test_p3.log 2>&1; echo "test rc $?" | tee $O/test.rc
tail -5 $O/test_p3.log
BA_ONLY=s1_conv,s2_conv,s3_conv python tools/bench_algos.py v5s fwd 20 >> $O/algos.txt 2>&1
BA_ONLY=s1_conv,s2_conv python tools/bench_algos.py v5l fwd 20 >> $O/algos.txt 2>&1
BA_TILE_N=32 BA_ONLY=s1_conv,s2_conv,s3_conv python tools/bench_algos.py v5s fwd 20 >> $O/algos_pt1.txt 2>&1
BA_TILE_N=32 BA_ONLY=s1_conv,s2_conv python tools/bench_algos.py v5l fwd 20 >> $O/algos_pt1.txt 2>&1
cat $O/algos.txt $O/algos_pt1.txt | cut -c1-250
