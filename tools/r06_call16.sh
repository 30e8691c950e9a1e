#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06q; mkdir -p $O
python tools/dp_judged_diag.py 10 small 64 640 > $O/diag_2rank_small.txt 2>&1
grep -v "amdgpu\|Gloo\|socket" $O/diag_2rank_small.txt | cut -c1-400 | tail -40
timeout -k 10 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_postproc.py tests/test_gpu_yolox.py -m gpu -x -q 2>&1 | tail -3
