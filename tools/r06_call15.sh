#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06o; mkdir -p $O
python tools/dp_judged_diag.py 8 small 64 640 > $O/diag_2rank_small.txt 2>&1
grep -v "amdgpu\|Gloo\|socket" $O/diag_2rank_small.txt | grep -B1 -A1 "head gradients" | cut -c1-900
