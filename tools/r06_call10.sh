#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests/test_gpu_postproc.py tests/test_gpu_yolox.py -x -q > $O/test_post.log 2>&1; echo "test rc $?" | tee $O/test.rc
tail -3 $O/test_post.log
python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/infer.json 2> $O/infer.err
python3 -c "
import json; j=json.load(open('gpurun_out/r06j/infer.json')); print(j['value'], j['ms_per_step'], j['decode_filter'], j['nms_synthetic'])"
