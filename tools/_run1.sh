export YH_TUNE_CACHE=$PWD/gpurun_out/tc_g.json
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "inplace_concat" 2>&1 | tail -6
