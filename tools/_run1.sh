export YH_TUNE_CACHE=$PWD/gpurun_out/tune_local.json
python3 -m pytest tests -x -q -m gpu > gpurun_out/r3_full2.log 2>&1; tail -3 gpurun_out/r3_full2.log
