bash tools/make_tune_defaults.sh 2 > gpurun_out/tune_r3.log 2>&1
tail -5 gpurun_out/tune_r3.log
ls -la gpurun_out/tune/
