rm -f gpurun_out/r3_sweep10.log
A="YH_TUNE_CACHE=$PWD/gpurun_out/tc_a.json YH_SKIP_ALGOS=8"
Bv="YH_TUNE_CACHE=$PWD/gpurun_out/tc_b.json"
tools/sweep_env.sh gpurun_out/r3_sweep10.log "" "$A YH_BWD_STREAMS=0" "$Bv YH_BWD_STREAMS=0" "$A YH_BWD_STREAMS=0" "$Bv YH_BWD_STREAMS=0" "$A" "$Bv" "$A" "$Bv"
