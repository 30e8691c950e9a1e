rm -f gpurun_out/r3_sweep8.log
tools/sweep_env.sh gpurun_out/r3_sweep8.log "" "YH_X=1" "YH_X=2"
tools/sweep_env.sh gpurun_out/r3_sweep8.log "--model large" "YH_X=1"
tools/sweep_env.sh gpurun_out/r3_sweep8.log "--workload yolox" "YH_X=1"
git stash -q 2>/dev/null
