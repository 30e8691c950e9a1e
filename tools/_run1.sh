export YH_TUNE_CACHE=$PWD/gpurun_out/tune_local.json
python3 tools/launch_floor.py 2>/dev/null
HIP_FORCE_DEV_KERNARG=1 python3 tools/launch_floor.py 2>/dev/null
HIP_FORCE_DEV_KERNARG=0 python3 tools/launch_floor.py 2>/dev/null
rm -f gpurun_out/r3_sweep4.log
tools/sweep_env.sh gpurun_out/r3_sweep4.log "" "YH_WGRAD_PARTIAL=0" "YH_WGRAD_PARTIAL=0 HIP_FORCE_DEV_KERNARG=1" "YH_WGRAD_PARTIAL=0 HIP_FORCE_DEV_KERNARG=0" "YH_WGRAD_PARTIAL=0 GPU_MAX_HW_QUEUES=2" "YH_WGRAD_PARTIAL=0 GPU_MAX_HW_QUEUES=8" "YH_WGRAD_PARTIAL=0 HSA_ENABLE_SDMA=0" "YH_WGRAD_PARTIAL=1" "YH_WGRAD_PARTIAL=0"
