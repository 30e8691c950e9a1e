rm -f gpurun_out/r3_sweep7.log
tools/sweep_env.sh gpurun_out/r3_sweep7.log "" "YH_WG_DEFER=0" "YH_WG_DEFER=1" "YH_WG_DEFER=0" "YH_WG_DEFER=1" "YH_WG_DEFER=1 YH_GZ_RING=64"
