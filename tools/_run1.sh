python3 -m pytest tests/test_gpu_postproc.py tests/test_gpu_model.py -x -q -k "multi_label or eval_mode_bn" -s > gpurun_out/r3_t4.log 2>&1; grep -v amdgpu.ids gpurun_out/r3_t4.log | tail -25
