python3 -m pytest tests/test_gpu_conv.py -x -q -k "patch3" 2>&1 | tail -2
BA_ONLY=s1_b_3x3,s2_b_3x3 python3 tools/bench_algos.py v5s dgrad3 20 2>&1 | grep -v amdgpu.ids | cut -c1-64,176-
