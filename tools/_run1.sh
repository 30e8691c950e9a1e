timeout -k 10 300 python3 tools/fuzz_new_kernels.py 60 3 2>&1 | tail -25
