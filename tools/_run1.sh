python3 -m pytest tests/test_gpu_elementwise.py -x -q -k "sppf" 2>&1 | tail -2
YH_BENCH_LAYERS=500 python3 bench.py --no-cpu-baseline --steps 10 2>&1 >/dev/null | grep -i "pool"
