set -e
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo80" 2>&1 | tail -3
for r in "" 1; do BA_RES=$r BA_BATCH=128 BA_ONLY=s1_b_3x3 python3 tools/bench_algos.py v5x1280 eval 10 2>&1 | grep eval | sed 's/.*h80/h80/'; done
