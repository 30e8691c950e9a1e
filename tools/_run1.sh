set -e
export YH_TUNE_CACHE=$PWD/gpurun_out/tc_h.json
timeout -k 10 600 python3 -m pytest tests/test_gpu_elementwise.py -x -q -m gpu -k "sppf" 2>&1 | tail -3
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "evaluator or golden" 2>&1 | tail -3
YH_BENCH_LAYERS=200 python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --no-cpu-baseline --steps 6 --warmup 3 > gpurun_out/b_v5x.json 2> gpurun_out/layers_v5x.txt || { tail -20 gpurun_out/layers_v5x.txt; exit 1; }
cut -c1-200 gpurun_out/b_v5x.json; grep -i "pool" gpurun_out/layers_v5x.txt | cut -c1-150
