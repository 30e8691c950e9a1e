set -e
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pointwise" 2>&1 | tail -5
BA_BATCH=128 BA_ONLY=s2_cba12,s3_cba12 python3 tools/bench_algos.py v5x1280 eval 10 2>&1 | grep eval | sed 's/TFLOP.s//'
export YH_TUNE_CACHE=$PWD/gpurun_out/tc_g.json
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or evaluator" 2>&1 | tail -3
YH_BENCH_LAYERS=200 python3 bench.py --workload infer --model xlarge --img 1280 --batch 128 --no-cpu-baseline --steps 6 --warmup 3 > gpurun_out/b_v5x.json 2> gpurun_out/layers_v5x.txt || { tail -20 gpurun_out/layers_v5x.txt; exit 1; }
cut -c1-200 gpurun_out/b_v5x.json
