export YH_TUNE_CACHE=$PWD/gpurun_out/tc_d.json
python3 bench.py --model middle --no-cpu-baseline --no-roofline --steps 10 2>&1 | tail -1 | cut -c1-200
python3 bench.py --model xlarge --batch 32 --no-cpu-baseline --no-roofline --steps 6 2>&1 | tail -1 | cut -c1-200
python3 bench.py --model small --batch 16 --img 1280 --no-cpu-baseline --no-roofline --steps 6 2>&1 | tail -1 | cut -c1-200
python3 bench.py --workload infer --model small --batch 64 --no-cpu-baseline --no-roofline --steps 6 2>&1 | tail -1 | cut -c1-200
python3 tools/check_models.py 2>&1 | tail -5
