set -e
bash tools/refresh_tune_missing.sh > gpurun_out/refresh2.log 2>&1
tail -3 gpurun_out/refresh2.log
cp gpurun_out/tune/tune_defaults.json yoloseries_amd/tune_defaults.json
run() { echo "$*" >> gpurun_out/ab_halo2.txt; env "${@:2}" python bench.py --no-cpu-baseline --no-roofline $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('tuning'))" >> gpurun_out/ab_halo2.txt; }
for i in 1 2; do
run "--workload infer --model xlarge --img 1280 --batch 128 --steps 5 --warmup 2" YH_HALO_W4=0
run "--workload infer --model xlarge --img 1280 --batch 128 --steps 5 --warmup 2" YH_X=1
done
cat gpurun_out/ab_halo2.txt
