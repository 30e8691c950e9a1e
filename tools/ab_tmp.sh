set -e
run() { echo "$*" >> gpurun_out/ab_group4.txt; env "${@:2}" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab_group4.txt; }
for i in 1 2; do
run "--model large" YH_WGS_GROUP=0
run "--model large" YH_WGS_GROUP=8
run "--workload yolox" YH_WGS_GROUP=0
run "--workload yolox" YH_WGS_GROUP=8
done
cat gpurun_out/ab_group4.txt
