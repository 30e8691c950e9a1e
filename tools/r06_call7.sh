#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
YH_BENCH_LAYERS=400 python3 bench.py --steps 10 --no-cpu-baseline > $O/bench_v5s.json 2> $O/layers_v5s.txt
YH_BENCH_LAYERS=600 python3 bench.py --model large --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_v5l.json 2> $O/layers_v5l.txt
python3 tools/ceiling.py $O/layers_v5s.txt 64
python3 tools/ceiling.py $O/layers_v5l.txt 64
python3 -c "
import json
for f in ('v5s','v5l'):
    j=json.load(open('gpurun_out/r06g/bench_%s.json'%f)); print(f, j['value'], j['ms_per_step'], j['roofline']['groups'].keys() if j.get('roofline') else None)
    print({k:(v['ms_per_step'],v['launches_per_step']) for k,v in j['roofline']['groups'].items() if isinstance(v,dict) and 'ms_per_step' in v})
"
