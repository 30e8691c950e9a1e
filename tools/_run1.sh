python3 -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" 2>&1 | tail -2
export YH_TUNE_CACHE=$PWD/gpurun_out/tc_c.json; rm -f $YH_TUNE_CACHE
python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
python3 -c "
import json; t=json.load(open('$YH_TUNE_CACHE')); print({k[:60]:v for k,v in t.items() if k.startswith('wgrad') and v[1]==40})"
bash tools/_run_tune.sh
