#!/bin/bash
# Is the shipped launch-parameter table (yoloseries_amd/tune_defaults.json) a cliff off the judged shape?  
# Same box, interleaved: the judged workload and one held-out shape per training workload (batch 48, 512 x 512 — none of its layer
# shapes are in the table) with (a) the shipped table + what this machine times for unknown shapes, (b) a table timed from scratch
# on this machine (YH_TUNE_DEFAULTS=0).  Three runs each, medians.  Run on an MI355X from the repo root.
NOB="--no-cpu-baseline --no-roofline --steps 20 --warmup 4"
val() { python3 bench.py "$@" $NOB 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['tuning']['timed_now'])"; }
run3() {  # label, env prefix, args...
  local label=$1; shift; local envs=$1; shift
  local vals=""
  for i in 1 2 3; do vals="$vals $(env $envs bash -c "$(declare -f val); NOB='$NOB'; val $*" | cut -d' ' -f1)"; done
  python3 -c "import sys; v=sorted(float(x) for x in sys.argv[2:]); print(f'{sys.argv[1]:58s} median {v[1]:8.1f} img/s   (runs {v})')" "$label" $vals
}
rm -f /tmp/tv_a.json /tmp/tv_b.json
for wl in "" "--model large" "--workload yolox"; do
  for shape in "" "--batch 48 --img 512"; do
    # first run of each arm times what its table lacks; the three measured runs then start from the same kind of cache
    YH_TUNE_CACHE=/tmp/tv_a.json python3 bench.py $wl $shape $NOB > /dev/null 2>&1
    YH_TUNE_CACHE=/tmp/tv_b.json YH_TUNE_DEFAULTS=0 python3 bench.py $wl $shape $NOB > /dev/null 2>&1
    run3 "shipped table      [$wl $shape]" "YH_TUNE_CACHE=/tmp/tv_a.json" $wl $shape
    run3 "timed from scratch [$wl $shape]" "YH_TUNE_CACHE=/tmp/tv_b.json YH_TUNE_DEFAULTS=0" $wl $shape
  done
done
