#!/bin/bash
# usage: tools/sweep_env.sh <out.log> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   — one bench line (img/s) per environment setting
OUT=$1; shift
ARGS=$1; shift
mkdir -p $(dirname $OUT)
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py $ARGS --no-cpu-baseline --no-roofline 2>>$OUT.err | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")
  echo "$cfg -> $v" | tee -a $OUT
done
