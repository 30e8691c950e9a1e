#!/bin/bash
# FETCH_SIZE of every conv_wgs_kernel launch of tools/bench_wgs.py on one layer, in launch order (warm-up of each G, then 3 timed of each G)
export TMPDIR=/tmp
L=${1:-l3_b_3x3}
OUT=gpurun_out/wgs_fetch
rm -rf $OUT; mkdir -p $OUT
WG_ONLY=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p -- python3 tools/bench_wgs.py 1 3 > $OUT/log.txt 2>&1
F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_wgs_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    print(r["Dispatch_Id"], r["Grid_Size"], r["Workgroup_Size"], f'{2*1024*float(r["Counter_Value"])/1e6:9.1f} MB (x2 corrected)')
PY
cat $OUT/log.txt | tail -3
rm -rf $OUT/p
