#!/bin/bash
# PMC counters of conv_wgs_kernel on one layer shape, isolated (rocprofv3 --pmc, counters in their own passes): MFMA busy cycles against
# the chip's cycles, LDS instructions / bank conflicts, wave wait states, HBM traffic.  Run on an MI355X from the repo root.
# usage: tools/wgs_pmc.sh [layer name of tools/bench_wgs.py, default l3_b_3x3]
export TMPDIR=/tmp
L=${1:-l3_b_3x3}
OUT=gpurun_out/wgs_pmc
rm -rf $OUT; mkdir -p $OUT
pass() { WG_ONLY=$L rocprofv3 --pmc "$@" --output-format csv -d $OUT/p -- python3 tools/bench_wgs.py 1 3 > $OUT/log.txt 2>&1
         F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
         python3 - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_wgs_kernel" in r["Kernel_Name"]]
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = sorted(v)
    print(f"  {k:34s} median per launch {v[len(v)//2]:16.0f}   ({len(v)} launches)")
PY
         rm -rf $OUT/p; }
echo "conv_wgs_kernel on $L (tools/bench_wgs.py shapes, B = 64), rocprofv3 --pmc, one pass per line group:"
pass SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pass SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pass SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
pass FETCH_SIZE
pass WRITE_SIZE
