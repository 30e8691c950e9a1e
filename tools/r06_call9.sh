#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06i; mkdir -p $O
python tools/wgs_stamps.py s4_cba3 20 512 512 1 1 192 > $O/stamps.txt 2>&1
python tools/wgs_stamps.py s4_cba3 20 512 512 1 1 256 >> $O/stamps.txt 2>&1
python tools/wgs_stamps.py s3_cba3 40 256 256 1 1 192 >> $O/stamps.txt 2>&1
python tools/wgs_stamps.py s2_cba3 80 128 128 1 1 192 >> $O/stamps.txt 2>&1
python tools/wgs_stamps.py s4_b 20 256 256 3 1 192 >> $O/stamps.txt 2>&1
cat $O/stamps.txt
