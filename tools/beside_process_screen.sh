#!/bin/bash
# the bit-exact launch-to-launch screen of every conv entry of the shipped table (tools/race_screen.py) while ANOTHER PROCESS
# trains YOLOv5s on the same GPU: do the packed-fp32 forms of the convolution epilogues share the fault of DESIGN §5 (l)?
export TMPDIR=/tmp
O=gpurun_out/beside; mkdir -p $O
python tools/loss_race_diag.py 1 ${2:-3000} model_full_load > $O/load.log 2>&1 &
LOAD=$!
sleep 20
python tools/race_screen.py ${1:-12} > $O/screen.log 2>&1; echo "screen rc $?"
tail -4 $O/screen.log | cut -c1-300
kill -0 $LOAD 2>/dev/null && echo "load still running at the end of the screen (good)" || echo "LOAD ENDED BEFORE THE SCREEN"
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
exit 0
