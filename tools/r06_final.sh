#!/bin/bash
# round 6: the GPU suite and the committed profile set, one box
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $O/gputest.log 2>&1; echo "gputest rc $?" | tee $O/gputest.rc
tail -18 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee $O/smoke.rc; tail -1 $O/smoke.log
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; echo "profile rc $?" | tee $O/profile.rc
tail -30 $O/profile_round.log
python3 tools/ceiling.py gpurun_out/prof_r06/profiles/r06_layers_v5s_train_b64.txt 64 > $O/ceiling.txt 2>&1
python3 tools/ceiling.py gpurun_out/prof_r06/profiles/r06_layers_v5l_train_b64.txt 64 >> $O/ceiling.txt 2>&1
python3 tools/ceiling.py gpurun_out/prof_r06/profiles/r06_layers_infer_v5x_1280_b128.txt 128 >> $O/ceiling.txt 2>&1
cat $O/ceiling.txt
