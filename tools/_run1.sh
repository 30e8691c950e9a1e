python3 -m pytest tests -x -q -m gpu > gpurun_out/r3_full3.log 2>&1; tail -3 gpurun_out/r3_full3.log
python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
python3 bench.py --no-cpu-baseline --no-roofline --model large 2>/dev/null | cut -c1-140
python3 bench.py --no-cpu-baseline --no-roofline --workload yolox 2>/dev/null | cut -c1-140
