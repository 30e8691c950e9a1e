export YH_BENCH_STAMP=1
for B in 64 32; do
(python3 bench.py --batch $B --no-cpu-baseline --no-roofline --steps 150 --warmup 10 2> gpurun_out/r3_dual_a.err | cut -c1-140 > gpurun_out/r3_dual_a.txt) &
python3 bench.py --batch $B --no-cpu-baseline --no-roofline --steps 150 --warmup 10 2> gpurun_out/r3_dual_b.err | cut -c1-140 > gpurun_out/r3_dual_b.txt
wait
echo "B=$B"; grep timed gpurun_out/r3_dual_a.err gpurun_out/r3_dual_b.err; cut -c60-140 gpurun_out/r3_dual_a.txt gpurun_out/r3_dual_b.txt
done
