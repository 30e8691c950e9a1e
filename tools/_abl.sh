for m in "" _ablate1 _ablate4 _ablate65; do
  echo "== lib$m"
  YH_LIBRARY=/root/repo/yoloseries_amd/libyolohip$m.so python tools/bench_algos.py v5s fwd 2>&1 | grep -E "s3_b_3x3|s2_b_3x3|s4_b_3x3" | sed 's/TFLOP.s  v2.*halo/halo/'
done
