set -e
L=$PWD/yoloseries_amd
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo" 2>&1 | tail -3
for i in 1 2; do for lib in libyolohip.so libyolohip_noburst.so; do echo $lib; YH_LIBRARY=$L/$lib BA_BATCH=64 BA_ONLY=s2_b_3x3,s3_b_3x3,s4_b_3x3 python3 tools/bench_algos.py v5x1280 eval 10 2>&1 | grep eval | sed 's/.*halo /halo /;s/| dg2.*//'; done; done
