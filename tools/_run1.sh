set -e
run() { YH_LIBRARY=$1 YH_H80_DBG=$2 BA_BATCH=128 BA_ONLY=s1_b_3x3 python3 tools/bench_algos.py v5x1280 eval 10 2>&1 | tail -1 | sed 's/.*h80/h80/'; }
L=$PWD/yoloseries_amd
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo80" 2>&1 | tail -3
for i in 1 2; do
echo "split dbg0"; run $L/libyolohip.so 0
echo "nosplit dbg0"; run $L/libyolohip_h80ns.so 0
done
echo "split dbg2"; run $L/libyolohip.so 2
echo "split dbg4"; run $L/libyolohip.so 4
echo "split dbg7"; run $L/libyolohip.so 7
