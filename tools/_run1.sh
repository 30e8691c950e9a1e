python3 -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -4
python3 -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -2
