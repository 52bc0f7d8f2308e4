timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "more_tiles or ragged" 2>&1 | tail -3
