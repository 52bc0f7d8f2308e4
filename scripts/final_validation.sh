#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1
timeout 2400 python scripts/fuzz_parity.py 2000 2000 2>&1 | tail -6
timeout 1200 python scripts/fuzz_knn.py 1000 2>&1 | tail -3
timeout 600 python scripts/long_run.py 3000 2>&1 | tail -3
timeout 600 python scripts/stress_large.py 2>&1 | tail -2
