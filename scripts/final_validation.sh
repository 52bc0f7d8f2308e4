#!/bin/bash
# The end-of-round batch on the GPU box: GPU tests, 2 000 fuzz cases, 1 000 k-NN cases, 3 000 training steps, the 1M-Gaussian run.
# Every step streams into a file under gpurun_out/ (a long silent run is taken for a hang); the summary goes to stdout at the end.
mkdir -p gpurun_out
O=gpurun_out/final_validation
mkdir -p $O
timeout 900 python -u -m pytest tests -m gpu -q > $O/pytest.log 2>&1
timeout 2400 python -u scripts/fuzz_parity.py ${1:-2000} ${2:-2000} > $O/fuzz_parity.log 2>&1
timeout 1200 python -u scripts/fuzz_knn.py 1000 > $O/fuzz_knn.log 2>&1
timeout 900 python -u scripts/long_run.py 3000 01 > $O/long_run.log 2>&1
timeout 600 python -u scripts/stress_large.py > $O/stress_large.log 2>&1
tail -1 $O/pytest.log; grep -v amdgpu.ids $O/fuzz_parity.log | tail -40; tail -2 $O/fuzz_knn.log; tail -3 $O/long_run.log; tail -2 $O/stress_large.log
