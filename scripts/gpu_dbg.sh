#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -5 gpurun_out/t.log
timeout 600 python scripts/bench_densify.py 2>&1 | tail -3
