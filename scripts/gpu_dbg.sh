#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --target smooth --warmup 600 2>&1 | tail -1 | cut -c1-200
timeout 300 python scripts/dbg_determinism.py 2>&1 | tail -1
