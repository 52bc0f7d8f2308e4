mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_latest.log
