set -x
mkdir -p gpurun_out
timeout 900 python bench.py --steps 100 --warmup 10 2>&1 | tail -5 | tee gpurun_out/bench_first.log
