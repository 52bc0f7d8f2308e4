timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/dbg_evolve.py 2>&1 | grep -v Warning | tail -11 | cut -c1-330
MOSS_BENCH_VERBOSE=1 timeout 900 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | grep -v Warn | cut -c1-900 | tail -3
