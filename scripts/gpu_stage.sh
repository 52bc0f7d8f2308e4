timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
for c in cfg3 cfg2 cfg5; do python scripts/stage_times.py --config $c 2>&1 | tail -1; done
