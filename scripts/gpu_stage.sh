timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
for k in 2 3; do echo "WGS_PER_CU=$k"; for c in cfg3 cfg2 cfg5; do MOSS_BLEND_WGS_PER_CU=$k python scripts/stage_times.py --config $c 2>&1 | tail -1; done; done
