timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
timeout 600 python scripts/stage_times.py --mode scale_rot 2>&1 | tail -1 | cut -c1-300
