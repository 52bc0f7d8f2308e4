timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
for v in 1 0; do echo "PREFWD_STAGE=$v"; MOSS_PREFWD_STAGE=$v timeout 600 python scripts/stage_times.py --mode scale_rot 2>&1 | tail -1 | cut -c1-200; done
