timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/stage_times.py --mode scale_rot 2>&1 | tail -1 | cut -c1-300
timeout 300 python scripts/wave_stamps.py 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['stages_ms'])"
