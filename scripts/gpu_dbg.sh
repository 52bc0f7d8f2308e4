timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['rasterizer_ms_per_step'])"; done
