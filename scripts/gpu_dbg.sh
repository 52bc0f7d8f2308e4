timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['stages_ms'])"
timeout 900 python bench.py --no-cpu-baseline --steps 300 --warmup 200 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('evolved(200+300)', d['value'], d['ms_per_step'], d['stages_ms'])"
timeout 600 python scripts/dbg_determinism.py 2>&1 | tail -1
