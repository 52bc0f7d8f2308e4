timeout 900 python bench.py --no-cpu-baseline --config cfg5 --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-1500
timeout 900 python bench.py --no-cpu-baseline --config cfg2 --steps 100 --warmup 10 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['stages_ms'])"
timeout 900 python bench.py --no-cpu-baseline --mode precomp --steps 100 --warmup 10 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('precomp', d['value'], d['ms_per_step'], d['stages_ms'])"
