# The configurations beside the headline one, for profiles/<tag>_other_configs.log: BASELINE configs[1] and configs[4] through bench.py,
# the `--target smooth` stress case after 600 steps, and a 3000-step run.  usage: bash scripts/other_configs.sh > gpurun_out/other.log
for c in cfg2 cfg5; do
  timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline --no-callers --steps 100 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$c', r['value'], 'it/s', {k: round(v*1000,1) for k,v in r['stages_ms'].items()})"
done
timeout -k 10 300 python3 bench.py --target smooth --no-cpu-baseline --no-callers --steps 200 --warmup 600 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('smooth after 600', r['value'], 'it/s', {k: round(v*1000,1) for k,v in r['stages_ms'].items()})"
timeout -k 10 300 python3 scripts/long_run.py 3000 2>&1 | tail -n 8
