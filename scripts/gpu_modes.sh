mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25
for m in "--forward sync --graph 0" "--forward async --graph 0" "--forward async --graph 1"; do
  for r in 1 2; do
    timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline $m 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$m', d['value'], d['ms_per_step'], d['rasterizer_ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['config']['launch'])
except Exception as e:
    print('FAIL $m', l[-600:])
"
  done
done 2>&1 | tee gpurun_out/modes.log
