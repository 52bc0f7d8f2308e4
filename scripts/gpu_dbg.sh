for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['stages_ms'], d['roofline'], d['config']['num_rendered'])"; done
timeout 900 python bench.py --no-cpu-baseline --graph 0 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'], d['stages_ms'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
