#!/bin/bash
mkdir -p gpurun_out
for t in body smooth; do for w in 20 600; do
echo "target=$t warmup=$w"; timeout 300 python bench.py --steps 200 --warmup $w --target $t --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/line.json; python -c "
import json,sys; d=json.load(open('gpurun_out/line.json')); print(d['value'], d['ms_per_step'], d['config']['num_rendered'], d['stages_ms'], d.get('densify_side_ms'))" || tail -5 gpurun_out/line.json
done; done
