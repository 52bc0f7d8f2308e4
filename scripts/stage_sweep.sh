# usage: bash scripts/stage_sweep.sh "SEG_HITS WGS" ...   -- bench stage times for (MOSS_SEG_HITS, MOSS_BWD_WGS_PER_CU) settings
for cfg in "${@:-64 3}"; do set -- $cfg; MOSS_SEG_HITS=$1 MOSS_BWD_WGS_PER_CU=$2 python bench.py --no-cpu-baseline --no-callers --steps 100 > gpurun_out/tmp_b.json 2>> gpurun_out/tmp_b.err; python -c "
import json; r=json.load(open('gpurun_out/tmp_b.json')); print('seg $1 wgs $2', r['value'], {k: round(v*1000,1) for k,v in r['stages_ms'].items()})"; done
