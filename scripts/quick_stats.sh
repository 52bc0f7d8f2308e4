# tests of one keyword + the default bench + rocprofv3 kernel statistics of a 100-step bench: bash scripts/quick_stats.sh <tag> [pytest -k expression]
TAG=${1:-q}; KEXPR=${2:-}
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "$KEXPR" > gpurun_out/t_$TAG.log 2>&1; tail -4 gpurun_out/t_$TAG.log; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-callers > gpurun_out/b_$TAG.json 2> gpurun_out/b_$TAG.err; tail -c 300 gpurun_out/b_$TAG.err
python -c "
import json; d=json.load(open('gpurun_out/b_$TAG.json')); print('VALUE', d['value'], d['ms_per_step'], d.get('long_run'))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-callers --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
python3 -c "
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/**/*kernel_stats.csv', recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f)))[:11]:
    print(r['Name'].replace('moss::(anonymous namespace)::','')[:44].ljust(46), r['Calls'].rjust(5), round(float(r['AverageNs'])/1000,1)); 
"
