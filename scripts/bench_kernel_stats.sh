# rocprofv3 per-kernel statistics of the default bench step (graph replay): usage bash scripts/bench_kernel_stats.sh [extra bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/bstats; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bs -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-callers "$@" > $OUT/stdout.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/bs_kernel_stats.csv")))
tot=0
for r in rows[:18]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "min_us=%8.1f"%(float(r["MinNs"])/1e3))
PY
grep '^{"metric' $OUT/stdout.log | cut -c1-160
rm -f $OUT/bs_kernel_trace.csv
