# per-kernel rocprofv3 statistics of the raw op on one configuration (scripts/stage_times.py): usage bash scripts/kernel_stats.sh [cfg3] [scale_rot]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $GRAFT_REPO_ROOT/scripts/stage_times.py --config ${1:-cfg3} --mode ${2:-scale_rot} --iters 60 > $OUT/stdout.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/ks_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "min_us=%8.1f"%(float(r["MinNs"])/1e3))
PY
grep cfg $OUT/stdout.log | cut -c1-250
rm -f $OUT/ks_kernel_trace.csv
