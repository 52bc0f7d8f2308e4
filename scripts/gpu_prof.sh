export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r1d; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms:", tot/1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:60]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "tot_ms=%7.2f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
tail -1 $OUT/bench_stdout.log | cut -c1-250
