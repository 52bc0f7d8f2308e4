export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
OUT=$PWD/gpurun_out/prof_r1b; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (55+50 profiled steps -> /105):", tot/1e6/105)
for r in rows[:32]:
    print(r["Name"][:95].ljust(95), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "tot_ms=%7.2f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
