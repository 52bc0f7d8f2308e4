export TMPDIR=/tmp
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'ms',d['ms_per_step'],'raster',d['rasterizer_ms_per_step'])"; done
OUT=$PWD/gpurun_out/prof_r1c; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (55+50 profiled steps -> /105):", tot/1e6/105, "launches per step:", sum(int(r["Calls"]) for r in rows)/105)
for r in rows[:22]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "tot_ms=%7.2f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
tail -1 $OUT/bench_stdout.log | cut -c1-200
