# Round profile set: (1) rocprofv3 kernel trace + stats of the default bench command, (2) PMC passes (separate runs, counters only)
# over the eager-launch form of the same bench step.  usage: bash scripts/gpu_profile_round.sh <tag>   (e.g. r02_final)
# Outputs under gpurun_out/<tag>/ ; copy the summaries into profiles/ (pmc_stage_summary.json -> profiles/pmc_latest.json: it is
# stamped with the sha256 of moss_amd/csrc/, and bench.py reports `traffic` only while that stamp matches the checkout).
export TMPDIR=/tmp
TAG=${1:-r03_final}
OUT=$PWD/gpurun_out/$TAG
if [ -z "$PMC_ONLY" ]; then rm -rf $OUT; fi
mkdir -p $OUT/pmc $OUT/markers; cd /tmp
if [ -z "$PMC_ONLY" ]; then
# the bench line as the driver sees it (no profiler attached), then the same command under rocprofv3 (whose tool perturbs the
# bench's own kernel-attached events by ~10 %: its line is kept beside the kernel statistics for reference only)
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_plain_stdout.log 2>&1
grep "^{\"metric\"" $OUT/bench_plain_stdout.log > $OUT/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_stdout.log 2>&1
grep "^{\"metric\"" $OUT/bench_stdout.log > $OUT/bench_line_under_rocprof.json
fi
pmc() { n=$1; shift
  # (--no-callers: the drop-in caller variants are thousands of torch launches, every one serialised under --pmc -- minutes of silence)
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc -o $n -- python3 $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 30 --warmup 10 --no-cpu-baseline --no-callers > $OUT/pmc/$n.log 2>&1
  echo "pmc pass $n done"
}
pmc p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc p2 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pmc p3 FETCH_SIZE
pmc p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# the scalar unit beside the vector unit (round 4's review, item 5: the forward blend executes as many SALU as VALU instructions)
pmc p5 SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc p6 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS
# the stage ranges (MOSS_DEBUG_TRACE: roctx ranges around every launch of the op, include/moss_raster.h) beside the kernels they hold:
# eager launches (ranges are host-side: a graph replay has none), no counters in this pass
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/markers -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --graph 0 --debug-bits 8 --steps 30 --warmup 10 --no-cpu-baseline --no-callers > $OUT/markers/trace.log 2>&1
rm -f $OUT/markers/trace_kernel_trace.csv
echo "marker pass done"
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_pmc.py $OUT/pmc $OUT/pmc_summary.json $OUT/pmc_stage_summary.json > $OUT/pmc_hbm_bytes.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms:", tot/1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:24]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), "avg_us=%8.1f"%(float(r["AverageNs"])/1e3), "tot_ms=%7.2f"%(float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
cat $OUT/pmc_hbm_bytes.txt | tail -20
cut -c1-400 $OUT/bench_line.json
rm -f $OUT/bench_kernel_trace.csv $OUT/pmc/*counter_collection.csv.bak $OUT/pmc/*counter_collection.csv   # (raw counter dumps are hundreds of MB: gpurun merges <= 64 MiB back)
ls -la $OUT $OUT/pmc | head -40
