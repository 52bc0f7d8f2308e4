export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmc; mkdir -p $OUT; cd /tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $n -- python3 $GRAFT_REPO_ROOT/scripts/stage_times.py --config cfg3 --iters 5 > $OUT/$n.log 2>&1
}
run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run p2 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run p3 FETCH_SIZE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls $OUT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "moss" not in k: continue
        short = k.split("::")[-1].split("(")[0]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-1])
    for k, cs in agg.items():
        print(" ", k.ljust(28), {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
