# The scalar-unit question of the forward blend: two counter-only passes over the eager form of the bench step.
# usage: bash scripts/pmc_scalar.sh   -> gpurun_out/pmc_scalar/summary.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_scalar
rm -rf $OUT; mkdir -p $OUT; cd /tmp
pmc() { n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $n -- python3 $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 30 --warmup 10 --no-cpu-baseline --no-callers > $OUT/$n.log 2>&1
  echo "pmc pass $n done"
}
pmc p5 SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc p6 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_pmc.py $OUT $OUT/summary.json > $OUT/summary.txt 2>&1
rm -f $OUT/*counter_collection.csv
grep "blend_" $OUT/summary.txt
