set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r1
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
ls -R $OUT | head -30
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $F"; head -45 "$F"
