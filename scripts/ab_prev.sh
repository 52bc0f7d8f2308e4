# A/B of the working tree against the previous commit's kernels: build the previous commit's diagnostic pair into moss_amd/lib_prev/
# first (git stash; python -m moss_amd.build --diag; cp -r moss_amd/lib_diag moss_amd/lib_prev; git stash pop; rebuild), then
#   bash scripts/ab_prev.sh [rounds]
# alternates the two libraries on the same box (bench.py --steps 200, no callers) and prints value, ms/step and per-stage times.
rounds=${1:-2}
for r in $(seq 1 $rounds); do
  for lib in ${AB_LIBS:-lib_prev lib_diag}; do
    env MOSS_AMD_LIB_DIR=$lib python bench.py --no-callers --no-cpu-baseline --steps 200 > gpurun_out/ab_${lib}_$r.json 2> gpurun_out/ab_${lib}_$r.err
    python - "$lib" "$r" <<'PY'
import json, sys
lab, r = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/ab_{lab}_{r}.json"))
    st = d["stages_ms"]
    print(f"{lab:9s} {d['value']:8.1f} it/s {d['ms_per_step']*1000:7.1f} us | " + " ".join(f"{k}={v*1000:.1f}" for k, v in st.items()) + f" | sum={sum(st.values())*1000:.1f}", flush=True)
except Exception as e:
    print(lab, "FAILED", e, open(f"gpurun_out/ab_{lab}_{r}.err").read()[-800:])
PY
  done
done
