# A/B of kernel variants through the DIAGNOSTIC build: bash scripts/ab_bench.sh "<label> ENV=.. ENV=.." ...   (one bench run per argument)
# prints value, ms/step and the per-stage kernel times of each; results under gpurun_out/ab_<label>.json
for spec in "$@"; do
  set -- $spec; label=$1; shift
  env MOSS_AMD_LIB_DIR=lib_diag "$@" python bench.py --no-callers --no-cpu-baseline --steps 200 > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{lab}.json"))
    st = d["stages_ms"]
    print(f"{lab:14s} {d['value']:8.1f} it/s {d['ms_per_step']*1000:7.1f} us | " + " ".join(f"{k}={v*1000:.1f}" for k, v in st.items()) + f" | sum={sum(st.values())*1000:.1f}")
except Exception as e:
    print(lab, "FAILED", e, open(f"gpurun_out/ab_{lab}.err").read()[-800:])
PY
done
