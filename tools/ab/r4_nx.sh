set -u
OUT=$PWD/gpurun_out/r04a; mkdir -p $OUT
timeout 900 python tools/bench_nx.py > $OUT/bench_state_dimension.json 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04a/bench_state_dimension.json"))
for r in d["rows"]: print(r.get("nx"), r.get("schedule"), r.get("us_per_timestep"), r.get("whole_timestep_roofline_frac"), r.get("loglik"), r.get("error","")[:200])
PY

