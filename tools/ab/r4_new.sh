mkdir -p gpurun_out/r4_new
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged_and_degenerate or tie or forms or chooses" > gpurun_out/r4_new/new_tests.log 2>&1; tail -15 gpurun_out/r4_new/new_tests.log
timeout 600 python tools/bench_nx.py > gpurun_out/r4_new/bench_state_dimension.json 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r4_new/bench_state_dimension.json'))
for r in d['rows']: print(r.get('nx'), r.get('schedule'), r.get('us_per_timestep'), r.get('whole_timestep_roofline_frac'), r.get('error'))"
