mkdir -p gpurun_out/r4_q
timeout 900 python -m pytest tests/test_quantile.py tests/test_gpu_parity.py -x -q -m gpu -k "quantile or engine_against or tie or weighted_cov" > gpurun_out/r4_q/tests.log 2>&1; tail -15 gpurun_out/r4_q/tests.log
timeout 300 python - <<'PY'
import sys, time; sys.path[:0]=['.','tests']
import numpy as np, models as M
from llpf_amd import _capi, _structs as S
g=_capi.FilterHandle(S.make_config(M.lg_test_model(),1000000,resample_threshold=0.5,seed=1)); g.reset(); g.correct([0.1],[0.3],0.0)
g.weighted_quantile([0.5]); t0=time.perf_counter(); q=g.weighted_quantile([0.05,0.5,0.95]); print("N=1e6 nx=2 weighted_quantile ms", 1e3*(time.perf_counter()-t0), q)
PY
