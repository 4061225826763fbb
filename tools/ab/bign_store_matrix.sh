#!/bin/bash
# Store policy of the split-schedule output loop against the KIND of step (a step that resamples gathers duplicated sources, a step that
# does not reads every index once): C4 share and a single filter of 1.6e7 particles at thresholds 1.0 (every step resamples) and 0.1 (the
# reference's default: 3 of 100 steps resample).   tools/ab/bign_store_matrix.sh lib1 lib2 ...   ("default" = the product)
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = default ]; then unset LLPF_LIB; else [ -f lib_$lib.so ] || continue; export LLPF_LIB=$PWD/lib_$lib.so; fi
  echo "== $lib rep $rep"
  echo "   bank thr 1.0: $(python tools/bench_bank.py --thr 1.0 | grep -E -o '"us_per_timestep": [0-9.]+' | head -1)"
  echo "   bank thr 0.1: $(python tools/bench_bank.py | grep -E -o '"us_per_timestep": [0-9.]+' | head -1)"
  for thr in 0.1 1.0; do
  echo "   single 1.6e7 thr $thr: $(python bench.py --particles 16000000 --T 100 --threshold $thr --no-cpu-baseline --no-other-configs --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['whole_timestep']['us'],1), d['config'].get('resamples_per_pass'))")"
  done
done; done
