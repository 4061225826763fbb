#!/bin/bash
# round 5, GPU lease 5: the shared batches done FIRST in their workgroups — parity, then the same-box A/B against the old kernel
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rbfull.py -x -q > $O/rbfull_tests.log 2>&1; echo "rbfull tests rc=$?" >> $O/rbfull_tests.log
run() { # lib particles tail
  if [ "$3" = "-" ] || [ "$3" = "auto" ]; then unset LLPF_RBF_TAIL; else export LLPF_RBF_TAIL=$3; fi
  if [ "$1" = "old" ]; then export LLPF_LIB=$PWD/lib_oldrbf.so; else unset LLPF_LIB; fi
  timeout 300 python bench.py --workload rbpf_full --particles $2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f k_rbfull %.2f' % (d['value'], d['roofline']['whole_timestep']['us'], d['roofline']['avg_launch_us']))"
}
for rep in 1 2; do
  for c in "old 196608 -" "new 196608 0" "old 200000 -" "new 200000 0" "new 200000 auto" "new 196672 1" "new 200000 20" "new 204800 auto" "old 204800 -"; do
    echo "$c rep$rep $(run $c)" >> $O/c5_ab.txt
  done
done
unset LLPF_RBF_TAIL
export LLPF_LIB=$PWD/lib_timing.so
timeout 300 python tools/dbg/rbf_tail_timing.py 200000 > $O/tail_timing_200000.txt 2>&1
ls -la $O
