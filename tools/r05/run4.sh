#!/bin/bash
# round 5, GPU lease 4: where the shared tail of k_rbfull stands — old (one wave per workgroup) against new kernel on ONE box, then the stamps
O=gpurun_out/r05d; mkdir -p $O
run() { # lib particles tail
  if [ "$3" = "-" ] || [ "$3" = "auto" ]; then unset LLPF_RBF_TAIL; else export LLPF_RBF_TAIL=$3; fi
  if [ "$1" = "old" ]; then export LLPF_LIB=$PWD/lib_oldrbf.so; else unset LLPF_LIB; fi
  timeout 300 python bench.py --workload rbpf_full --particles $2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f k_rbfull %.2f' % (d['value'], d['roofline']['whole_timestep']['us'], d['roofline']['avg_launch_us']))"
}
for rep in 1 2; do
  for c in "old 196608 -" "new 196608 0" "old 200000 -" "new 200000 0" "new 200000 auto" "new 196672 0" "new 196672 1" "new 200000 20" "new 200000 53"; do
    echo "$c rep$rep $(run $c)" >> $O/c5_ab.txt
  done
done
unset LLPF_RBF_TAIL
export LLPF_LIB=$PWD/lib_timing.so
timeout 300 python tools/dbg/rbf_tail_timing.py 200000 > $O/tail_timing_200000.txt 2>&1
LLPF_RBF_TAIL=1 timeout 300 python tools/dbg/rbf_tail_timing.py 196672 > $O/tail_timing_196672_tail1.txt 2>&1
LLPF_RBF_TAIL=0 timeout 300 python tools/dbg/rbf_tail_timing.py 200000 > $O/tail_timing_200000_tail0.txt 2>&1
ls -la $O
