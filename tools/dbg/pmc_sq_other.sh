#!/bin/bash
# SQ counters (waves, busy cycles, VALU / SALU instructions, active-instruction cycles) of the dominant kernels of C3 and C5,
# one rocprofv3 --pmc pass each (never combined with trace domains):  tools/dbg/pmc_sq_other.sh <tag>  ->  gpurun_out/<tag>/pmc_sq_{qt,c5}.txt
set -u
TAG=${1:-sq}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in quadtank rbpf_full; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq_$w -o p -- python $ROOT/bench.py --workload $w --steps 1 --warmup 0 --T 100 --no-cpu-baseline > $OUT/pmc_sq_$w.log 2>&1
done
cd $ROOT
for w in quadtank rbpf_full; do
  python tools/rocprof_pmc_summary.py $OUT/pmc_sq_$w.txt $(find $OUT/pmc_sq_$w -name "*.db" | head -1)
  rm -rf $OUT/pmc_sq_$w $OUT/pmc_sq_$w.log
done
ls -la $OUT
