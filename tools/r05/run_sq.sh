#!/bin/bash
# round 5: where the waves of the bank's two kernels (and of C2 / C3) spend their cycles — SQ counters, one rocprofv3 --pmc pass per
# workload (never combined with a trace domain).  WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) +
# ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles).     tools/r05/run_sq.sh  ->  gpurun_out/r05s/pmc_sq_<workload>.txt
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r05s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-bank quadtank lg}; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq_$w -o p -- python $ROOT/bench.py --workload $w --steps 1 --warmup 0 --T 100 --no-cpu-baseline > $OUT/pmc_sq_$w.log 2>&1
done
cd $ROOT
for w in ${WORKLOADS:-bank quadtank lg}; do
  python tools/rocprof_pmc_summary.py $OUT/pmc_sq_$w.txt $(find $OUT/pmc_sq_$w -name "*.db" | head -1)
  rm -rf $OUT/pmc_sq_$w
  grep -v "rocclr\|k_init\|k_post" $OUT/pmc_sq_$w.txt | cut -c1-60,70-200
done
