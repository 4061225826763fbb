#!/bin/bash
# SQ counters of the C4 share (128 filters x 1e5, threshold 0.1) on the final sources: separate rocprofv3 --pmc pass, no trace domain.
#   tools/dbg/pmc_sq_bank.sh <tag>   ->  gpurun_out/<tag>/pmc_sq_bank.txt
set -u
TAG=${1:-sqbank}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload bank --steps 1 --warmup 0 --T 100 --no-cpu-baseline --no-other-configs"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
cd $ROOT
for p in p1 p2; do
  db=$(find $OUT/$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_pmc_summary.py $OUT/pmc_sq_bank_$p.txt $db; else echo "no db for $p"; tail -5 $OUT/$p.log; fi
  rm -rf $OUT/$p
done
cat $OUT/pmc_sq_bank_p1.txt $OUT/pmc_sq_bank_p2.txt | grep -E "k_resprop|k_norm|##" | cut -c1-40,70-220 > $OUT/pmc_sq_bank.txt
cat $OUT/pmc_sq_bank.txt
