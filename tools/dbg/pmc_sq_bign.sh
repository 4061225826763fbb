#!/bin/bash
# SQ / TCC counters of the fused launch of a filter beyond the Infinity Cache (C2 system, N = 1.6e7): separate rocprofv3 --pmc passes,
# never combined with a trace domain.   tools/dbg/pmc_sq_bign.sh <tag>   ->  gpurun_out/<tag>/pmc_*_c2_big.txt
set -u
TAG=${1:-sqbig}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --particles 16000000 --steps 1 --warmup 0 --T 20 --no-cpu-baseline --no-other-configs"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_WAVES_EQ_64 -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum -d $OUT/p4 -o p -- $CMD > $OUT/p4.log 2>&1
cd $ROOT
for p in p1 p2 p3 p4; do
  db=$(find $OUT/$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_pmc_summary.py $OUT/pmc_${p}_c2_big.txt $db; else echo "no db for $p"; tail -5 $OUT/$p.log; fi
  rm -rf $OUT/$p
done
grep -h "k_resprop\|k_norm\|##" $OUT/pmc_p*_c2_big.txt | cut -c1-40,70-220
