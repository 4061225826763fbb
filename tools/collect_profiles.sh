#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      e.g. r01d
# Writes everything under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
#   1. bench lines (C2 default, C2 with the reference's default threshold 0.1, C3 quad-tank, one-GPU share of C4)
#   2. rocprofv3 --kernel-trace of the default bench command, summarised per kernel (tools/rocprof_summary.py)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (never combined with other traces), per-kernel
#      averages per dispatch (tools/rocprof_pmc_summary.py)
set -u
TAG=${1:-prof}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --no-other-configs > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --threshold 0.1 --no-cpu-baseline > $OUT/bench_c2_thr01.json 2>> $OUT/bench_c2.err
python bench.py --workload quadtank > $OUT/bench_c3_quadtank.json 2>> $OUT/bench_c2.err
python tools/bench_bank.py > $OUT/bench_c4_bank_128x1e5_one_gpu.json 2>> $OUT/bench_c2.err
python tools/bench_bank.py --thr 1.0 > $OUT/bench_c4_bank_128x1e5_thr1.json 2>> $OUT/bench_c2.err
python bench.py --workload bank --steps 2 > $OUT/bench_c4_bank_workload.json 2>> $OUT/bench_c2.err
python bench.py --gpus 2 --dist-backend gloo --steps 2 --no-cpu-baseline | grep '^{' > $OUT/bench_c4_two_ranks_one_gpu_gloo.json 2>> $OUT/bench_c2.err
python bench.py --workload aux > $OUT/bench_aux.json 2>> $OUT/bench_c2.err
python bench.py --workload rbpf > $OUT/bench_rbpf.json 2>> $OUT/bench_c2.err
python bench.py --workload rbpf_full > $OUT/bench_c5_rbpf_full.json 2>> $OUT/bench_c2.err
python tools/bench_smooth.py > $OUT/bench_ffbs_smoother.json 2>> $OUT/bench_c2.err
python tools/bench_smooth.py --particles 1000 --T 200 --M 100 --cpu-M 100 >> $OUT/bench_ffbs_smoother.json 2>> $OUT/bench_c2.err
python tools/bench_mc.py > $OUT/bench_reference_mc_run_test.json 2>> $OUT/bench_c2.err
python tools/bench_nx.py > $OUT/bench_state_dimension.json 2>> $OUT/bench_c2.err
python tools/bench_n.py > $OUT/bench_particle_count.json 2>> $OUT/bench_c2.err
python bench.py --particles 16000000 --T 100 --cpu-quick > $OUT/bench_c2_n1.6e7.json 2>> $OUT/bench_c2.err
tools/ab/schedules.sh > $OUT/schedules_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_c2_big -o kt -- python $ROOT/bench.py --particles 16000000 --T 100 --steps 3 --no-cpu-baseline > $OUT/kt_c2_big.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_bank -o kt -- python $ROOT/tools/bench_bank.py > $OUT/kt_bank.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_qt -o kt -- python $ROOT/bench.py --workload quadtank --steps 2 --no-cpu-baseline > $OUT/kt_qt.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_aux -o kt -- python $ROOT/bench.py --workload aux --no-cpu-baseline > $OUT/kt_aux.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_rbpf -o kt -- python $ROOT/bench.py --workload rbpf --no-cpu-baseline > $OUT/kt_rbpf.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt_rbpf_full -o kt -- python $ROOT/bench.py --workload rbpf_full --no-cpu-baseline > $OUT/kt_rbpf_full.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_c5 -o p -- python $ROOT/bench.py --workload rbpf_full --steps 1 --warmup 0 --T 200 --no-cpu-baseline > $OUT/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_c5 -o p -- python $ROOT/bench.py --workload rbpf_full --steps 1 --warmup 0 --T 200 --no-cpu-baseline > $OUT/pmc_write_c5.log 2>&1
for w in quadtank bank; do for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c -d $OUT/pmc_${c}_$w -o p -- python $ROOT/bench.py --workload $w --steps 1 --warmup 0 --T 100 --no-cpu-baseline > $OUT/pmc_${c}_$w.log 2>&1
done; done
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c -d $OUT/pmc_${c}_c2_big -o p -- python $ROOT/bench.py --particles 16000000 --steps 1 --warmup 0 --T 20 --no-cpu-baseline > $OUT/pmc_${c}_c2_big.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --T 200 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --T 200 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --T 200 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
cd $ROOT
python tools/rocprof_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
python tools/rocprof_summary.py $(find $OUT/kt_bank -name "*.db" | head -1) > $OUT/kernel_stats_bank.txt
for w in c2_big qt aux rbpf rbpf_full; do python tools/rocprof_summary.py $(find $OUT/kt_$w -name "*.db" | head -1) > $OUT/kernel_stats_$w.txt; done
python tools/rocprof_pmc_summary.py $OUT/pmc_traffic.txt $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) $(find $OUT/pmc_sq -name "*.db" | head -1)
python tools/rocprof_pmc_summary.py $OUT/pmc_traffic_c5.txt $(find $OUT/pmc_fetch_c5 -name "*.db" | head -1) $(find $OUT/pmc_write_c5 -name "*.db" | head -1)
python tools/rocprof_pmc_summary.py $OUT/pmc_traffic_c2_big.txt $(find $OUT/pmc_FETCH_SIZE_c2_big -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE_c2_big -name "*.db" | head -1); rm -rf $OUT/pmc_FETCH_SIZE_c2_big $OUT/pmc_WRITE_SIZE_c2_big
for w in quadtank bank; do python tools/rocprof_pmc_summary.py $OUT/pmc_traffic_$w.txt $(find $OUT/pmc_FETCH_SIZE_$w -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE_$w -name "*.db" | head -1); rm -rf $OUT/pmc_FETCH_SIZE_$w $OUT/pmc_WRITE_SIZE_$w; done
python tools/make_pmc_json.py $OUT $OUT/pmc_traffic.json $TAG
rm -rf $OUT/kt_c2_big $OUT/kt_rbpf_full $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 $OUT/kt $OUT/kt_bank $OUT/kt_qt $OUT/kt_aux $OUT/kt_rbpf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/*.log
bash tools/dbg/pmc_sq_other.sh $TAG > /dev/null 2>&1      # SQ counters of the C3 / C5 kernels: pmc_sq_quadtank.txt, pmc_sq_rbpf_full.txt
ls -la $OUT
