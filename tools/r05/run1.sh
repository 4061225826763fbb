#!/bin/bash
# round 5, GPU lease 1: suite on the new build, driver-style bench, k_norm one-wave-per-tile A/B, exp scaling through v_ldexp_f64 A/B
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
for rep in 1 2; do for v in 0 1; do
  echo "norm_wave_tile=$v rep$rep $(LLPF_NORM_WAVE_TILE=$v python tools/bench_bank.py 2>/dev/null | tail -1)" >> $O/norm_ab.txt
done; done
LLPF_LIB=$PWD/lib_ldexp.so python -m pytest tests/test_gpu_parity.py -x -q -k "device_math or c1_trajectory or c2_full" > $O/ldexp_tests.log 2>&1
tools/ab/all_libs.sh lib_ldexp.so > $O/ldexp_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  LLPF_NORM_WAVE_TILE=$v rocprofv3 --kernel-trace -d $R/$O/kt_bank$v -o kt -- python $R/tools/bench_bank.py > $R/$O/kt_bank$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $R/$O/kt_bank$v -name "*.db" | head -1) > $R/$O/kernel_stats_bank_wt$v.txt
  rm -rf $R/$O/kt_bank$v
done
ls -la $R/$O
