#!/bin/bash
# round 5, k_norm: prologue from preloaded SGPRs (lib_normA) and blocks that walk several tiles (product build; LLPF_NORM_TPB=1 pins one
# tile per block) against the round-4 kernel (lib_base).  Parity first, then the bank workload under rocprofv3 --kernel-trace per variant.
O=gpurun_out/r05n; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
run() {  # name lib [env]
  name=$1; lib=$2; shift 2
  for rep in 1 2; do
    env LLPF_LIB=$lib "$@" timeout 300 python $R/bench.py --workload bank --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name rep$rep %.4e part-steps/s  %.2f us/timestep' % (d['value'], d['roofline']['whole_timestep']['us']))" >> $R/$O/bank_ab.txt
  done
  env LLPF_LIB=$lib "$@" timeout 300 rocprofv3 --kernel-trace -d $R/$O/kt_$name -o kt -- python $R/bench.py --workload bank --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/kt_$name.log 2>&1
  python $R/tools/rocprof_summary.py $(find $R/$O/kt_$name -name "*.db" | head -1) | grep "k_norm\|k_resprop<.*true, false" | cut -c1-60,100-175 | sed "s/^/$name /" >> $R/$O/bank_kernels.txt
  rm -rf $R/$O/kt_$name
}
run base $R/lib_base.so
run prologue $R/lib_normA.so
run tiles1 $R/lowlevelparticlefilters.jl_amd/libllpf_hip.so LLPF_NORM_TPB=1
run tiles $R/lowlevelparticlefilters.jl_amd/libllpf_hip.so
run tiles5 $R/lowlevelparticlefilters.jl_amd/libllpf_hip.so LLPF_NORM_TPB=5
run tiles14 $R/lowlevelparticlefilters.jl_amd/libllpf_hip.so LLPF_NORM_TPB=14
cd $R
cat $O/bank_ab.txt $O/bank_kernels.txt
