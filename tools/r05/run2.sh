#!/bin/bash
# round 5, GPU lease 2: the shared tail batches of k_rbfull — parity first, then C5 with and without them, then the whole suite
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rbfull.py -x -q > $O/rbfull_tests.log 2>&1; echo "rbfull tests rc=$?" >> $O/rbfull_tests.log
if grep -q "rc=0" $O/rbfull_tests.log; then
for rep in 1 2; do for v in 0 auto; do
  if [ $v = 0 ]; then export LLPF_RBF_TAIL=0; else unset LLPF_RBF_TAIL; fi
  echo "tail=$v rep$rep $(timeout 300 python bench.py --workload rbpf_full --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f k_rbfull %.2f' % (d['value'], d['roofline']['whole_timestep']['us'], d['roofline']['avg_launch_us']))")" >> $O/c5_tail_ab.txt
done; done
unset LLPF_RBF_TAIL
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $R/$O/kt_c5 -o kt -- python $R/bench.py --workload rbpf_full --steps 2 --no-cpu-baseline > $R/$O/kt_c5.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/kt_c5 -name "*.db" | head -1) > $R/$O/kernel_stats_rbpf_full_tail.txt
rm -rf $R/$O/kt_c5
cd $R
fi
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
ls -la $O
