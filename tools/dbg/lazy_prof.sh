mkdir -p gpurun_out/r07a
cd /tmp && export TMPDIR=/tmp
for lazy in 1 0; do
  LLPF_LAZY_Q=$lazy rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r07a/kt_lazy$lazy -o kt -- python $GRAFT_REPO_ROOT/bench.py --particles 16000000 --T 100 --threshold 0.1 --steps 3 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r07a/bench_lazy$lazy.json 2> /dev/null
done
cd $GRAFT_REPO_ROOT
for lazy in 1 0; do
  python tools/rocprof_summary.py $(find gpurun_out/r07a/kt_lazy$lazy -name "*.db" | head -1) > gpurun_out/r07a/kernel_stats_lazy$lazy.txt
  rm -rf gpurun_out/r07a/kt_lazy$lazy
  echo "== lazy $lazy"; head -12 gpurun_out/r07a/kernel_stats_lazy$lazy.txt | cut -c1-60,100-175
  python -c "
import json; d=json.loads(open('gpurun_out/r07a/bench_lazy$lazy.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'device_ms', d.get('device_ms_per_step'))"
done
