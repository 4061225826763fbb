# A/B on one box: 10-round (product) vs 7-round Philox builds, C2 / C3 / C4 share / C5 rates
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_philox7.so; do
  for w in lg quadtank rbpf_full; do
  LLPF_LIB=$PWD/$lib timeout 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w rep$rep', '%.4e' % d['value'], 'us/timestep %.2f' % d['roofline']['whole_timestep']['us'])"
  done
  LLPF_LIB=$PWD/$lib timeout 200 python tools/bench_bank.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib bank rep$rep', '%.4e' % d['value'])"
done; done
