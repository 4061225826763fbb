for rep in 1 2; do for n in 196608 200000 262144; do
python bench.py --workload rbpf_full --particles $n --no-cpu-baseline --steps 2 --T 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n rep$rep', '%.3e' % d['value'], 'us/step %.1f' % (d['ms_per_step']*1000/300), 'k_rbfull', round(list(d['kernel_us'].values())[0],1), 'whole frac %.3f' % d['roofline']['whole_timestep']['frac'])"
done; done
