#!/bin/bash
# k_rbfull us against the number of waves per SIMD (N = 65536 is one wave on each of the 1024 SIMDs)
for n in 65536 131072 196608 200000 262144 524288; do
  python bench.py --workload rbpf_full --particles $n --no-cpu-baseline --steps 1 --T 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, '%.3e' % d['value'], [round(v,1) if v else None for v in list(d['kernel_us'].values())[:3]])"
done
