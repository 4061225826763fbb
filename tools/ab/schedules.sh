#!/bin/bash
# merged vs split schedule on one box (C2 single filter and the 128 x 1e5 bank)
for s in merged split; do
  c2=$(LLPF_SCHEDULE=$s python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  bk=$(LLPF_SCHEDULE=$s python tools/bench_bank.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e' % d['particle_steps_per_s'])")
  echo "schedule=$s C2_us_per_timestep=$c2 bank_128x1e5=$bk"
done
