#!/bin/bash
# round 5: after the store-hazard fix — the suite, the randomised sweep (small and full size), the history stress once more
O=gpurun_out/r05q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
for seed in 21 22; do timeout 900 python tools/fuzz_parity.py --cases 300 --seed $seed 2>&1 | grep -a "FAIL\|failed" | cut -c1-1200 | tee -a $O/fuzz.log; done
for seed in 14 15 31 32 33 34 35 36 37 38 39 40; do timeout 600 python tools/fuzz_parity.py --cases 25 --seed $seed --big 2>&1 | grep -a "FAIL\|failed" | cut -c1-1500 | tee -a $O/fuzz_big.log; done
python tools/r05/stress_hist.py 3000 8 hist 2>&1 | grep -v "^ \[\|^\[\|rows" | tail -3 | tee $O/stress_hist.log
