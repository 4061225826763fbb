#!/bin/bash
# round 5: the suite on the k_norm prologue change, then the randomised parity sweep widened to 16 states / 8 inputs
O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
for seed in 11 12 13; do timeout 900 python tools/fuzz_parity.py --cases 300 --seed $seed > $O/fuzz_$seed.log 2>&1; tail -2 $O/fuzz_$seed.log; done
timeout 600 python tools/fuzz_parity.py --cases 25 --seed 14 --big > $O/fuzz_big.log 2>&1; tail -2 $O/fuzz_big.log
