#!/bin/bash
# round 5, last lease: the driver's command on the final build (the line it will record), the suite and the smoke test
O=gpurun_out/r05z; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -3 $O/gputests.log; tail -2 $O/smoke.log; tail -4 $O/bench_default.err
