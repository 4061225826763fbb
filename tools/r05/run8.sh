#!/bin/bash
# round 5, GPU lease 8: the suite on the final build
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
ls -la $O
