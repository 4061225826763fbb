#!/bin/bash
# round 5, GPU lease 6: the head's tile-sum burst (tools/head_burst.hip); the suite and the C2 / C3 / C4 / C5 lines on the restored build
O=gpurun_out/r05f; mkdir -p $O
tools/head_burst > $O/head_burst.txt 2>&1
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tools/ab/all_libs.sh lib_oldrbf.so > $O/all_ab.txt 2>&1
ls -la $O
