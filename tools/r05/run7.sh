#!/bin/bash
# round 5, GPU lease 7: LLPF_MAX_DIM 16 / RB ny <= 4 — the suite, and the four configs against the build before the change (lib_oldrbf.so)
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tools/ab/all_libs.sh lib_oldrbf.so > $O/all_ab.txt 2>&1
ls -la $O
