"""Where the blocks of k_step<QuadTank> (BASELINE config C3) spend their time, measured inside the real kernel.
Needs a library built with the stamps (kept as a patch so that the product sources carry no developer code):
    git apply tools/dbg/step_timing.patch && tools/ab/build_variant.sh steptiming k_step -DLLPF_STEP_TIMING && git apply -R tools/dbg/step_timing.patch
    LLPF_LIB=$PWD/lib_steptiming.so python tools/dbg/step_timing.py [N]
The first wave of every block writes s_memtime at the phase boundaries (kernels/step.hpp: STEP_STAMP) and its HW_ID / XCC_ID; the
report gives the median ticks per phase, for the blocks that start with the launch and for the ones that take a freed slot, and how the
blocks of one CU follow each other."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from llpf_amd import _capi, _structs as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
T = 8
model, U, Y, kind, thr, label = bench.build_workload("quadtank", N, T)
pf = _capi.FilterHandle(S.make_config(model, N, kind, S.RESAMPLE_SYSTEMATIC, thr, 1000, 0))
L = ctypes.CDLL(_capi.LIB_PATH)
blocks = (N + 1023) // 1024 * 2
assert L.llpf_debug_step_timing_arm(ctypes.c_int64(blocks)) == 0
pf.reset()
pf.run(U, Y, 1.0)
buf = np.zeros((blocks, 16), dtype=np.uint64)
assert L.llpf_debug_step_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf[:, :8].astype(np.int64)
hw = buf[:, 13].astype(np.int64)
xcc = buf[:, 14].astype(np.int64) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)        # cu, sh, se, xcc
names = ["", "scalars back, tables in LDS", "ancestors back", "runs of equal ancestors counted", "shared dynamics evaluated, handed out",
         "noise drawn, particles formed", "weights, exp-sums, stores issued", "block reductions, atomics"]
print("N = %d: %d blocks of 256 threads (the run's last launch is the closing propagate: stamps 0-5 and 7 are its own)" % (N, blocks))
ids, cnt = np.unique(cu, return_counts=True)
print("CUs seen %d; blocks per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
start = np.zeros(blocks, dtype=np.int64)
end = np.zeros(blocks, dtype=np.int64)
for c in ids:
    m = cu == c
    z = st[m, 0].min()
    start[m] = st[m, 0] - z
    end[m] = st[m, 7] - z
early = start < 3000
for nm, sel in (("start with the launch", early), ("take a freed slot", ~early)):
    print("blocks that %s: %d; start p50 %d, end p50 %d p90 %d max %d" % (nm, sel.sum(), np.median(start[sel]), np.median(end[sel]), np.percentile(end[sel], 90), end[sel].max()))
    for k in (1, 2, 3, 4, 5, 7):
        prev = {1: 0, 2: 1, 3: 2, 4: 3, 5: 4, 7: 5}[k]
        d = (st[:, k] - st[:, prev])[sel]
        print("    %-44s median %6d  p10 %6d  p90 %6d" % (names[k] if k != 7 else "weights, exp-sums (none here), reductions", np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    print("    %-44s median %6d" % ("whole block", np.median((st[:, 7] - st[:, 0])[sel])))
