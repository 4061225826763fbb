"""Where the blocks of the two C3 kernels (k_resample_fx, k_step<QuadTank>) spend their time, measured inside the real kernels.
Needs a library built with the stamps:  tools/ab/build_variant.sh steptiming k_step -DLLPF_STEP_TIMING
    LLPF_LIB=$PWD/lib_steptiming.so python tools/dbg/qt_phases.py [N]
Thread 0 of every block writes s_memtime at the phase boundaries of timestep 5 (kernels/step.hpp: DBG_STAMP) and its HW_ID / XCC_ID."""
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
nb = [(N + 1023) // 1024 * 2, (N + 1023) // 1024]
for w in (0, 1):
    assert L.llpf_debug_timing_arm(w, ctypes.c_int64(nb[w])) == 0
pf.reset()
pf.run(U, Y, 1.0)
NAMES = {"k_step": ["", "scalars back, tables in LDS", "marks back and scanned", "f(x[anc]) gathered", "noise drawn, particles formed",
                    "weights, exp-sums, stores issued", "block reductions, atomics"],
         "k_resample_fx": ["", "head", "scan + counts", "survivors listed", "x of survivors back, own marks", "heavy / stale marks", "dynamics + stores"],
         "k_fxstep": ["", "head", "noise of tile A", "scan + counts", "survivors, marks, noise of tile B", "heavy marks, dynamics + stores", "grid barrier",
                      "both output tiles", "block reductions, atomics"]}
for w in (1, 0):
    buf = np.zeros((nb[w], 16), dtype=np.uint64)
    assert L.llpf_debug_timing_read(w, buf.ctypes.data_as(ctypes.c_void_p)) == 0
    fused = w == 1 and (buf[:, 8] > 1 << 20).any()
    title = "k_fxstep" if fused else ("k_resample_fx" if w == 1 else "k_step")
    names = NAMES[title]
    last = len(names) - 1
    st = buf[:, :last + 1].astype(np.int64)
    ok = st[:, last] > 0
    if ok.sum() == 0:
        print("== %s: no stamps" % title)
        continue
    print("== %s: %d blocks, %d stamped to the end" % (title, nb[w], ok.sum()))
    hw = buf[:, 13].astype(np.int64)
    xcc = buf[:, 14].astype(np.int64) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    # s_memtime counters of different XCDs are not aligned: starts / ends relative to the first block of the same CU
    start = np.zeros(nb[w], dtype=np.int64)
    end = np.zeros(nb[w], dtype=np.int64)
    for c in np.unique(cu[ok]):
        m = ok & (cu == c)
        z = st[m, 0].min()
        start[m] = st[m, 0] - z
        end[m] = st[m, last] - z
    ids, cnt = np.unique(cu[ok], return_counts=True)
    print("CUs seen %d; blocks per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
    early = ok & (start < 3000)
    late = ok & ~early
    for nm, sel in (("start with the launch", early), ("take a freed slot", late)):
        if sel.sum() == 0:
            continue
        print("blocks that %s: %d; start p50 %d, end p50 %d p90 %d max %d" % (nm, sel.sum(), np.median(start[sel]), np.median(end[sel]), np.percentile(end[sel], 90), end[sel].max()))
        for k in range(1, last + 1):
            d = (st[:, k] - st[:, k - 1])[sel]
            print("    %-36s median %6d  p10 %6d  p90 %6d  max %6d" % (names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max()))
        print("    %-36s median %6d" % ("whole block", np.median((st[:, last] - st[:, 0])[sel])))
    if title == "k_resample_fx":
        D = buf[ok, 8].astype(np.int64); H = buf[ok, 9].astype(np.int64); outs = buf[ok, 10].astype(np.int64)
        print("survivors per tile: mean %.2f max %d, total %d; heavy sources %d; outputs per tile max %d" % (D.mean(), D.max(), D.sum(), H.sum(), outs.max()))
