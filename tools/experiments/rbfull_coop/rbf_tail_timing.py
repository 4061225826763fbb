"""The shared tail batches of k_rbfull (kernels/rbfull.hpp, shared/llpf_rbfull_coop.h) on the launch's time line, measured inside the kernel.
Needs a library built with the stamps:  tools/ab/build_variant.sh timing k_rbfull -DLLPF_RBF_TIMING
    LLPF_LIB=$PWD/lib_timing.so python tools/dbg/rbf_tail_timing.py [N]
Rows of ordinary batches carry the stamps of tools/dbg/rbf_timing.py (0 = start of the batch, 12 = its end); the rows of the shared
batches carry the nonlinear wave's: it passes every barrier as soon as the three Kalman waves reach it, so the gaps between its
barrier stamps are the Kalman waves' stages.  s_memtime is a per-XCD counter: every row is referred to the first start on its own XCD."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import llpf_amd  # noqa: E402
from llpf_amd import _capi, _structs as S  # noqa: E402
import rbfull_models as RM  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
T = 6
model = RM.quadtank_case()
U, Y = RM.simulate_io(model, T, seed=3)
cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 1000, 0)
pf = _capi.FilterHandle(cfg)
L = ctypes.CDLL(_capi.LIB_PATH)
rows = (N + 1023) // 1024 * 16
assert L.llpf_debug_rbf_timing_arm(ctypes.c_int64(rows)) == 0
os.environ["LLPF_GRAPH"] = "0"
pf.reset()
pf.run(U, Y, 1.0)
buf = np.zeros((rows, 32), dtype=np.uint64)
assert L.llpf_debug_rbf_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.astype(np.int64)
nreal = (N + 63) // 64
tail_env = os.environ.get("LLPF_RBF_TAIL")
ntail = int(tail_env) if tail_env else (nreal % 1024 if 0 < nreal % 1024 <= 128 and nreal > 1024 else 0)
nfull = nreal - ntail
xcc = st[:, 14] & 0xf
main = np.arange(rows) < nfull
t0 = {g: st[main & (xcc == g), 0].min() for g in range(8) if (main & (xcc == g)).any()}
rel = lambda r, k: st[r, k] - t0[int(xcc[r])]
ends = np.array([rel(r, 12) for r in range(nfull)])
starts = np.array([rel(r, 0) for r in range(nfull)])
print("N = %d: %d ordinary batches, %d shared; ordinary batches: last start %d, ends p50 %d p90 %d max %d ticks after their XCD's first start"
      % (N, nfull, ntail, starts.max(), np.median(ends), np.percentile(ends, 90), ends.max()))
second = starts > 4000
if second.any():
    print("  second-round batches: %d, start p50 %d, end p50 %d max %d; first-round ends p50 %d max %d"
          % (second.sum(), np.median(starts[second]), np.median(ends[second]), ends[second].max(), np.median(ends[~second]), ends[~second].max()))
if ntail:
    names = ["entry (all four waves through with their own batches)", "xn, xl back", "noise", "barrier 1 (An R exchanged)", "An xl", "f_n (RK4)",
             "barrier 2 (V, x~l)", "barrier 3 (M = Al R~)", "barrier 4 (C R1)", "barrier 5 (K)", "weight, stores"]
    tr = np.arange(nfull, nreal)
    ent = np.array([rel(r, 0) for r in tr])
    fin = np.array([rel(r, 10) for r in tr])
    print("shared batches: entry p10 %d p50 %d p90 %d, end p50 %d p90 %d max %d" % (np.percentile(ent, 10), np.median(ent), np.percentile(ent, 90), np.median(fin), np.percentile(fin, 90), fin.max()))
    for k in range(1, 11):
        d = st[tr, k] - st[tr, k - 1]
        print("  %-52s median %6d  p10 %6d  p90 %6d" % (names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
