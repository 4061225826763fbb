"""Where the waves of k_rbfull (BASELINE config C5) spend their time, measured inside the real kernel.
Needs a library built with the stamps:  tools/ab/build_variant.sh timing k_rbfull -DLLPF_RBF_TIMING
    LLPF_LIB=$PWD/lib_timing.so python tools/dbg/rbf_timing.py [N]
Every wave writes s_memtime at the stage boundaries of the recursion (csrc/shared/llpf_rbfull_body.h: RBF_STAMP) and its
HW_ID; the report gives the median ticks per phase and, per SIMD, how much of the launch it had 0 / 1 / 2 waves resident."""
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
waves = (N + 1023) // 1024 * 16
assert L.llpf_debug_rbf_timing_arm(ctypes.c_int64(waves)) == 0
pf.reset()
pf.run(U, Y, 1.0)
buf = np.zeros((waves, 16), dtype=np.uint64)
assert L.llpf_debug_rbf_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf[:, :13].astype(np.int64)
hw = buf[:, 13].astype(np.int64)
names = ["", "prologue, RK4, generator (gather in flight)", "coupling rows: An, An R, Nt", "Cholesky, V, x~l, R~", "Al x~l + Bl u",
         "upper panel + upper-left block", "lower-left block", "lower panel", "lower-right block + R1l", "(predict -> correct)",
         "C R, S, Cholesky, log", "gain, mean, covariance", "exp-sums, stores, tail"]
t0 = st[:, 0].min()
print("N = %d: %d waves, last launch of the run; first start -> last end %d ticks" % (N, waves, st[:, 12].max() - t0))
for k in range(1, 13):
    d = st[:, k] - st[:, k - 1]
    print("  %-46s median %6d  p10 %6d  p90 %6d" % (names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
tot = st[:, 12] - st[:, 0]
print("  %-46s median %6d  p10 %6d  p90 %6d" % ("whole wave", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; the XCC comes from another register, so
# waves of different XCDs share a key here: residency is therefore reported per (key, overlapping intervals) only as a histogram
key = (hw >> 4) & 0xfff
start, end = st[:, 0] - t0, st[:, 12] - t0
span = end.max()
print("start times: p50 %d p90 %d max %d; end times: p10 %d p50 %d max %d" % (np.median(start), np.percentile(start, 90), start.max(),
                                                                              np.percentile(end, 10), np.median(end), end.max()))
order = np.argsort(start)
print("waves started in ticks [0,5%%) %d, [5,50%%) %d, [50,100%%) %d of the launch" % ((start < 0.05 * span).sum(), ((start >= 0.05 * span) & (start < 0.5 * span)).sum(), (start >= 0.5 * span).sum()))
