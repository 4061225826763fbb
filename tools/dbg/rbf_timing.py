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
buf = np.zeros((waves, 32), dtype=np.uint64)
assert L.llpf_debug_rbf_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf[:, :24].astype(np.int64)
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
# s_memtime is a per-XCD counter: the eight XCDs' values are far apart, so the waves are grouped by the gaps between their sorted
# start stamps and every group is referred to its own first start.  (The run's last launch is the closing time update: stamps 0-9
# and 12 are its own, 10 and 11 are left from the launch before.)
hw = buf[:, 13].astype(np.int64)
xcc = buf[:, 14].astype(np.int64) & 0xf          # HW_REG_XCC_ID: the XCD of the wave; s_memtime is a per-XCD counter
grp = xcc
simd = ((hw >> 4) & 0x3) | (((hw >> 8) & 0xf) << 2) | (((hw >> 12) & 0x1) << 6) | (((hw >> 13) & 0x7) << 7) | (xcc << 10)   # simd, cu, sh, se, xcc
start = np.zeros(waves, dtype=np.int64)
end = np.zeros(waves, dtype=np.int64)
for g in range(grp.max() + 1):
    m = grp == g
    start[m] = st[m, 0] - st[m, 0].min()
    end[m] = st[m, 12] - st[m, 0].min()
print("%d clock groups (XCDs); launch span per group: %s ticks" % (grp.max() + 1, [int(end[grp == g].max()) for g in range(grp.max() + 1)]))
span = end.max()
early = start < 4000
print("waves that start with the launch: %d; later: %d" % (early.sum(), (~early).sum()))
print("start of the later ones: p10 %d p50 %d p90 %d; ends: first-round p50 %d p90 %d, later p50 %d p90 %d max %d" % (
    np.percentile(start[~early], 10), np.median(start[~early]), np.percentile(start[~early], 90), np.median(end[early]), np.percentile(end[early], 90),
    np.median(end[~early]), np.percentile(end[~early], 90), end.max()))
for nm, sel in (("started with the launch", early), ("started later", ~early)):
    print("%s (%d waves)" % (nm, sel.sum()))
    for k in range(1, 10):
        d = (st[:, k] - st[:, k - 1])[sel]
        print("    %-46s median %6d  p10 %6d  p90 %6d" % (names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    d = (st[:, 12] - st[:, 9])[sel]
    print("    %-46s median %6d  p10 %6d  p90 %6d" % ("stores, tail", np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    print("    %-46s median %6d" % ("whole wave", np.median((st[:, 12] - st[:, 0])[sel])))
# per SIMD: how many waves it ran and how its second-round waves fared
ids, cnt = np.unique(simd, return_counts=True)
print("SIMDs seen %d; waves per SIMD: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
# The batches of a SIMD in start order (its waves share one counter whatever the clock domains are).  The launch is persistent:
# a wave's second batch carries 1 in column 15 and starts at the end of its first.
cont = buf[:, 15].astype(np.int64)
order_s = [0, 16, 17, 18, 19, 20, 1, 2, 3, 4, 5, 6, 7, 8, 12]
lab = ["start", "scalars", "operands", "planes req.", "generator", "xn back", "RK4", "coupling", "Chol/V/R~", "Al x", "up panel", "ll block", "lo panel", "lr block", "end"]
tl = []
kinds = []
for sid in ids[cnt == 3]:
    m = np.where(simd == sid)[0]
    m = m[np.argsort(st[m, 0])]
    z = st[m[0], 0]
    tl.append([[st[m[k], q] - z for q in order_s] for k in range(3)])
    kinds.append(tuple(int(cont[m[k]]) for k in range(3)))
tl = np.median(np.array(tl), axis=0)
print("continuation flags of the three batches (start order): %s" % dict(zip(*np.unique(np.array(kinds), axis=0, return_counts=True))) if False else "kinds: %s" % {k: kinds.count(k) for k in set(kinds)})
print("%-14s %8s %8s %8s" % ("stamp", "first", "second", "third"))
for q in range(len(order_s)):
    print("%-14s %8d %8d %8d" % (lab[q], tl[0, q], tl[1, q], tl[2, q]))
# when does a SIMD finish, by the number of batches it ran (N = 2e5 is 3125 batches on 1024 SIMDs: 53 of them run four)
fin = {}
for sid in ids:
    m = np.where(simd == sid)[0]
    fin.setdefault(len(m), []).append(int(st[m, 12].max() - st[m, 0].min()))
for n in sorted(fin):
    v = np.array(fin[n])
    print("SIMDs with %d batches: %d; first start -> last end: median %d  p90 %d  max %d ticks" % (n, len(v), np.median(v), np.percentile(v, 90), v.max()))
