"""phase report of one timestep of the persistent kernel (LLPF_PERSIST_TIMING=<step> writes gpurun_out/llpf_ptiming.txt)"""
import numpy as np, sys
a = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = a[:, 0].min()
cols = [0, 1, 2, 3, 4, 6]
rel = (a[:, cols] - t0) * 10.0 / 1000.0      # 100 MHz ticks -> us
names = ["step start", "after head", "after counts", "after loop", "after tail", "after barrier"]
print("blocks", len(a))
for i, n in enumerate(names):
    print("%-14s min %.2f  median %.2f  p90 %.2f  max %.2f us" % (n, rel[:, i].min(), np.median(rel[:, i]), np.percentile(rel[:, i], 90), rel[:, i].max()))
d = np.diff(rel, axis=1)
for i, n in enumerate(["head", "counts", "loop", "tail", "barrier wait"]):
    print("phase %-12s min %.2f median %.2f  p90 %.2f  max %.2f us" % (n, d[:, i].min(), np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
print("outputs per block: min %d median %d max %d" % (a[:, 5].min(), np.median(a[:, 5]), a[:, 5].max()))
print("whole step (first start -> last leaves the barrier): %.2f us" % (rel[:, 5].max()))
