"""Phase stamps of ONE fused launch of a filter beyond the Infinity Cache (developer build: lib_dev.so = the engine with k_resprop*.o compiled
-DLLPF_DEVTOOLS; run with LLPF_LIB=$PWD/lib_dev.so LLPF_DEBUG_TIMING=<timestep>): how long a block lives, in which phase, and how many
blocks are in flight at a time.   python tools/dbg/timing_bign.py gpurun_out/llpf_timing.txt"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = a[:, 0].min()
rel = (a[:, :5] - t0) / 100.0          # 100 MHz ticks -> us
print("blocks", len(a), " launch length %.1f us" % rel[:, 4].max())
d = np.diff(rel, axis=1)
for i, n in enumerate(["head", "counts", "loop", "tail"]):
    print("phase %-7s median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90), d[:, i].max()))
life = rel[:, 4] - rel[:, 0]
print("lifetime      median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max()))
# blocks in flight, sampled
ts = np.linspace(0, rel[:, 4].max(), 23)[1:-1]
infl = [(int(((rel[:, 0] <= t) & (rel[:, 4] > t)).sum()), [int(((rel[:, k] <= t) & (rel[:, k + 1] > t)).sum()) for k in range(4)]) for t in ts]
print("in flight at 21 instants (total, [head, counts, loop, tail]):")
for t, (n, ph) in zip(ts, infl):
    print("  t = %7.1f us: %5d %s" % (t, n, ph))
print("outputs per block: min %d median %d max %d" % (a[:, 5].min(), np.median(a[:, 5]), a[:, 5].max()))
