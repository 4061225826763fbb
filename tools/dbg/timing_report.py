import numpy as np, sys
a=np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0=a[:,0].min()
rel=(a[:,:5]-t0)*10.0/1000.0   # 100 MHz ticks -> us
print("blocks", len(a))
names=["start","after head","after counts","after loop","end"]
for i,n in enumerate(names):
    print("%-14s min %.2f  median %.2f  p90 %.2f  max %.2f us"%(n, rel[:,i].min(), np.median(rel[:,i]), np.percentile(rel[:,i],90), rel[:,i].max()))
d=np.diff(rel,axis=1)
for i,n in enumerate(["head","counts","loop","tail"]):
    print("phase %-8s median %.2f  p90 %.2f  max %.2f us"%(n, np.median(d[:,i]), np.percentile(d[:,i],90), d[:,i].max()))
print("outputs per block: min %d median %d max %d"%(a[:,5].min(), np.median(a[:,5]), a[:,5].max()))
