import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_mc as B
model = M.lg_c1_model()
A = np.array(model.A[:4]).reshape(2, 2); Bm = np.array(model.B[:4]).reshape(2, 2); Cm = np.array(model.C[:4]).reshape(2, 2)
m0 = S.gaussian_mean(model.initial_density)
rng = np.random.default_rng(0)
for rep in range(2):
  for T in (20, 100, 200):
    for N in (10, 20, 50, 100, 200, 500, 1000):
        runs = 400000 // T // N
        U, Y, X = B.simulate_cell(A, Bm, Cm, m0, runs, T, rng)
        cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 5, 0)
        bank = _capi.BankHandle(cfg, [model] * runs)
        t0 = time.perf_counter(); r = bank.run_multi(U, Y, 0.0, xmean=True); t1 = time.perf_counter()
        r2 = bank.run_multi(U, Y, 0.0, xmean=False); 
        ms2 = bank.last_run_ms()
        if rep: print("T=%3d N=%4d F=%5d  device %.2f ms (%.1f us/step) wall %.2f ms ; without xmean %.2f ms resamples %d" % (T, N, runs, bank.last_run_ms(), 0, (t1-t0)*1e3, ms2, bank.resample_count()))
        del bank
