"""The quad-tank timestep in regimes other than BASELINE C3's (which is heavily degenerate: ~0.8 % of the particles survive a resampling):
measurement noise 50x larger (healthy ESS), resampling at every step or almost never.  Compares the source-side form (default) with the
round-3 form (LLPF_SOURCE_FX=0) and with what the handle chooses by itself from the previous run's survivor fraction (unset); one
subprocess per row."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import time
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import models as M
    from llpf_amd import _capi, _structs as S
    sig, thr = float(sys.argv[1]), float(sys.argv[2])
    m = M.quadtank_model()
    m.measurement_density = S.make_gaussian(np.zeros(2), np.full(2, sig ** 2))
    U, Y = M.quadtank_data(300, seed=2)
    Y = Y + sig * np.random.default_rng(1).standard_normal(Y.shape)
    pf = _capi.FilterHandle(S.make_config(m, 1000000, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 3, 0))
    for _ in range(3):
        pf.reset(); pf.run(U, Y, 1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        pf.reset(); r = pf.run(U, Y, 1.0)
    print(json.dumps({"us": round(1e6 * (time.perf_counter() - t0) / 3 / 300, 2), "resamples": int(pf.resample_count()), "distinct": int(len(np.unique(pf.ancestors()))), **pf.last_run_stats()}))
    sys.exit(0)
for sig, thr in ((0.01, 0.5), (0.03, 0.5), (0.05, 0.5), (0.1, 0.5), (0.5, 0.5), (0.1, 1.0), (0.5, 0.1)):
    for fx in ("1", "0", "auto"):
        e = dict(os.environ); e.pop("LLPF_SOURCE_FX", None)
        if fx != "auto": e["LLPF_SOURCE_FX"] = fx
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(sig), str(thr)], env=e, capture_output=True, text=True)
        print("sigma_meas", sig, "threshold", thr, "source_fx", fx, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
