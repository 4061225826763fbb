"""The timestep of a single linear-Gaussian filter (N = 1e6, resample every step) against the state dimension and the schedule:
  fused      one launch (k_resprop), precompiled for nx <= 4                                   LLPF_UNFUSED=0
  balanced   k_resample + k_step, ancestors through HBM                                        LLPF_UNFUSED=1
  default    what the engine picks with no switch set (host/run.hpp: fused for nx <= 2, balanced from nx = 3 on)
nx >= 5 is compiled on demand (hiprtc) and has the balanced form only.  The round-3 review asked for the step between nx = 4 and nx = 5
to be visible: measuring it showed the cliff was at nx = 3 — the fused kernel needs three waves per SIMD there — which is why those
dimensions now run balanced.  (The source-side form of the quad-tank, k_resample_fx + k_step<MARKS>, was measured for this model too and
costs it 60-130 %: EXPERIMENTS.md 4.14.)
    python tools/bench_nx.py [--particles 1000000] [--T 200]          (one subprocess per row: the schedule switches are read per process)"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1000000)
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--one", type=int, default=0, help="(internal) run one row: nx")
a = ap.parse_args()

if a.one:
    import time
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    from llpf_amd import _capi, _structs as S
    nx, ny, nu = a.one, 2, 1
    rng = np.random.default_rng(nx)
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = Q @ np.diag(np.linspace(0.5, 0.95, nx)) @ Q.T
    B = rng.standard_normal((nx, nu)); Cm = rng.standard_normal((ny, nx))
    g = S.make_gaussian
    m = S.make_lg_model(A, B, Cm, g(np.zeros(nx), 0.01), g(np.zeros(ny), 1.0), g(np.zeros(nx), 4.0), 1.0)
    # data simulated from the model itself (simulate semantics, src/filtering.jl:462-477): a filter that tracks, not one that fights its data
    U = rng.standard_normal((a.T, nu)); Y = np.zeros((a.T, ny))
    x = np.zeros(nx)
    for k in range(a.T):
        Y[k] = Cm @ x + rng.standard_normal(ny)
        x = A @ x + B @ U[k] + 0.1 * rng.standard_normal(nx)
    pf = _capi.FilterHandle(S.make_config(m, a.particles, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 1.0, 3, 0))
    for _ in range(3):
        pf.reset(); pf.run(U, Y, 1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        pf.reset(); r = pf.run(U, Y, 1.0)
    print(json.dumps({"us": 1e6 * (time.perf_counter() - t0) / 3 / a.T, "ll": r["ll"]}))
    sys.exit(0)

rows = []
for nx in (2, 3, 4, 5, 8):
    for name, env in (("fused", {"LLPF_UNFUSED": "0"}), ("balanced", {"LLPF_UNFUSED": "1"}), ("default", {})):
        if nx > 4 and name == "fused":
            continue
        e = dict(os.environ); e.pop("LLPF_UNFUSED", None); e.update(env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(nx), "--particles", str(a.particles), "--T", str(a.T)],
                             env=e, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            rows.append({"nx": nx, "schedule": name, "error": (out.stderr or out.stdout)[-300:]})
            continue
        rows.append({"nx": nx, "ny": 2, "schedule": name, "compiled": "hiprtc" if nx > 4 else "precompiled", "us_per_timestep": round(d["us"], 2), "loglik": d["ll"],
                     "B_alg_bytes": 16 * nx + 40, "whole_timestep_roofline_frac": round(a.particles * (16 * nx + 40) / (d["us"] * 1e-6) / 8e12, 4)})
print(json.dumps({"particles": a.particles, "T": a.T, "default_schedule": "fused for nx <= 2, balanced from nx = 3 on", "rows": rows}, indent=1))
