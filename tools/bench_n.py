"""The timestep of ONE linear-Gaussian filter (BASELINE C2's system: nx = 2, systematic, resample every step) against the particle
count, from the size BASELINE quotes (1e6: the whole working set, ~64 MB, sits in the 256 MB Infinity Cache) to the largest filter
the engine builds (3e8: 19 GB).  What it shows (round 6):
  * the time per particle is flat once the working set has left the Infinity Cache — the head of the resampling kernels takes its tile
    prefix in O(tiles) (k_tile_prefix, kernels/resample.hpp) where every block used to read every tile sum: O(tiles^2) above 1024
    tiles, the reference's cumsum being O(N) (src/resample.jl:19-22);
  * the HBM-resident roofline fraction of the fused timestep (B_alg = 72 B per particle-step), which the 1e6-particle
    headline cannot show because its loads hit the MALL.
    python tools/bench_n.py [--sizes 1000000,4000000,...] [--schedule fused|balanced]
Device time from the engine's own HIP events around the run (llpf_last_run_ms); three timed passes per size after two warm-up
passes (the second pass captures the run loop into a hipGraph)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1000000,4000000,16000000,64000000,300000000")
ap.add_argument("--schedule", default="fused", choices=["fused", "balanced"])
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--threshold", type=float, default=1.0, help="resample_threshold: 1.0 = every step resamples (the stress case the headline is quoted on), 0.1 = the reference's default")
a = ap.parse_args()
if a.schedule == "balanced":
    os.environ["LLPF_UNFUSED"] = "1"

import numpy as np
from llpf_amd import _capi, _structs as S
import models as M

model = M.lg_test_model()
rows = []
for N in [int(x) for x in a.sizes.split(",")]:
    T = max(8, min(1000, int(2e9 / N)))           # ~2e9 particle-steps per pass
    _, U, Y = M.simulate_lg(model, T, seed=1)
    pf = _capi.FilterHandle(S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, a.threshold, 1000, 0))
    for _ in range(2):
        pf.reset(); pf.run(U, Y, 1.0)
    ms = []
    for _ in range(a.passes):
        pf.reset(); r = pf.run(U, Y, 1.0)
        ms.append(pf.last_run_ms())
    us = 1e3 * min(ms) / T
    tiles = (N + 1023) // 1024
    rows.append({"particles": N, "timesteps": T, "tiles": tiles, "tile_prefix": "k_tile_prefix (two levels)" if tiles > 1024 else "every block reads the tile sums",
                 "working_set_MB": round(N * (2 * 2 * 8 + 8 + 4 + 2 * 8) / 1e6, 1),
                 "us_per_timestep": round(us, 2), "us_per_timestep_passes": [round(1e3 * m / T, 2) for m in ms],
                 "ns_per_particle_step": round(1e3 * us / N, 5), "particle_steps_per_s": N / (us * 1e-6),
                 "B_alg_bytes": 72, "whole_timestep_roofline_frac": round(N * 72 / (us * 1e-6) / 8e12, 4), "loglik": r["ll"], "resamples": int(pf.resample_count())})
    del pf
flat = [r["ns_per_particle_step"] for r in rows if r["particles"] >= 4000000]
print(json.dumps({"workload": "C2 system (2-D linear-Gaussian, systematic, resample_threshold %g), one filter" % a.threshold, "schedule": a.schedule, "resample_threshold": a.threshold, "rows": rows,
                  "ns_per_particle_step_spread_from_4e6": (round(max(flat) / min(flat), 4) if flat else None)}, indent=1))
