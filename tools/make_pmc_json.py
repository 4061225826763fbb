#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the per-kernel PMC summaries tools/collect_profiles.sh writes (rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE in separate passes, tools/rocprof_pmc_summary.py).  The file carries the hash of the engine sources it was measured
with: bench.py quotes `roofline.traffic` from it only when that hash matches the sources of the build it is running.
usage: make_pmc_json.py <dir with pmc_traffic*.txt> <out.json> <tag>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import engine_source_hash

d, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def counters(path, kernel):
    res = {}
    if not os.path.exists(path):
        return res
    for line in open(path):
        if kernel in line:
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+)\s+avg=\s*([\d.]+)", line)
            if m and m.group(1) not in res:          # the first row that names the kernel
                res[m.group(1)] = float(m.group(3))
    return res


def entry(path, kernel, model_bytes, extra=None):
    c = counters(path, kernel)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    e = {"fetch_kb": c["FETCH_SIZE"], "write_kb": c["WRITE_SIZE"], "bytes": int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)),
         "kernel_model_bytes": model_bytes}
    if extra:
        e.update(extra)
    return e


N = 1000000
j = {"engine_source_hash": engine_source_hash(),
     "source": "profiles/%s_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, C2 workload N=1e6, T=200)" % tag,
     "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md #HBM: gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported; KB = 1024 B",
     "n_particles": N, "nx": 2, "resample_threshold": 1.0}
e = entry(os.path.join(d, "pmc_traffic.txt"), "k_resprop<llpf::LinGauss<2, 1>", 60 * N, {"algorithmic_bytes": 72 * N})
if e:
    j["k_resprop"] = e
# round 6: the same kernel with a working set beyond the Infinity Cache (N = 1.6e7: 1 GB)
NB = 16000000
e = entry(os.path.join(d, "pmc_traffic_c2_big.txt"), "k_resprop<llpf::LinGauss<2, 1>", 52 * NB, {"algorithmic_bytes": 72 * NB})      # split schedule: quanta 8 + gather 16, x 16 + w 8 + ancestors 4 (k_norm moves the other 16)
if e:
    j["c2_big"] = {"source": "profiles/%s_pmc_traffic_c2_big.txt (same recipe, C2 system at N=1.6e7, T=20)" % tag, "k_resprop": e, "n_particles": NB}
# "k_rbfull<": the template kernel, not k_rbfull_init (whose later row a bare substring match would return: round 2's summary did)
e = entry(os.path.join(d, "pmc_traffic_c5.txt"), "k_rbfull<", 788 * 200000)
if e:
    j["c5"] = {"source": "profiles/%s_pmc_traffic_c5.txt (same recipe, workload rbpf_full N=2e5, T=200)" % tag, "k_rbfull": e, "n_particles": 200000}
# round 4: k_step<..., MARKS>: marks 4 + gather f(x[anc]) 8 nx (mostly L2 hits: few distinct ancestors) + ancestors 4 + x 8 nx + w 8 + quanta 8
e = entry(os.path.join(d, "pmc_traffic_quadtank.txt"), "k_step<llpf::QuadTank", 88 * N)
er = entry(os.path.join(d, "pmc_traffic_quadtank.txt"), "k_resample_fx<llpf::QuadTank", 8 * N)
if e:
    j["c3"] = {"source": "profiles/%s_pmc_traffic_quadtank.txt (same recipe, workload quadtank N=1e6, T=100)" % tag, "k_step": e, "n_particles": N}
    if er:
        j["c3"]["k_resample_fx"] = er
# round 6, threshold 0.1 (95 % of the filter-steps do not resample): k_norm reads the weights and stores no quanta (8); the fused kernel reads
# weights 8 + x 16 and writes x 16 + w 8 (48; a step that resamples: + ancestors 4)
e1 = entry(os.path.join(d, "pmc_traffic_bank.txt"), "k_resprop<llpf::LinGauss<2, 1>", 48 * 12800000)
e2 = entry(os.path.join(d, "pmc_traffic_bank.txt"), "k_norm", 8 * 12800000)
if e1 and e2:
    j["c4"] = {"source": "profiles/%s_pmc_traffic_bank.txt (same recipe, workload bank 128 x 1e5, T=100)" % tag, "k_resprop": e1, "k_norm": e2,
               "filters": 128, "n_particles": 100000}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else "...") for k, v in j.items()}))
