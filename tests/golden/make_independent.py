"""Freezes the outputs of the INDEPENDENT numpy restatement (oracle/independent.py — shares no code with the engine or with
oracle/llpf_oracle.c) on the cases of tests/independent_cases.py as tests/golden/indep_<case>.npz, and prints how far the C
oracle's two orders are from them.  These are NOT reference outputs (Julia cannot run in this image): they pin the C oracle and
the engine against a second restatement of the same reference lines, in particular the per-particle Riccati / Kalman recursion
of the Rao-Blackwellized filter (src/rbpf.jl:206-221, src/filtering.jl:100-128), which oracle and engine share as one header.
    python tests/golden/make_independent.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import independent_cases as IC
import oracle_binding as ob

for name, case in IC.cases().items():
    r = IC.run_independent(ob, case)
    np.savez_compressed(os.path.join(HERE, "indep_%s.npz" % name), U=case["U"], Y=case["Y"], **r)
    line = "%-22s resamples %2d" % (name, int(r["resamples"]))
    for tag, order in (("reference order", ob.ORDER_REFERENCE), ("device order", ob.ORDER_DEVICE)):
        o = IC.oracle_of(ob, case, order)
        o.reset()
        ro = o.run(case["U"], case["Y"], case["t0"], ll_steps=True)
        line += " | %s: max|dll| %.1e max|dx| %.1e anc %s" % (tag, np.max(np.abs(ro["ll_steps"] - r["ll_steps"])),
                                                            np.max(np.abs(o.particles() - r["x_final"])), np.array_equal(o.ancestors(), r["anc_final"]))
        if case.get("rb"):
            line += " max|dR| %.1e" % np.max(np.abs(o.rb_linear_state()[1] - r["R_final"]))
    print(line)
