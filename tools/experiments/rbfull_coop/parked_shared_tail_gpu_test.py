"""parked with the experiment (EXPERIMENTS.md 5.2): was part of tests/test_gpu_rbfull.py; green on the MI355X (29 passed, gpurun_out r05c / r05e)"""
# (not collected: pytest.ini restricts collection to tests/, and the file name does not match test_*.py.  To run it, build the
#  parked variant library of this directory, point LLPF_LIB at it and invoke pytest on this file explicitly.)
import os
import sys

import numpy as np
import pytest

_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from llpf_amd import _capi, _structs as S      # noqa: E402
import oracle_binding as ob                     # noqa: E402
import rbfull_models as M                       # noqa: E402
from test_gpu_rbfull import CASES, _cfg, _same_bits, _compare_state, _compare_linear_state      # noqa: E402

@pytest.mark.parametrize("name,strategy,tail", [("lin_4_8_2", S.RESAMPLE_SYSTEMATIC, 3), ("quadtank_4_8_2", S.RESAMPLE_SYSTEMATIC, 5),
                                                ("quadtank_4_8_2", S.RESAMPLE_STRATIFIED, 1), ("lin_4_8_2", S.RESAMPLE_RESIDUAL, 7),
                                                ("lin_2_8_1", S.RESAMPLE_SYSTEMATIC, 2)])
def test_rbfull_shared_tail_batches(name, strategy, tail, monkeypatch):
    """Round 5: the last batches of a k_rbfull launch are shared by the four waves of a workgroup (csrc/shared/llpf_rbfull_coop.h: one
    wave for the nonlinear state and the weight, three for the Kalman recursion by columns) instead of running as a whole extra batch
    on a few SIMDs.  The launcher takes that form by itself only at sizes like N = 2e5 (test_rbfull_large_and_repeated_runs,
    test_rbfull_awkward_sizes at 131073 / 131136); LLPF_RBF_TAIL forces `tail` shared batches on a small filter here: whole trajectories
    with a missing measurement, resampling and non-resampling steps, the propagate-only last step, single steps — bit-identical to the
    device-order oracle (which runs the sequential recursion) and to the engine itself with the tail switched off."""
    model = M.linear_case(2, 8, 1, seed=3)[0] if name == "lin_2_8_1" else CASES[name]()
    N, T = 4000 + 37, 30           # 64 batches (16 workgroups: room for every forced tail below), the last one partly filled
    U, Y = M.simulate_io(model, T)
    Y[11] = np.nan
    cfg = _cfg(model, N, strategy, 0.5, seed=11)
    monkeypatch.setenv("LLPF_RBF_TAIL", str(tail))
    monkeypatch.setenv("LLPF_GRAPH", "0")
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    rg = g.run(U, Y, 0.0, ll_steps=True); ro = o.run(U, Y, 0.0, ll_steps=True)
    assert o.resample_count() > 2
    assert _same_bits(rg["ll_steps"], ro["ll_steps"])
    _compare_state(g, o); _compare_linear_state(g, o)
    assert np.array_equal(g.ancestors(), o.ancestors())
    for k in range(6):
        assert g.update(U[k], Y[k], k * 1.0) == o.update(U[k], Y[k], k * 1.0)
    _compare_state(g, o); _compare_linear_state(g, o)
    monkeypatch.setenv("LLPF_RBF_TAIL", "0")
    g0 = _capi.FilterHandle(cfg)
    g0.reset()
    r0 = g0.run(U, Y, 0.0, ll_steps=True)
    assert _same_bits(r0["ll_steps"], ro["ll_steps"])
    # a bank of such filters (blockIdx.y): every filter with its own shared batches
    monkeypatch.setenv("LLPF_RBF_TAIL", str(tail))
    bank = _capi.BankHandle(cfg, None, n_filters=3)
    bank.reset()
    rb = bank.run(U, Y, 0.0, ll_steps=True)
    g1 = _capi.FilterHandle(cfg); g1.reset()
    assert _same_bits(rb["ll_steps"][:, 0], g1.run(U, Y, 0.0, ll_steps=True)["ll_steps"])
