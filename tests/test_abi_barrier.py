"""include/llpf.h: "no C++ exception crosses the ABI".  Every export of libllpf_hip.so is a function-try-block whose handler
turns std::bad_alloc into LLPF_ERR_ALLOC and anything else into LLPF_ERR_INTERNAL (csrc/capi.hip: LLPF_TRY / LLPF_GUARD); the
host threads of a multi-GPU bank catch inside the thread.  The reference's analogue: a throw inside the likelihood becomes
-Inf, never a dead session (src/smoothing.jl:275-279).

Faults are injected with LLPF_TEST_THROW="<alloc|error|other>:<site>" (capi.hip: test_throw)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from llpf_amd import _capi, _structs as S
import models as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAPI = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "csrc", "capi.hip")


def _declared():
    txt = open(os.path.join(ROOT, "include", "llpf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(llpf_[a-z0-9_]+)\s*\(", txt)))


def test_every_export_is_guarded():
    src = open(CAPI).read()
    guarded = re.findall(r"LLPF_GUARD\((llpf_[a-z0-9_]+)\)", src)
    assert len(guarded) == len(set(guarded))
    # llpf_last_error returns the message itself (a c_str() of a thread-local: cannot throw) and is the one export without a status
    assert sorted(guarded + ["llpf_last_error"]) == _declared()
    # each guard closes a function-try-block of the export it names
    for name in guarded:
        assert re.search(r"^int %s\([^;{]*\)\s*LLPF_TRY\s*\{" % name, src, flags=re.M | re.S), name
    # no thread body without a handler of its own
    mb = open(os.path.join(os.path.dirname(CAPI), "host", "mbank.hpp")).read()
    assert mb.count("emplace_back([&") == 1 and "noexcept {" in mb and 'guard_catch("shard worker")' in mb


class _Inject:
    def __init__(self, spec):
        self.spec = spec

    def __enter__(self):
        os.environ["LLPF_TEST_THROW"] = self.spec      # os.environ assigns through putenv: the library's getenv sees it

    def __exit__(self, *a):
        del os.environ["LLPF_TEST_THROW"]


@pytest.mark.parametrize("kind,code,needle", [("alloc", _capi.ERR_ALLOC, b"out of host memory"),
                                              ("error", _capi.ERR_INTERNAL, b"injected at create"),
                                              ("other", _capi.ERR_INTERNAL, b"unknown exception")])
def test_a_throw_in_a_constructor_becomes_a_status(kind, code, needle):
    L = _capi.lib()
    cfg = S.make_config(M.lg_test_model(), 1000)
    h = C.c_void_p()
    with _Inject(kind + ":create"):
        assert L.llpf_create(C.byref(cfg), C.byref(h)) == code
        msg = L.llpf_last_error()
        assert needle in msg and b"llpf_create" in msg
        assert not h.value
        hb = C.c_void_p()
        assert L.llpf_bank_create(C.byref(cfg), None, 4, C.byref(hb)) == code
        assert b"llpf_bank_create" in L.llpf_last_error() and not hb.value
    # and the process, the library and the error slot are alive: the next call behaves as always
    assert L.llpf_reset(None) == _capi.ERR_ARG
    assert b"null handle" in L.llpf_last_error()


@pytest.mark.gpu
def test_a_throw_inside_a_run_leaves_a_usable_handle():
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 12)
    g = _capi.FilterHandle(S.make_config(model, 4096, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 7, 0))
    g.seed(7); g.reset()
    ref = g.run(U, Y, 0.0)["ll"]
    g.reset()
    for kind, code in (("alloc", _capi.ERR_ALLOC), ("error", _capi.ERR_INTERNAL)):
        with _Inject(kind + ":run"):
            with pytest.raises(_capi.LLPFError) as ei:
                g.run(U, Y, 0.0)
            assert ei.value.code == code
    g.seed(7); g.reset()          # (reset! alone keeps drawing fresh noise: the same seed again gives the same run)
    assert g.run(U, Y, 0.0)["ll"] == ref


@pytest.mark.gpu
def test_an_absurd_horizon_is_a_status():
    # T = 2^46 timesteps: the staging of the inputs alone is beyond any allocation; must come back as a status
    model = M.lg_test_model()
    g = _capi.FilterHandle(S.make_config(model, 1024, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 7, 0))
    g.reset()
    L = _capi.lib()
    y = np.zeros((4, model.ny))
    u = np.zeros((4, max(model.nu, 1)))
    ll = C.c_double()
    rc = L.llpf_run(g.h, _capi.dptr(u), _capi.dptr(y), C.c_int64(1 << 46), C.c_double(0.0), C.byref(ll), None)
    assert rc == _capi.ERR_ALLOC, rc
    assert L.llpf_last_error()
    # ... and the refused allocation does not poison the handle's next call (the runtime's sticky last-error is cleared)
    _, U, Y = M.simulate_lg(model, 5)
    g.seed(7); g.reset()
    assert np.isfinite(g.run(U, Y, 0.0)["ll"])


@pytest.mark.gpu
@pytest.mark.parametrize("site", ["shard", "thread"])
def test_a_throw_in_a_shard_thread_is_a_status(site):
    # two shards folded onto one GPU: mbank_foreach drives them from two host threads
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 6)
    cfg = S.make_config(model, 2048, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 7, 0)
    mb = _capi.MBankHandle(cfg, None, 4, devices=[0, 0])
    mb.seed(7); mb.reset()
    ref = mb.run(U, Y, 0.0)["ll"]
    for kind, code in (("alloc", _capi.ERR_ALLOC), ("error", _capi.ERR_INTERNAL)):
        with _Inject(kind + ":" + site):
            with pytest.raises(_capi.LLPFError) as ei:
                mb.reset()
            assert ei.value.code == code
    mb.seed(7); mb.reset()
    assert np.array_equal(ref, mb.run(U, Y, 0.0)["ll"])
