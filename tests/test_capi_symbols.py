"""The C-ABI shared library loads on a machine without a GPU, exports exactly the symbols include/llpf.h
declares, and refuses to construct a filter (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from llpf_amd import _capi, _structs as S
import models as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "llpf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(llpf_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    decl = _declared_symbols()
    assert len(decl) >= 40
    L = _capi.lib()
    for name in decl:
        assert hasattr(L, name), "libllpf_hip.so does not export %s" % name
    assert sorted(_capi.SYMBOLS) == decl, "ctypes binding and include/llpf.h disagree"


def test_struct_layout_matches_header():
    # struct_size is checked by llpf_create; a wrong layout must be rejected with LLPF_ERR_ARG
    model = M.lg_test_model()
    cfg = S.make_config(model, 100)
    cfg.struct_size = 8
    h = C.c_void_p()
    rc = _capi.lib().llpf_create(C.byref(cfg), C.byref(h))
    assert rc == _capi.ERR_ARG
    assert b"struct_size" in _capi.lib().llpf_last_error()


def test_version_and_resample_uniforms_need_no_gpu():
    ma, mi = C.c_int32(-1), C.c_int32(-1)
    assert _capi.lib().llpf_version(C.byref(ma), C.byref(mi)) == 0 and ma.value == 0
    import oracle_binding as ob
    for strategy, m in ((S.RESAMPLE_SYSTEMATIC, 1), (S.RESAMPLE_STRATIFIED, 64)):
        assert np.array_equal(_capi.resample_uniforms(strategy, m, 77, 5), ob.resample_uniforms(strategy, m, 77, 5))


@pytest.mark.skipif(_capi.device_count() > 0, reason="this check is for machines without a GPU")
def test_no_cpu_fallback():
    cfg = S.make_config(M.lg_test_model(), 100)
    with pytest.raises(_capi.LLPFError) as ei:
        _capi.FilterHandle(cfg)
    assert ei.value.code == _capi.ERR_NO_DEVICE
    with pytest.raises(_capi.LLPFError):
        _capi.logsumexp(np.zeros(4))


def test_bad_arguments_are_rejected():
    L = _capi.lib()
    h = C.c_void_p()
    cfg = S.make_config(M.lg_test_model(), 0)
    assert L.llpf_create(C.byref(cfg), C.byref(h)) == _capi.ERR_ARG
    cfg = S.make_config(M.lg_test_model(), 100, strategy=7)
    assert L.llpf_create(C.byref(cfg), C.byref(h)) == _capi.ERR_ARG
    assert L.llpf_reset(None) == _capi.ERR_ARG
