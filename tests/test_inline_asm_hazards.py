"""The engine's hand-written instruction blocks keep the hardware hazards themselves (the compiler's hazard pass does not look
into inline assembly).  Round 5 found one of them a wait state short for gfx950 — a 16-byte write-through store whose data
registers the following instruction could overwrite, once in ~600 runs of a 10^6-particle filter (EXPERIMENTS.md 5.9).  These
tests hold the blocks to what THIS toolchain's compiler inserts for its own instructions in the same situation, so that a
change of either side shows up here instead of as a rare wrong particle on the device.  No GPU needed: hipcc cross-compiles."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REDUCE = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "csrc", "kernels", "reduce.hpp")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def _isa(src):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "k.hip")
        open(p, "w").write(src)
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", p, "-o", out], check=True, stderr=subprocess.DEVNULL)
        return [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def _wait_states(lines, first, second):
    """wait states the compiler left between the first instruction matching `first` and the next one matching `second`"""
    i = next(k for k, l in enumerate(lines) if re.match(first, l))
    n = 0
    for l in lines[i + 1:]:
        if re.match(second, l):
            return n
        m = re.match(r"s_nop (\d+)", l)
        n += int(m.group(1)) + 1 if m else 1
    raise AssertionError("no %s after %s" % (second, first))


def test_wide_store_wait_states_match_the_compiler():
    # the compiler's own dwordx4 store, its data registers pinned and overwritten by the next VALU instruction
    lines = _isa("""#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(u32x4* p, unsigned a, unsigned b) {
    u32x4 v; v.x = a + threadIdx.x; v.y = b; v.z = a; v.w = b + 1;
    u32x4* q = p + threadIdx.x;
    __builtin_nontemporal_store(v, q);
    asm volatile("" : "+v"(v));
    v.x = v.x ^ a; v.y = v.y ^ b; v.z = v.z + 1; v.w = v.w + 2;
    __builtin_nontemporal_store(v, q + 64);
}""")
    # (the s_nop straight behind the store is the store's; a further one behind the register pin is the compiler's caution about
    # the inline asm's own output)
    i = next(k for k, l in enumerate(lines) if l.startswith("global_store_dwordx4"))
    m0 = re.match(r"s_nop (\d+)", lines[i + 1])
    assert m0, "the compiler left no wait state behind its own 16-byte store: %s" % lines[i:i + 4]
    need = int(m0.group(1)) + 1
    src = open(REDUCE).read()
    stores = re.findall(r'asm volatile\("global_store_dwordx4[^"]*"', src)
    assert stores, "the 16-byte write-through store of wt_store is gone: update this test"
    for s in stores:
        m = re.search(r"s_nop (\d+)", s)
        assert m, "no wait state behind the inline store: %s" % s
        assert int(m.group(1)) + 1 >= need, "wt_store leaves %d wait state(s) behind its 16-byte store, the compiler leaves %d on gfx950" % (int(m.group(1)) + 1, need)


def test_dpp_wait_states_match_the_compiler():
    # a VALU write of a register and a DPP read of it: the scan blocks of reduce.hpp put one instruction and `s_nop 0` (or three
    # instructions) between the two, i.e. two wait states, and end with `s_nop 1` for a DPP instruction of the compiler's behind them
    lines = _isa("""#include <hip/hip_runtime.h>
__global__ void k(int* p, int a) {
    int v = p[threadIdx.x];
    int w = v + a;
    int t = __builtin_amdgcn_update_dpp(w, w, 0x111, 0xF, 0xF, false);
    p[threadIdx.x] = t + w;
}""")
    need = _wait_states(lines, r"v_mov_b32_e32 v\d+, v\d+", r"v_mov_b32_dpp")
    assert need <= 2, "this toolchain wants %d wait states between a VALU write and a DPP read: the blocks of reduce.hpp give 2" % need
    src = open(REDUCE).read()
    assert 'LLPF_ADD64_DPP(C) "v_add_co_u32_dpp %0, vcc, %0, %0 " C "\\n\\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " C "\\n\\ts_nop 0\\n\\t"' in src
    assert src.count('"s_nop 1"') >= 2 and src.count('asm("s_nop 4\\n\\t"') >= 2
