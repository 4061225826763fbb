"""The engine's remaining hand-written instruction blocks keep the hardware hazards themselves (the compiler's hazard pass does not
look into inline assembly).  Round 5 found one of them a wait state short for gfx950 — a 16-byte write-through store whose data
registers the following instruction could overwrite, once in ~600 runs of a 10^6-particle filter (EXPERIMENTS.md 5.9).  Round 6 took
that store out of our hands (kernels/reduce.hpp: wt_store2 is a compiler-scheduled buffer-store builtin); what is left in assembly are
the DPP-modified adds of the wave scans (no builtin carries a DPP modifier on an add).  These tests compile the REAL functions of
reduce.hpp into a small kernel and read the emitted ISA — nothing is matched against the source text.  No GPU needed: hipcc
cross-compiles."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")

PRELUDE = """#include "engine.hpp"
namespace llpf {
#define DEV __device__ __forceinline__
#include "kernels/reduce.hpp"
"""


def _isa(src, flags=()):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "k.hip")
        open(p, "w").write(src)
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                        "-Wno-unused-value", "-S", p, "-o", out] + list(flags), check=True, stderr=subprocess.DEVNULL)
        return [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def _states(l):
    m = re.match(r"s_nop (\d+)", l)
    return int(m.group(1)) + 1 if m else 1


def test_the_16_byte_write_through_store_is_the_compilers():
    """wt_store2 emits ONE buffer_store_dwordx4 with the sc1 policy, from a uniform descriptor (no waterfall loop), and the compiler
    itself pads it when the next VALU instruction overwrites the data registers; no hand-written store is left in reduce.hpp's ISA."""
    lines = _isa(PRELUDE + """
__global__ void k(double* base, const double* in) {
    const int64_t i = 2 * (int64_t)(blockIdx.x * 256 + threadIdx.x);
    double a = in[i], b = in[i + 1];
    wt_store2(base, i, a, b);
    asm volatile("" : "+v"(a), "+v"(b));        // the same registers ...
    a = a * 3.0; b = b * 5.0;                    // ... overwritten by the next VALU instructions
    wt_store2(base + 4096, i, a, b);
}
}""")
    st = [k for k, l in enumerate(lines) if l.startswith("buffer_store_dwordx4")]
    assert len(st) == 2 and all(" sc1" in lines[k] and "offen" in lines[k] for k in st), [lines[k] for k in st]
    assert not any(l.startswith(("global_store_dwordx4", "v_readfirstlane", "s_and_saveexec")) for l in lines), "a hand-written store or a waterfall loop"
    # between the first store and the first VALU write: the wait states are the compiler's (gfx950: two behind a store of > 64 bits)
    k = st[0] + 1
    n = 0
    while not lines[k].startswith("v_"):
        n += _states(lines[k]); k += 1
    assert n >= 2, "the compiler left %d wait state(s) between a 16-byte store and a VALU write of its data: %s" % (n, lines[st[0]:k + 1])


def test_narrow_write_through_stores_are_atomics_the_compiler_sees():
    lines = _isa(PRELUDE + """
__global__ void k(double* p, int32_t* q, double v) { wt_store(p + threadIdx.x, v); wt_store(q + threadIdx.x, (int32_t)threadIdx.x); }
}""")
    assert any(re.match(r"global_store_dwordx2 .* sc1", l) for l in lines) and any(re.match(r"global_store_dword .* sc1", l) for l in lines)


def _scan_isa():
    return _isa(PRELUDE + """
__global__ void k(uint64_t* p, llpf_u128* q) {
    uint64_t v = p[threadIdx.x];
    v = v * 3 + 1;                                   // a VALU write right in front of the block
    p[threadIdx.x] = wave_scan_u64(v);
    llpf_u128 w = q[threadIdx.x];
    w.lo += 1;
    q[threadIdx.x] = wave_sum_u128(w);
    p[threadIdx.x + 64] = (uint64_t)wave_scan_max_u32((uint32_t)v);      // DPP instructions of the compiler's behind the blocks
}
}""")


def test_dpp_blocks_keep_their_wait_states_in_the_emitted_isa():
    """(1) what this toolchain itself leaves between a VALU write and a DPP read of the register (two wait states on gfx950); (2) every
    hand-written DPP add of the scan blocks has at least that many instruction slots since the last write of the register it reads,
    the blocks open with five wait states (a VALU write of EXEC in front of DPP) and close with two."""
    ref = _isa("""#include <hip/hip_runtime.h>
__global__ void k(int* p, int a) {
    int v = p[threadIdx.x];
    int w = v + a;
    int t = __builtin_amdgcn_update_dpp(w, w, 0x111, 0xF, 0xF, false);
    p[threadIdx.x] = t + w;
}""")
    kd = next(k for k, l in enumerate(ref) if "_dpp" in l)
    src = re.match(r"v_mov_b32_dpp v\d+, (v\d+) ", ref[kd]).group(1)
    need, j = 0, kd - 1
    while not re.match(r"v_\w+ %s[, ]" % src, ref[j]):
        need += _states(ref[j]); j -= 1
    assert 1 <= need <= 2, "this toolchain wants %d wait states between a VALU write and a DPP read: the blocks of reduce.hpp give 2" % need
    lines = _scan_isa()
    dpp_adds = [k for k, l in enumerate(lines) if re.match(r"v_addc?_co_u32_dpp", l)]
    assert len(dpp_adds) == 6 * 2 + 6 * 4, "six steps of two (64-bit) and six of four (128-bit) DPP adds expected, found %d" % len(dpp_adds)
    for k in dpp_adds:
        dst = re.match(r"v_addc?_co_u32_dpp (v\d+),", lines[k]).group(1)
        # slots since the previous instruction that wrote the register this add reads through DPP (itself: %0 op= dpp(%0))
        n, j = 0, k - 1
        while j >= 0 and not re.match(r"v_\w+ %s[, ]" % dst, lines[j]):
            n += _states(lines[j]); j -= 1
        assert n >= 2, "only %d wait state(s) in front of `%s` (written by `%s`)" % (n, lines[k], lines[j])
    # each block: s_nop 4 in front of its first DPP add, s_nop 1 behind its last
    firsts = [k for k in dpp_adds if k - 1 not in dpp_adds and not (lines[k - 1].startswith("s_nop 0") and k - 2 in dpp_adds)]
    assert len(firsts) == 2 and all(lines[k - 1] == "s_nop 4" for k in firsts), [lines[k - 1] for k in firsts]
    lasts = [k for k in dpp_adds if k + 1 not in dpp_adds and not (lines[k + 1].startswith("s_nop 0") and k + 2 in dpp_adds)]
    assert len(lasts) == 2 and all(_states(lines[k + 1]) >= 2 or lines[k + 1] == "s_nop 0" and _states(lines[k + 2]) >= 2 for k in lasts), [lines[k + 1:k + 3] for k in lasts]
