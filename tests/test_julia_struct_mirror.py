"""The Julia wrapper cannot be executed in this image; this keeps its C-struct mirrors honest: the byte size of every
`struct C...` of julia/LLPFAmd.jl (computed from its field declarations) must equal the size of the ctypes mirror of the
same struct of include/llpf.h (lowlevelparticlefilters.jl_amd/_structs.py), and the model ids / strategy codes must agree."""
import ctypes as C
import os
import re

from llpf_amd import _structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "julia", "LLPFAmd.jl")

PRIM = {"Int32": (4, 4), "UInt32": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "Float64": (8, 8), "Cint": (4, 4)}


def _parse_structs(text):
    out = {}
    for m in re.finditer(r"^struct (C\w+)(.*?)^end", text, re.S | re.M):
        body = re.sub(r"#.*", "", m.group(2))
        fields = []
        for decl in re.split(r"[;\n]", body):
            decl = decl.strip()
            if "::" in decl:
                fields.append(decl.split("::", 1)[1].strip())
        out[m.group(1)] = fields
    return out


def _size_align(t, structs):
    if t in PRIM:
        return PRIM[t]
    if t.startswith("Ptr{"):
        return 8, 8
    m = re.match(r"NTuple\{(\d+),\s*(\w+)\}", t)
    if m:
        s, a = _size_align(m.group(2), structs)
        return s * int(m.group(1)), a
    off, al = 0, 1
    for f in structs[t]:
        s, a = _size_align(f, structs)
        off = (off + a - 1) // a * a + s
        al = max(al, a)
    return (off + al - 1) // al * al, al


def test_julia_struct_sizes_match_the_c_abi():
    structs = _parse_structs(open(JL).read())
    pairs = {"CGaussian": S.Gaussian, "CRBCoupling": S.RBCoupling, "CModel": S.Model, "CConfig": S.Config, "CRunOutputs": S.RunOutputs}
    for name, ct in pairs.items():
        assert name in structs, name
        assert _size_align(name, structs)[0] == C.sizeof(ct), (name, _size_align(name, structs)[0], C.sizeof(ct))


def test_julia_model_ids_match_the_header():
    jl = open(JL).read()
    hdr = open(os.path.join(ROOT, "include", "llpf.h")).read()
    ids = dict(re.findall(r"(LLPF_MODEL_\w+)\s*=\s*(\d+)", hdr))
    assert ids == {"LLPF_MODEL_LINEAR_GAUSSIAN": "0", "LLPF_MODEL_QUADTANK_RK4": "1", "LLPF_MODEL_RB_LINEAR": "2", "LLPF_MODEL_RB_BILINEAR": "3"}
    # the wrapper builds CModel(<id>, ...) literally: one constructor call per model kind
    assert re.search(r"cmodel\(m::LinearGaussianModel.*?CModel\(0,", jl, re.S)
    assert re.search(r"cmodel\(m::QuadTankModel.*?CModel\(1,", jl, re.S)
    assert re.search(r"cmodel\(m::RBLinearModel.*?CModel\(2,", jl, re.S)
    assert re.search(r"cmodel\(m::RBBilinearModel.*?CModel\(3,", jl, re.S)
    # every ccall names a symbol the header declares
    declared = set(re.findall(r"\b(llpf_\w+)\s*\(", hdr))
    for sym in set(re.findall(r"ccall\(\(:(llpf_\w+), LIB\)", jl)):
        assert sym in declared, sym
    for sym in set(re.findall(r"getvec\(:(llpf_\w+)", jl)):
        assert sym in declared, sym
