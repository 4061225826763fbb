#!/usr/bin/env python3
"""Instruction census of one kernel of the engine from the compiler's assembly (no GPU needed):

    python tools/isa_census.py [kernel-name-substring]      default: the C2 fused kernel

Compiles csrc/k_resprop.hip for gfx950 with the Makefile's flags to assembly (hipcc --cuda-device-only -S, ~3 min), cuts the
kernel out, splits it at the s_setprio markers of k_resprop (head+counts | output loop | tail) when they are present, and prints
per region the number of vector / scalar / LDS / memory instructions and the most frequent opcodes, plus the register and scratch
figures of the kernel.  This is the census EXPERIMENTS.md Appendix A (round 2) quotes."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "csrc")
DEFAULT = "k_respropINS_8LinGaussILi2ELi1EEELi2ELi1ELb1ELb1ELb0ELb0EEE"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    out = os.path.join(tempfile.gettempdir(), "llpf_kernels.s")
    flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 --offload-arch=gfx950".split()
    subprocess.check_call(["make", "-C", CSRC, "jit_prelude.inc"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "k_resprop.hip", "-o", out], cwd=CSRC,
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4llpf\S*%s\S*:" % re.escape(want), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section") and i > start + 10)
    body = lines[start:end]
    info = [l.strip("; ").strip() for l in lines[end:end + 60] if re.search(r"NumVgprs|NumAgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte", l)]
    print(lines[start].split(":")[0][:110])
    print("  " + "  ".join(info))
    code = [l for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
    marks = [i for i, l in enumerate(code) if l.strip().startswith("s_setprio")]
    regions = [("whole kernel", code)]
    if len(marks) >= 3:
        regions = [("head + counts (to s_setprio 0)", code[:marks[1]]), ("output loop region (s_setprio 0 .. 3)", code[marks[1]:marks[2]]),
                   ("tail", code[marks[2]:])]
    for name, reg in regions:
        ops = [l.split()[0] for l in reg]
        c = collections.Counter(ops)
        cls = lambda p: sum(n for o, n in c.items() if o.startswith(p))
        print("%-40s instructions %5d  vector %5d  scalar %5d  LDS %4d  global/scratch %4d" %
              (name, len(ops), cls("v_"), cls("s_"), cls("ds_"), cls("global_") + cls("scratch_") + cls("flat_")))
        print("    " + "  ".join("%s %d" % (o, n) for o, n in c.most_common(14)))
    print("(static counts: both sides of every branch are included; the loop region also holds its preheader and exit blocks)")


if __name__ == "__main__":
    main()
