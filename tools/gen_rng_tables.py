"""Prints the tables of csrc/shared/llpf_rngmath.h (log centres 0.75 + i/64 and 64 base angles), computed in 80-bit
long double and rounded to double.  The tables are part of the definition of the noise stream: regenerate only
together with the golden fixtures."""
import numpy as np
LD = np.longdouble
c = [0.75 + i / 64.0 for i in range(49)]
print("invc", [repr(float(LD(1) / LD(x))) for x in c])
print("lnc", [repr(float(np.log(LD(x)))) for x in c])
pi = LD("3.14159265358979323846264338327950288")
print("sin", [repr(float(np.sin(LD(2) * pi * LD(j) / LD(64)))) for j in range(65)])
print("cos", [repr(float(np.cos(LD(2) * pi * LD(j) / LD(64)))) for j in range(65)])
