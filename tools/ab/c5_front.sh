#!/bin/bash
# What the generator and the RK4 cost INSIDE k_rbfull (timing only, the variants do not compute the filter): lib_c5_nonoise.so (no
# generator), lib_c5_ldnoise.so (normals read from memory as if precomputed by an earlier launch), lib_c5_nodyn.so (no RK4),
# lib_c5_nofront.so (neither) — built with tools/ab/build_variant.sh <name> k_rbfull -DLLPF_RBF_ABL_NOISE=1|2 / -DLLPF_RBF_ABL_DYN=1
tools/ab/c5_libs.sh lib_c5_nonoise.so lib_c5_ldnoise.so lib_c5_nodyn.so lib_c5_nofront.so
