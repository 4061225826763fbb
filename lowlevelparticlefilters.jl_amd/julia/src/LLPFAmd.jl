# Package entry point (Pkg expects src/<Name>.jl): the module itself is ../LLPFAmd.jl, next to which the engine's shared library is looked
# for (`const LIB`, overridable with ENV["LLPF_HIP_LIB"]).  `] dev lowlevelparticlefilters.jl_amd/julia` makes `using LLPFAmd` work.
include(joinpath(@__DIR__, "..", "LLPFAmd.jl"))
