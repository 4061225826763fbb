// k_resprop_split.hip — the split-schedule instantiations of k_resprop (kernels/resprop.hpp): the translation unit k_resprop.hip compiled a
// second time with the constants of the shared polynomials (llpf_horner) as SGPR pairs.  See the comment at LLPF_RESPROP_SPLIT_TU there.
#define LLPF_RESPROP_SPLIT_TU 1
#define LLPF_HORNER_C(c) "s"(c)
#include "k_resprop.hip"
