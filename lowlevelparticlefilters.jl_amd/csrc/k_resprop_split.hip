// k_resprop_split.hip — the split-schedule instantiations of k_resprop (kernels/resprop.hpp): the translation unit k_resprop.hip compiled a
// second time with the constants of the shared polynomials (llpf_horner) as SGPR pairs.  See the comment at LLPF_RESPROP_SPLIT_TU there.
#define LLPF_RESPROP_SPLIT_TU 1
#define LLPF_HORNER_C(c) "s"(c)
// The split schedule is the one of filters and banks beyond 3 M particles (host/run.hpp): their planes do not fit the Infinity Cache, and
// what the write-through stores of the output loop buy a 10^6-particle filter (no write-back of dirty L2 lines at the kernel boundary)
// they lose here — same box, us per timestep of ONE filter, write-through / plain / nontemporal stores: N = 4e6 63.5 / 60.0 / 75.8,
// N = 1.6e7 272 / 240 / 249; plain stores with the sources of two loop rounds requested together (LLPF_RESPROP_PF): 60.4, 236; one
// GPU's share of C4 (128 x 1e5): 194 / 193 / 186 (profiles/r06_bign_store_policy_ab.txt)
#ifndef LLPF_RESPROP_ST
#define LLPF_RESPROP_ST 0
#endif
// ... and by kind of step (same box, tools/ab/bign_store_matrix.sh, us per timestep, plain / nontemporal / write-through): a step that
// resamples — N = 1.6e7 252.7 / 261.8 / 279.4, C4 share 217.3 / 217.5 / 232.2; a step that does not (every output reads its own index,
// nothing a store leaves in the L2 is read again by the launch; 97 of 100 steps at the reference's threshold) — N = 1.6e7 237.5 / 234.7 /
// 241.6, C4 share 204.1 / 195.9 / 209.7: nontemporal stores on those (profiles/r06_bign_store_matrix.txt)
#ifndef LLPF_RESPROP_ST_ID
#define LLPF_RESPROP_ST_ID 2
#endif
// ... whose sources are read nontemporal as well (each is read once): N = 1.6e7 at threshold 0.1 230.0 -> 219.7, C4 share 192.2 -> 188.9,
// steps that resample unchanged (profiles/r06_bign_nt_loads_ab.txt)
#ifndef LLPF_RESPROP_LD_ID
#define LLPF_RESPROP_LD_ID 1
#endif
#include "k_resprop.hip"
