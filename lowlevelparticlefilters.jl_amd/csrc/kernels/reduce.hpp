// kernels/reduce.hpp — wave / block reductions and scans on DPP.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// wave / block reductions and scans (wave = 64 lanes) on DPP.
// Measured on gfx950 (tools/inst_cost.hip): a ds_bpermute_b32 (what __shfl_* compiles to) costs ~10 ns of a SIMD's
// time, a DPP-modified VALU move ~1 ns; a 64-lane reduction of one 64-bit value is 12 bpermutes vs 12 DPP moves.
// Row = 16 lanes.  Inclusive scan: row_shr 1,2,4,8 (Hillis–Steele inside a row, out-of-row sources read as the
// identity), then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3.  Lane 63 ends with the total.
// ------------------------------------------------------------------------------------------------
// Write-through store (global_store ... sc1).  The per-XCD L2s are not coherent with one another, so the end of every
// kernel writes back whatever the launch left dirty in L2 before the next launch may start; data a kernel produces for the
// NEXT launch (particles, weights, quanta, ancestors) is better written through while the kernel is still computing.
// Same-box A/B: C2 step 26.45 -> 25.63 us (nontemporal stores instead: 27.7 us), quad-tank +3 %, auxiliary filter +4 %;
// the Rao-Blackwellized propagate (5 planes per particle) and the auxiliary look-ahead lose 4 % with it and keep plain
// stores, k_norm / k_rbfull show no difference (plain).
typedef uint32_t llpf_u32x4 __attribute__((ext_vector_type(4)));
#ifndef LLPF_WT
#define LLPF_WT 1
#endif
template <bool WT = true, class T> DEV void wt_store(T* p, T v) {
    static_assert(sizeof(T) <= 8, "16 bytes at once: wt_store2");
#if LLPF_WT
    if constexpr (!WT) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
// Two consecutive 8-byte values, base[i] and base[i + 1], as ONE 16-byte write-through store (the two-particles-per-thread step kernel).
// Rounds 1-5 wrote this instruction by hand (asm "global_store_dwordx4 ... sc1"): the compiler cannot see into an asm block, so the
// wait states between the store and the next VALU write of its data registers were ours to keep — and were one short for gfx950
// (two are needed for a store of more than 64 bits; one wrong particle in ~600 runs of a 10^6-particle filter for four rounds,
// EXPERIMENTS 5.9).  Round 6: the same instruction class through a builtin the compiler schedules and pads itself — a raw buffer store
// (128 bits, cache policy sc1) on a descriptor made from the UNIFORM plane base, the lane's byte offset in the VGPR; `base` must be
// wave-uniform (a divergent one would be served by a waterfall loop).  Byte offsets fit 32 bits: Ns * 8 < 2^32 is checked at create.
template <bool WT = true, class T> DEV void wt_store2(T* base, int64_t i, T a, T b) {
    static_assert(sizeof(T) == 8, "two 8-byte values");
#if LLPF_WT && defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WT) {
        llpf_u32x4 r;
        __builtin_memcpy(&r, &a, 8);
        __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &b, 8);
        // dword 3 of the descriptor: raw buffer (no swizzle, no format conversion) as the compiler's own buffer accesses on gfx9 use it
        // (num_records = 2^32 - 1 bytes: offsets are compared UNSIGNED against it, and a plane of 3e8 particles reaches 2.4e9 — with
        //  2^31 - 1 the upper half of such a filter was silently dropped by the range check: tests/test_gpu_parity.py, near-maximum test)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)0xffffffffu, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(r, rs, (int)((uint32_t)i * 8u), 0, 16 /* sc1 */);
        return;
    }
#endif
    struct alignas(16) Pair { T x, y; };
    Pair v{a, b};
    *reinterpret_cast<Pair*>(base + i) = v;
}

#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_QUAD_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL, int ROW_MASK, bool BOUND>
DEV uint64_t dpp_u64(uint64_t old, uint64_t v) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, BOUND);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, BOUND);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
DEV uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix sum over the wave.  The add itself carries the DPP modifier (v_add_co_u32_dpp / v_addc_co_u32_dpp: two
// instructions per step for 64 bits, four for 128); written through update_dpp the compiler emits a v_mov_b32_dpp per half
// and then the add (four per step), and a 128-bit sum had to go limb-wise through three 64-bit scans.
// Hazards inside the block are ours to keep (the compiler does not see into it): a VALU write of a VGPR needs 2 wait states
// before a DPP read of it (s_nop 0 + the other half's instruction; s_nop 1 at the end for a DPP instruction of the compiler's
// that may follow), and the block starts with s_nop 4 (a VALU write of EXEC needs 5 before DPP).
#define LLPF_DPP_SHR(n) "row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define LLPF_DPP_B15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define LLPF_DPP_B31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
#define LLPF_ADD64_DPP(C) "v_add_co_u32_dpp %0, vcc, %0, %0 " C "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " C "\n\ts_nop 0\n\t"
#define LLPF_ADD128_DPP(C) "v_add_co_u32_dpp %0, vcc, %0, %0 " C "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " C "\n\t" \
                           "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc " C "\n\tv_addc_co_u32_dpp %3, vcc, %3, %3, vcc " C "\n\t"
DEV uint64_t wave_scan_u64(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    asm("s_nop 4\n\t"
        LLPF_ADD64_DPP(LLPF_DPP_SHR(1)) LLPF_ADD64_DPP(LLPF_DPP_SHR(2)) LLPF_ADD64_DPP(LLPF_DPP_SHR(4)) LLPF_ADD64_DPP(LLPF_DPP_SHR(8))
        LLPF_ADD64_DPP(LLPF_DPP_B15) LLPF_ADD64_DPP(LLPF_DPP_B31) "s_nop 1"
        : "+v"(lo), "+v"(hi) : : "vcc");
    return ((uint64_t)hi << 32) | lo;
}
DEV uint32_t wave_scan_max_u32(uint32_t x) {
#define LLPF_MAXSTEP(CTRL, RM, BC) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, RM, 0xF, BC); x = t > x ? t : x; }
    LLPF_MAXSTEP(DPP_ROW_SHR(1), 0xF, true) LLPF_MAXSTEP(DPP_ROW_SHR(2), 0xF, true) LLPF_MAXSTEP(DPP_ROW_SHR(4), 0xF, true)
    LLPF_MAXSTEP(DPP_ROW_SHR(8), 0xF, true) LLPF_MAXSTEP(DPP_ROW_BCAST15, 0xA, false) LLPF_MAXSTEP(DPP_ROW_BCAST31, 0xC, false)
#undef LLPF_MAXSTEP
    return x;
}
// total over the wave, returned uniformly to every lane
DEV uint64_t wave_sum_u64(uint64_t v) { return readlane_u64(wave_scan_u64(v), 63); }
DEV llpf_u128 wave_sum_u128(llpf_u128 v) {
    uint32_t w0 = (uint32_t)v.lo, w1 = (uint32_t)(v.lo >> 32), w2 = (uint32_t)v.hi, w3 = (uint32_t)(v.hi >> 32);
    asm("s_nop 4\n\t"
        LLPF_ADD128_DPP(LLPF_DPP_SHR(1)) LLPF_ADD128_DPP(LLPF_DPP_SHR(2)) LLPF_ADD128_DPP(LLPF_DPP_SHR(4)) LLPF_ADD128_DPP(LLPF_DPP_SHR(8))
        LLPF_ADD128_DPP(LLPF_DPP_B15) LLPF_ADD128_DPP(LLPF_DPP_B31) "s_nop 1"
        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : : "vcc");
    llpf_u128 r;
    r.lo = readlane_u64(((uint64_t)w1 << 32) | w0, 63);
    r.hi = readlane_u64(((uint64_t)w3 << 32) | w2, 63);
    return r;
}
DEV double wave_max(double v) {
    // running maximum with the same DPP sequence; out-of-row / masked lanes read the lane's own value
    uint64_t x = llpf_d2u(v);
#define LLPF_FMAXSTEP(CTRL, RM) { const double t = llpf_u2d(dpp_u64<CTRL, RM, false>(x, x)); const double c = llpf_u2d(x); x = llpf_d2u(llpf_fmax(c, t)); }
    LLPF_FMAXSTEP(DPP_ROW_SHR(1), 0xF) LLPF_FMAXSTEP(DPP_ROW_SHR(2), 0xF) LLPF_FMAXSTEP(DPP_ROW_SHR(4), 0xF)
    LLPF_FMAXSTEP(DPP_ROW_SHR(8), 0xF) LLPF_FMAXSTEP(DPP_ROW_BCAST15, 0xA) LLPF_FMAXSTEP(DPP_ROW_BCAST31, 0xC)
#undef LLPF_FMAXSTEP
    return llpf_u2d(readlane_u64(x, 63));
}
// fixed-order fp64 sum over the wave (used only for the weighted-mean output, never fed back)
DEV double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}

DEV double block_max(double v, double* sm /* [4] */) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int k = 1; k < BLOCK / 64; ++k) r = llpf_fmax(r, sm[k]);
    return r;
}
