// fr16.hpp -- A/B ARTEFACT, not part of the library (round 6; lived in go-kzg_amd/csrc/fr_fft4096.hpp in round 5).
//
// The 256-lane x 16-register form of the 4096-point F_r transform.  It is bit-exact (tests/test_host_arith.py emulates it lane by lane against the oracle) and
// 25 % SLOWER than k_fr_fft4096_r4 on the MI355X (profiles/r05_fr_fft_ab.md: 253 spill stores + 394 loads per lane at the 128 registers two workgroups per CU leave;
// the 512-lane x 8-value shape needs 72 + 67 registers and does not fit 128 either -- profiles/r06_fr_r16_fate.md has the register report).  A second implementation
// of one transform that only an environment variable reaches is a maintenance cost, so the library ships k_fr_fft4096_r4 alone; this header, the kernel and a
// stand-alone harness (r16_ab.hip: same inputs through this kernel and through the library, results compared, both timed) stay here so that the measurement can be
// repeated:   tools/ab_fr_r16/build.sh && tools/ab_fr_r16/r16_ab
#pragma once
#include "../../go-kzg_amd/csrc/fr_fft4096.hpp"

namespace kzg {

// ---------------------------------------------------------------------------------------------------------
// The same network on 256 lanes x 16 values (round 5): what bounds k_fr_fft4096_r4 is not arithmetic but that ONE 1024-lane workgroup owns a
// CU (146 KiB of LDS): its global loads, its stores and its five barriers overlap with nothing (41 % of the wave-cycles parked).  Here a lane
// keeps 16 values in registers and runs FOUR radix-2 stages on them per pass -- the very same units with the very same twiddle file as fr4
// (pass A = fr4's strides 1 and 4, pass B = 16 and 64, pass C = 256 and 1024), so bounds and values are unchanged -- and the data crosses
// lanes only twice, through an LDS area that holds HALF of the transform at a time: 72 KiB per workgroup, TWO workgroups per CU, one computing
// while the other loads, stores or waits.
//   positions p = 256 g + 16 k' + k (three hex digits).  pass A: lane <-> block (g, k'), registers k;  pass B: lane <-> (g, k), registers k';
//   pass C: lane <-> (k', k), registers g.
//   transposition A -> B in two stages, stage s = the positions with bit 3 xor bit 7 = s: every lane WRITES 8 of its registers and then READS 8
//   new ones per stage (which 8 is wave-uniform: the lane maps below put the deciding bit into the wavefront index), so a lane never holds more
//   than 16 values; B -> C likewise with bit 7 xor bit 11.  LDS addresses a1 / a2: five bank bits chosen so that the 32 lanes of a half
//   wavefront hit 32 banks both when writing and when reading (limb-major, one word per lane and access).
// ---------------------------------------------------------------------------------------------------------
namespace fr16 {

static constexpr uint32_t LANES = 256, HALF = 2048, LDS_BYTES = 9 * HALF * 4;

KZG_HD uint32_t a1(uint32_t p) {   // stage region of the A -> B transposition: 11 address bits of a position with bit3 ^ bit7 fixed
    const uint32_t bank = ((p >> 8) & 7u) | ((((p >> 11) ^ p) & 1u) << 3) | ((((p >> 6) ^ (p >> 1)) & 1u) << 4);
    return (((p >> 7) & 1u) << 10) | (((p >> 4) & 3u) << 8) | ((p & 7u) << 5) | bank;
}
KZG_HD uint32_t a2(uint32_t p) {   // ... of the B -> C transposition (bit7 ^ bit11 fixed)
    const uint32_t bank = (p & 3u) | ((((p >> 2) ^ (p >> 8)) & 7u) << 2);
    return (((p >> 11) & 1u) << 10) | (((p >> 5) & 3u) << 8) | (((p >> 8) & 7u) << 5) | bank;
}
template <int WHICH> KZG_HD void put(uint32_t *s, uint32_t p, const frl &v) {
    const uint32_t a = WHICH == 1 ? a1(p) : a2(p);
#pragma unroll
    for (int k = 0; k < 9; k++) s[k * HALF + a] = v.l[k];
}
template <int WHICH> KZG_HD frl get(const uint32_t *s, uint32_t p) {
    const uint32_t a = WHICH == 1 ? a1(p) : a2(p);
    frl v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = s[k * HALF + a];
    return v;
}
// registers BASE .. BASE + 7 of a lane <-> positions p0 + stride * register
// (compile-time recursion, not loops: every index into a lane's register array must be a constant BEFORE the optimiser decides where the array
// lives -- with loops, even fully unrollable ones, both arrays of the kernel ended up in scratch: 900 scratch stores and loads per lane)
template <int WHICH, int BASE, int I = 0> KZG_HD void put8(uint32_t *s, const frl (&v)[16], uint32_t p0, uint32_t stride) {
    if constexpr (I < 8) { put<WHICH>(s, p0 + stride * (uint32_t)(BASE + I), v[BASE + I]); put8<WHICH, BASE, I + 1>(s, v, p0, stride); }
}
template <int WHICH, int BASE, int I = 0> KZG_HD void get8(const uint32_t *s, frl (&v)[16], uint32_t p0, uint32_t stride) {
    if constexpr (I < 8) { v[BASE + I] = get<WHICH>(s, p0 + stride * (uint32_t)(BASE + I)); get8<WHICH, BASE, I + 1>(s, v, p0, stride); }
}

// lane maps (t = 0 .. 255, wavefront w = t >> 6, l = t & 63).  The bit that decides which 8 registers a lane moves in a stage sits in w.
KZG_HD uint32_t lane_a_nat(uint32_t t) {   // pass A: the natural index n (< 256) whose column n + 256 q the lane loads; n4 = w0, n7 = w1: a wavefront covers four runs of 16 elements
    const uint32_t w = t >> 6, l = t & 63u;
    return ((w >> 1) << 7) | ((l >> 4) << 5) | ((w & 1u) << 4) | (l & 15u);
}
KZG_HD uint32_t bitrev8(uint32_t n) {
    uint32_t u = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) u |= ((n >> k) & 1u) << (7 - k);
    return u;
}
KZG_HD void lane_b(uint32_t t, uint32_t &g, uint32_t &j) {   // pass B: g3 = w0, j3 = w1
    const uint32_t w = t >> 6, l = t & 63u;
    g = ((w & 1u) << 3) | (l & 7u);
    j = ((w >> 1) << 3) | (l >> 3);
}
// (pass C: lane t <-> position t + 256 register)

// the first two stages on canonical inputs (fr4::pass_first's unit): X0..X3 at position offsets 0..3, twiddles 1, 1, w_4
KZG_HD void first_unit(const frl &X0, const frl &X1, const frl &X2, const frl &X3, const frl &w4, frl &b0, frl &b1, frl &b2, frl &b3) {
    const frl n2 = frl_sub<2>(frl_zero(), X2), n3 = frl_sub<2>(frl_zero(), X3);
    const frl s01 = frl_add(X0, X1);
    b0 = frl_add(s01, frl_add(X2, X3));
    b2 = frl_add(s01, frl_add(n2, n3));
    const frl a1_ = frl_add(X0, frl_sub<2>(frl_zero(), X1));
    const frl a3_ = frl_add(X2, n3);
    const frl tq = frl_mul(a3_, w4);
    b1 = frl_add(a1_, tq); b3 = frl_sub<3>(a1_, tq);
}
#define KZG_R16_BR2(o) ((((o) & 1) << 1) | (((o) >> 1) & 1))
// between the units of a pass: keeps the twiddle loads of the LATER units from being hoisted above the earlier units (15 twiddles x 9 limbs
// beside the lane's 144 data registers: 392 spills without it)
#if defined(__HIP_DEVICE_COMPILE__)
#define KZG_R16_FENCE() asm volatile("" ::: "memory")
#else
#define KZG_R16_FENCE() ((void)0)
#endif
// pass A: x[q] = element n + 256 q of the transform (canonical), n = lane_a_nat(t)  ->  p[k] = position 16 bitrev8(n) + k after the stages of
// half-size 1, 2 (first_unit on the offsets 4 a + 0..3, which hold the natural quarters q = 4 bitrev2(o) + bitrev2(a)) and 4, 8 (fr4's stride-4 units)
// (the inputs stay PACKED, 8 words each, until their unit runs: 128 registers for the sixteen loads in flight instead of 144 + the unpacking temporaries)
KZG_HD void pass_a(const fr (&x)[16], frl (&p)[16], const uint32_t *tw) {
    const frl w4 = fr4::tw_u(tw, 0, 2);
#define KZG_R16_FIRST(a) first_unit(frl_unpack(x[4 * KZG_R16_BR2(0) + KZG_R16_BR2(a)]), frl_unpack(x[4 * KZG_R16_BR2(1) + KZG_R16_BR2(a)]), frl_unpack(x[4 * KZG_R16_BR2(2) + KZG_R16_BR2(a)]), \
                                    frl_unpack(x[4 * KZG_R16_BR2(3) + KZG_R16_BR2(a)]), w4, p[4 * (a)], p[4 * (a) + 1], p[4 * (a) + 2], p[4 * (a) + 3])
    KZG_R16_FIRST(0); KZG_R16_FENCE(); KZG_R16_FIRST(1); KZG_R16_FENCE(); KZG_R16_FIRST(2); KZG_R16_FENCE(); KZG_R16_FIRST(3); KZG_R16_FENCE();
#undef KZG_R16_FIRST
#define KZG_R16_U4(k) fr4::unit(p[k], p[(k) + 4], p[(k) + 8], p[(k) + 12], fr4::tw_u(tw, 1 + (k), 0), fr4::tw_u(tw, 1 + (k), 1), fr4::tw_u(tw, 1 + (k), 2))
    KZG_R16_U4(0); KZG_R16_FENCE(); KZG_R16_U4(1); KZG_R16_FENCE(); KZG_R16_U4(2); KZG_R16_FENCE(); KZG_R16_U4(3);
#undef KZG_R16_U4
}
// pass B: p[k'] = position 256 g + j + 16 k': fr4's stride-16 units (all four with the twiddles of j) and stride-64 units (j64 = j + 16 k)
KZG_HD void pass_b(frl (&p)[16], uint32_t j, const uint32_t *tw) {
    {
        const frl w1 = fr4::tw_u(tw, 5 + j, 0), w2 = fr4::tw_u(tw, 5 + j, 1), w3 = fr4::tw_u(tw, 5 + j, 2);
        fr4::unit(p[0], p[1], p[2], p[3], w1, w2, w3); fr4::unit(p[4], p[5], p[6], p[7], w1, w2, w3);
        fr4::unit(p[8], p[9], p[10], p[11], w1, w2, w3); fr4::unit(p[12], p[13], p[14], p[15], w1, w2, w3);
    }
#define KZG_R16_U64(k) fr4::unit(p[k], p[(k) + 4], p[(k) + 8], p[(k) + 12], fr4::tw_v(tw, fr4::TW_V64, 64, 0, j + 16 * (k)), fr4::tw_v(tw, fr4::TW_V64, 64, 1, j + 16 * (k)), \
                                 fr4::tw_v(tw, fr4::TW_V64, 64, 2, j + 16 * (k)))
    KZG_R16_FENCE(); KZG_R16_U64(0); KZG_R16_FENCE(); KZG_R16_U64(1); KZG_R16_FENCE(); KZG_R16_U64(2); KZG_R16_FENCE(); KZG_R16_U64(3);
#undef KZG_R16_U64
}
// pass C: p[g] = position t + 256 g: fr4's stride-256 units (twiddles of j = t) and stride-1024 units (j1024 = t + 256 k); natural order, raw
KZG_HD void pass_c(frl (&p)[16], uint32_t t, const uint32_t *tw) {
    {
        const frl w1 = fr4::tw_v(tw, fr4::TW_V256, 256, 0, t), w2 = fr4::tw_v(tw, fr4::TW_V256, 256, 1, t), w3 = fr4::tw_v(tw, fr4::TW_V256, 256, 2, t);
        fr4::unit(p[0], p[1], p[2], p[3], w1, w2, w3); fr4::unit(p[4], p[5], p[6], p[7], w1, w2, w3);
        fr4::unit(p[8], p[9], p[10], p[11], w1, w2, w3); fr4::unit(p[12], p[13], p[14], p[15], w1, w2, w3);
    }
#define KZG_R16_U1K(k) fr4::unit(p[k], p[(k) + 4], p[(k) + 8], p[(k) + 12], fr4::tw_v(tw, fr4::TW_V1024, 1024, 0, t + 256 * (k)), fr4::tw_v(tw, fr4::TW_V1024, 1024, 1, t + 256 * (k)), \
                                 fr4::tw_v(tw, fr4::TW_V1024, 1024, 2, t + 256 * (k)))
    KZG_R16_FENCE(); KZG_R16_U1K(0); KZG_R16_FENCE(); KZG_R16_U1K(1); KZG_R16_FENCE(); KZG_R16_U1K(2); KZG_R16_FENCE(); KZG_R16_U1K(3);
#undef KZG_R16_U1K
}
// the lane's 16 inputs: element off + (n + 256 q) es of the source (zero beyond n_in), n = lane_a_nat(t)
template <int Q = 0> KZG_HD void load_n(uint32_t n, const fr *src, uint64_t n_in, uint64_t es, uint64_t off, fr (&x)[16]) {
    if constexpr (Q < 16) {
        const uint64_t i = off + (uint64_t)(n + 256u * (uint32_t)Q) * es;
        x[Q] = src[i < n_in ? i : 0];                        // (an in-range address either way: the sixteen loads issue back to back, no branch)
        if (i >= n_in) x[Q] = zero<FrP>();
        load_n<Q + 1>(n, src, n_in, es, off, x);
    }
}
KZG_HD void load(uint32_t t, const fr *src, uint64_t n_in, uint64_t es, uint64_t off, fr (&x)[16]) { load_n<0>(lane_a_nat(t), src, n_in, es, off, x); }
template <bool SCALE, int G = 0> KZG_HD void store(uint32_t t, const frl (&p)[16], const frl &sc, fr *dst) {
    if constexpr (G < 16) {
        if (SCALE) dst[t + 256u * (uint32_t)G] = frl_canon_lt2r(frl_mul(p[G], sc));      // (a unit leaves limbs < 5 * 2^29: a valid left operand)
        else dst[t + 256u * (uint32_t)G] = frl_canon(p[G]);
        store<SCALE, G + 1>(t, p, sc, dst);
    }
}
// one stage of a transposition for one lane: WRITE (the 8 registers with bit 3 == sel) resp. READ
// (the two arms differ only in the register numbers; left alone, the optimiser merges them into ONE arm with selected POINTERS into the register
// array, which then has to live in scratch memory -- the distinct empty asm statements at the end of each arm keep them apart)
#if defined(__HIP_DEVICE_COMPILE__)
#define KZG_R16_KEEP_APART(tag) asm volatile("; fr16 arm " tag)
#else
#define KZG_R16_KEEP_APART(tag) ((void)0)
#endif
template <int WHICH> KZG_HD void stage_put(uint32_t *s, const frl (&v)[16], uint32_t p0, uint32_t stride, uint32_t sel) {
    if (sel) { put8<WHICH, 8>(s, v, p0, stride); KZG_R16_KEEP_APART("put hi"); } else { put8<WHICH, 0>(s, v, p0, stride); KZG_R16_KEEP_APART("put lo"); }
}
template <int WHICH> KZG_HD void stage_get(const uint32_t *s, frl (&v)[16], uint32_t p0, uint32_t stride, uint32_t sel) {
    if (sel) { get8<WHICH, 8>(s, v, p0, stride); KZG_R16_KEEP_APART("get hi"); } else { get8<WHICH, 0>(s, v, p0, stride); KZG_R16_KEEP_APART("get lo"); }
}

}  // namespace fr16

}  // namespace kzg
