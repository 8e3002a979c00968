// fr_fft4096.hpp -- the 4096-point (I)FFT over F_r as SIX radix-4 passes on lazily reduced 29-bit limbs (fr_lazy.hpp), one
// workgroup of 1024 lanes per transform, the data resident in LDS between passes.  Replaces the twelve radix-2 stages of
// fft_fr.go:30-53 (same values: the transform is unique; decimation in time on bit-reversed input, natural output).
//
// Pass with stride m = 1, 4, 16, 64, 256, 1024 fuses the radix-2 stages of half-size m and 2 m: lane-local unit on the positions
// p, p + m, p + 2 m, p + 3 m (p = 4 m g + j, j < m) with the twiddles w1 = w_{2m}^j (both first-stage butterflies),
// w2 = w_{4m}^j, w3 = w_{4m}^{j+m}: four products, like the two radix-2 stages it replaces, but half the LDS traffic and barriers.
//   * pass 1 runs on the values as they come from global memory: lane t loads the natural indices t + 1024 q -- coalesced --
//     which ARE the four bit-reversed positions 4 u + (0, 2, 1, 3), u = bitrev10(t): no bit-reversal pass, one product (w_4).
//   * the last pass leaves natural order in registers, position t + 1024 q: canonicalised and stored coalesced, never written to LDS.
//   * LDS layout: limb-major, position p = 64 hi + lo at hi * 65 + lo (row pitch 65).  Passes with stride >= 64 run their lanes along
//     lo, passes with stride < 64 along hi: bank (hi + lo) mod 32 either way -- every LDS access of the kernel is conflict-free.
//     In the passes with stride < 64 all lanes of a wavefront share (g, j), so their twiddles are wave-uniform (scalar loads).
//   * no reduction anywhere: limbs stay raw (< 6 * 2^29), bounds grow by <= 6 per pass from 8 after the first: <= 38 < 64.
// The functions are __host__ __device__ so that tests/host/host_emul.cpp runs the same passes lane by lane against the oracle.
#pragma once
#include "fr_lazy.hpp"

namespace kzg {
namespace fr4 {

static constexpr uint32_t N = 4096, NPAD = 64 * 65, LDS_BYTES = 9 * NPAD * 4;
// twiddle file (u32 words, one per direction): uniform entries [e][which][9] for the strides 1 (e = 0), 4 (e = 1 + j), 16 (e = 5 + j),
// then per-lane files [which][limb][j] for the strides 64, 256, 1024
static constexpr uint32_t TW_U = 0, TW_V64 = 576, TW_V256 = TW_V64 + 27 * 64, TW_V1024 = TW_V256 + 27 * 256, TW_WORDS = TW_V1024 + 27 * 1024;

KZG_HD uint32_t addr(uint32_t p) { return (p >> 6) * 65u + (p & 63u); }
KZG_HD frl get(const uint32_t *s, uint32_t p) {
    const uint32_t a = addr(p);
    frl v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = s[k * NPAD + a];
    return v;
}
KZG_HD void put(uint32_t *s, uint32_t p, const frl &v) {
    const uint32_t a = addr(p);
#pragma unroll
    for (int k = 0; k < 9; k++) s[k * NPAD + a] = v.l[k];
}
KZG_HD frl tw_u(const uint32_t *tw, uint32_t e, int which) {
    frl v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = tw[TW_U + (e * 3 + which) * 9 + k];
    return v;
}
KZG_HD frl tw_v(const uint32_t *tw, uint32_t base, uint32_t m, int which, uint32_t j) {
    frl v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = tw[base + (which * 9 + k) * m + j];
    return v;
}

// the unit: x0, x2 any raw (swept here), x1, x3 raw < 6 * 2^29.  Out: raw limbs < 3, 4, 4, 5 (* 2^29), bounds + 4, 5, 5, 6.
KZG_HD void unit(frl &x0, frl &x1, frl &x2, frl &x3, const frl &w1, const frl &w2, const frl &w3) {
    frl_sweep(x0); frl_sweep(x2);
    const frl t1 = frl_mul(x1, w1), t3 = frl_mul(x3, w1);
    const frl a0 = frl_add(x0, t1), a1 = frl_sub<3>(x0, t1), a2 = frl_add(x2, t3), a3 = frl_sub<3>(x2, t3);
    const frl t2 = frl_mul(a2, w2), t4 = frl_mul(a3, w3);
    x0 = frl_add(a0, t2); x2 = frl_sub<3>(a0, t2);
    x1 = frl_add(a1, t4); x3 = frl_sub<3>(a1, t4);
}

// pass 1 (stride 1): lane t, natural inputs t + 1024 q (zero beyond n_in) -> positions 4 u + 0..3, u = bitrev10(t)
// (es, off: element i of this transform is src[off + i es] -- a row of a longer transform is every es-th element of it: k_fr_fft_upper)
KZG_HD void pass_first(uint32_t t, const fr *src, uint64_t n_in, uint32_t *s, const uint32_t *tw, uint64_t es = 1, uint64_t off = 0) {
    frl x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t i = off + (uint64_t)(t + 1024u * q) * es;
        x[q] = (i < n_in) ? frl_unpack(src[i]) : frl_zero();
    }
    // position offset o holds the natural quarter bitrev2(o): X0 = x[0], X1 = x[2], X2 = x[1], X3 = x[3]; twiddles 1, 1, w_4
    const frl &X0 = x[0], &X1 = x[2], &X2 = x[1], &X3 = x[3];
    const frl n2 = frl_sub<2>(frl_zero(), X2), n3 = frl_sub<2>(frl_zero(), X3);           // 2 r - X (inputs are canonical): limbs < 2^30
    const frl s01 = frl_add(X0, X1);
    const frl b0 = frl_add(s01, frl_add(X2, X3));                                             // < 4 L, bound 4
    const frl b2 = frl_add(s01, frl_add(n2, n3));                                             // < 6 L, bound 6
    const frl a1 = frl_add(X0, frl_sub<2>(frl_zero(), X1));                                   // < 3 L, bound 3
    const frl a3 = frl_add(X2, n3);                                                           // < 3 L, bound 3
    const frl tq = frl_mul(a3, tw_u(tw, 0, 2));
    const frl b1 = frl_add(a1, tq), b3 = frl_sub<3>(a1, tq);                                    // < 4 L / 5 L, bounds 5 / 6
    uint32_t u = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) u |= ((t >> k) & 1u) << (9 - k);
    put(s, 4 * u + 0, b0); put(s, 4 * u + 1, b1); put(s, 4 * u + 2, b2); put(s, 4 * u + 3, b3);
}
// strides 4 and 16: lanes along hi, (g, j) from the wavefront index a (wave-uniform twiddles)
template <uint32_t M> KZG_HD void pass_lo(uint32_t a, uint32_t b, uint32_t *s, const uint32_t *tw) {
    const uint32_t j = a & (M - 1), g = a / M, e = (M == 4 ? 1u : 5u) + j;
    const uint32_t p = 64 * b + 4 * M * g + j;
    frl x0 = get(s, p), x1 = get(s, p + M), x2 = get(s, p + 2 * M), x3 = get(s, p + 3 * M);
    unit(x0, x1, x2, x3, tw_u(tw, e, 0), tw_u(tw, e, 1), tw_u(tw, e, 2));
    put(s, p, x0); put(s, p + M, x1); put(s, p + 2 * M, x2); put(s, p + 3 * M, x3);
}
// strides 64 and 256: lanes along lo
template <uint32_t M> KZG_HD void pass_hi(uint32_t t, uint32_t *s, const uint32_t *tw) {
    const uint32_t j = t & (M - 1), g = t / M, base = (M == 64 ? TW_V64 : TW_V256);
    const uint32_t p = 4 * M * g + j;
    frl x0 = get(s, p), x1 = get(s, p + M), x2 = get(s, p + 2 * M), x3 = get(s, p + 3 * M);
    unit(x0, x1, x2, x3, tw_v(tw, base, M, 0, j), tw_v(tw, base, M, 1, j), tw_v(tw, base, M, 2, j));
    put(s, p, x0); put(s, p + M, x1); put(s, p + 2 * M, x2); put(s, p + 3 * M, x3);
}
// last pass (stride 1024): outputs t + 1024 q in natural order, canonical; SCALE: multiplied by the constant sc (image 2^261) first
template <bool SCALE> KZG_HD void pass_last(uint32_t t, const uint32_t *s, const uint32_t *tw, const frl &sc, fr *dst) {
    frl x0 = get(s, t), x1 = get(s, t + 1024), x2 = get(s, t + 2048), x3 = get(s, t + 3072);
    unit(x0, x1, x2, x3, tw_v(tw, TW_V1024, 1024, 0, t), tw_v(tw, TW_V1024, 1024, 1, t), tw_v(tw, TW_V1024, 1024, 2, t));
    if (SCALE) {
        dst[t] = frl_canon_lt2r(frl_mul(x0, sc)); dst[t + 1024] = frl_canon_lt2r(frl_mul(x1, sc));
        dst[t + 2048] = frl_canon_lt2r(frl_mul(x2, sc)); dst[t + 3072] = frl_canon_lt2r(frl_mul(x3, sc));
    } else {
        dst[t] = frl_canon(x0); dst[t + 1024] = frl_canon(x1); dst[t + 2048] = frl_canon(x2); dst[t + 3072] = frl_canon(x3);
    }
}

// ---- transforms of m = 4 .. 2048 points, 4096 / m of them per workgroup ----
// The first stages of the 4096-point network ARE independent m-point transforms on the consecutive blocks of m positions (a stage of half-size
// h uses w_{2h}^j whatever the total length), so a workgroup runs log4(m) of the passes above on 4096 / m transforms at once; m = 2 * 4^a adds
// one radix-2 pass of half-size 4^a, whose twiddle w_{2h}^j is the first entry (w1) of the radix-4 table of stride h.
// first pass: unit U = lane, positions 4 U + o of block U / (m / 4); they hold the natural indices v + (m / 4) bitrev2(o), v = bitrev(U mod (m / 4))
template <int LOGM> KZG_HD void pass_first_small(uint32_t t, const fr *in, uint64_t in_stride, uint64_t n_in, uint64_t first, uint64_t batch, uint32_t *s,
                                                const uint32_t *tw) {
    constexpr uint32_t m = 1u << LOGM, q4 = m / 4;
    const uint32_t blk = t / q4, u = t % q4;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < LOGM - 2; k++) v |= ((u >> k) & 1u) << (LOGM - 3 - k);
    const bool live = first + blk < batch;
    const fr *src = in + (first + blk) * in_stride;
    frl x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t i = v + q4 * q;
        x[q] = (live && i < n_in) ? frl_unpack(src[i]) : frl_zero();
    }
    const frl &X0 = x[0], &X1 = x[2], &X2 = x[1], &X3 = x[3];                                 // as in pass_first
    const frl n2 = frl_sub<2>(frl_zero(), X2), n3 = frl_sub<2>(frl_zero(), X3);
    const frl s01 = frl_add(X0, X1);
    const frl b0 = frl_add(s01, frl_add(X2, X3));
    const frl b2 = frl_add(s01, frl_add(n2, n3));
    const frl a1 = frl_add(X0, frl_sub<2>(frl_zero(), X1));
    const frl a3 = frl_add(X2, n3);
    const frl tq = frl_mul(a3, tw_u(tw, 0, 2));
    const frl b1 = frl_add(a1, tq), b3 = frl_sub<3>(a1, tq);
    put(s, 4 * t + 0, b0); put(s, 4 * t + 1, b1); put(s, 4 * t + 2, b2); put(s, 4 * t + 3, b3);
}
// radix-2 pass of half-size H (4, 16: lanes along hi, wave-uniform twiddle; 64, 256, 1024: lanes along lo): two butterflies per lane
template <uint32_t H> KZG_HD void pass_r2(uint32_t t, uint32_t a, uint32_t b, uint32_t *s, const uint32_t *tw) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
        uint32_t p; frl w;
        if (H < 64) {
            const uint32_t c = a + 16u * r, j = c & (H - 1), g = c / H;
            p = 64 * b + 2 * H * g + j;
            w = tw_u(tw, (H == 4 ? 1u : 5u) + j, 0);
        } else {
            const uint32_t e = t + 1024u * r, j = e & (H - 1), g = e / H;
            p = 2 * H * g + j;
            w = tw_v(tw, H == 64 ? TW_V64 : (H == 256 ? TW_V256 : TW_V1024), H, 0, j);
        }
        frl x0 = get(s, p);
        const frl y = get(s, p + H);
        frl_sweep(x0);
        const frl tq = frl_mul(y, w);
        put(s, p, frl_add(x0, tq)); put(s, p + H, frl_sub<3>(x0, tq));
    }
}
// positions t + 1024 q -> canonical values (SCALE: times the constant sc first); position p of the workgroup is element p of its 4096 / m transforms
template <bool SCALE> KZG_HD void pass_store(uint32_t t, const uint32_t *s, const frl &sc, fr *dst, uint64_t limit) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t p = t + 1024u * q;
        if (p >= limit) continue;
        frl v = get(s, p);
        if (SCALE) { frl_sweep(v); dst[p] = frl_canon_lt2r(frl_mul(v, sc)); }
        else dst[p] = frl_canon(v);
    }
}

// ---- transforms of R * 4096 points, R = 2 .. 16: the upper stages in registers (the rows come from the passes above with es = R) ----
#define LOGR_OF(R) ((R) == 2 ? 1 : (R) == 4 ? 2 : (R) == 8 ? 3 : 4)
// The stages above 4096 of a transform of R * 4096 points (R = 2, 4, 8, 16), all in one pass over the data: lane k2 holds the R values
// k2 + 4096 a in registers and runs the log2 R radix-2 stages of half-size m = 4096 2^s on them (fft_fr.go:40-52 with the lazy limbs of
// fr_lazy.hpp; twiddle w_{2m}^j, j = (a mod 2^s) 4096 + k2, from the settings' roots pre-scaled to the 2^261 image), then canonicalises
// (SCALE: after the product with 1/n).  Bounds: a stage adds 2 (sum) / 3 (difference) to a bound and 2^29 / 2 * 2^29 to the limbs; from the
// fourth stage on both operands are swept first, so every product sees limbs < 6 * 2^29 and no limb passes 2^32; final bounds <= 13.
// (compile-time recursion instead of loops: every index into x[] must be a constant for the array to live in registers, and the
// unroller gives up on loops whose bodies hold a 153-multiply-add product each)
template <int R, int S, int B = 0> KZG_HD void upper_stage(frl (&x)[R], const fr *roots_l, uint64_t W, uint32_t k2, frl &w) {
    if constexpr (B < R / 2) {
        constexpr int jh = B >> (LOGR_OF(R) - 1 - S), hi = B & ((R >> (S + 1)) - 1), a = (hi << (S + 1)) + jh;   // butterflies ordered by twiddle
        if constexpr (hi == 0) w = frl_unpack(roots_l[((uint64_t)jh * N + k2) * (W / ((uint64_t)N << (S + 1)))]);
        frl x0 = x[a], y = x[a + (1 << S)];
        if (S >= 3) { frl_sweep(x0); frl_sweep(y); }
        const frl tq = frl_mul(y, w);
        x[a] = frl_add(x0, tq);
        x[a + (1 << S)] = frl_sub<3>(x0, tq);
        upper_stage<R, S, B + 1>(x, roots_l, W, k2, w);
    }
}
template <int R, int A = 0> KZG_HD void upper_load(frl (&x)[R], const fr *base, uint32_t k2) {
    if constexpr (A < R) { x[A] = frl_unpack(base[(uint64_t)A * N + k2]); upper_load<R, A + 1>(x, base, k2); }
}
template <int R, bool SCALE, int A = 0> KZG_HD void upper_store(frl (&x)[R], fr *base, uint32_t k2, const frl &sc) {
    if constexpr (A < R) {
        if (SCALE) { frl v = x[A]; frl_sweep(v); base[(uint64_t)A * N + k2] = frl_canon_lt2r(frl_mul(v, sc)); }
        else base[(uint64_t)A * N + k2] = frl_canon(x[A]);
        upper_store<R, SCALE, A + 1>(x, base, k2, sc);
    }
}
// one lane of k_fr_fft_upper: the R values k2 + 4096 a of the transform at `base`
template <int LOGR, bool SCALE> KZG_HD void upper_lane(fr *base, uint32_t k2, const fr *roots_l, uint64_t W, const fr *scale) {
    constexpr int R = 1 << LOGR;
    frl x[R], w;
    upper_load<R>(x, base, k2);
    upper_stage<R, 0>(x, roots_l, W, k2, w);
    if constexpr (LOGR > 1) upper_stage<R, 1>(x, roots_l, W, k2, w);
    if constexpr (LOGR > 2) upper_stage<R, 2>(x, roots_l, W, k2, w);
    if constexpr (LOGR > 3) upper_stage<R, 3>(x, roots_l, W, k2, w);
    frl sc = frl_zero();
    if (SCALE) sc = frl_const_from_kilic(*scale);
    upper_store<R, SCALE>(x, base, k2, sc);
}

// host side: the twiddle file from a root table of width W (roots[i] = w_W^i, Kilic images; ExpandedRootsOfUnity for the
// forward transform, ReverseRootsOfUnity for the inverse)
inline void build_twiddles(const fr *roots, uint64_t W, uint32_t *out) {
    for (uint32_t i = 0; i < TW_WORDS; i++) out[i] = 0;
    const uint32_t ms[6] = {1, 4, 16, 64, 256, 1024}, vbase[6] = {0, 0, 0, TW_V64, TW_V256, TW_V1024}, ubase[3] = {0, 1, 5};
    for (int pi = 0; pi < 6; pi++) {
        const uint32_t m = ms[pi];
        for (uint32_t j = 0; j < m; j++) {
            // (a settings object narrower than 4096 only runs the passes whose roots it has: 2 m <= W for w1, 4 m <= W for w2 and w3)
            if (2ull * m > W) continue;
            const bool wide = 4ull * m <= W;
            const fr w[3] = {roots[(uint64_t)j * (W / (2 * m))], wide ? roots[(uint64_t)j * (W / (4 * m))] : roots[0],
                             wide ? roots[(uint64_t)(j + m) * (W / (4 * m))] : roots[0]};
            for (int which = 0; which < 3; which++) {
                const frl c = frl_const_from_kilic(w[which]);
                for (int k = 0; k < 9; k++) {
                    if (pi < 3) out[TW_U + ((ubase[pi] + j) * 3 + which) * 9 + k] = c.l[k];
                    else out[vbase[pi] + (which * 9 + k) * m + j] = c.l[k];
                }
            }
        }
    }
}

}  // namespace fr4

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
