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

}  // namespace kzg
