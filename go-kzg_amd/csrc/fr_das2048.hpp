// fr_das2048.hpp -- DASFFTExtension of 2048 values (das_extension.go:7-84: the recursion unrolled into 11 "down" stages
// a0 = a + b, a1 = (a - b) rev[2 i s] for half-lengths h = 1024 .. 1, 11 "up" stages x +- y ex[(1 + 2 i) s] for h = 1 .. 1024,
// s = n / 2h, then a scale by 1/n) as ELEVEN lane-local passes on lazy 29-bit limbs (fr_lazy.hpp), one workgroup of 512 lanes per row,
// the row resident in LDS (74 KiB: two rows per CU):
//   5 radix-4 down passes (h, h/2) = (1024, 512) (256, 128) (64, 32) (16, 8) (4, 2); one pass for the two h = 1 stages (down then up on the
//   same pairs: one product per pair); 5 radix-4 up passes (h, 2h) = (2, 4) (8, 16) (32, 64) (128, 256) (512, 1024) -- the up unit is the
//   decimation-in-time unit of fr_fft4096.hpp with other twiddles.  First pass reads global memory, last pass scales, canonicalises and writes it.
// Table indices are those of the reference: straight into the full-width ExpandedRootsOfUnity / ReverseRootsOfUnity (das_extension.go:38,59).
// Down unit on x0..x3 at p, p + h/2, p + h, p + 3h/2 (i < h/2), inputs raw limbs < 2 * 2^29 with bound <= 4 (the LDS invariant of the down half):
//   c0 = x0 + x2, c1 = x1 + x3 (bound 8);  c2 = (x0 - x2) D_h[i], c3 = (x1 - x3) D_h[i + h/2];
//   d0 = reduce(c0 + c1) (bound 16 -> < 1.13 r: the only reduction of the unit; the sum path would double its bound every stage),
//   d1 = (c0 - c1) D_{h/2}[i], d2 = c2 + c3 (raw, bound 4), d3 = (c2 - c3) D_{h/2}[i].
// LDS: limb-major, rows of 32 positions at pitch 33; passes with stride >= 32 run their lanes along a row, the others along the rows:
// bank (row + column) mod 32 either way, conflict-free; in the passes along the rows the twiddles are wave-uniform.
#pragma once
#include "fr_fft4096.hpp"

namespace kzg {
namespace das2k {

static constexpr uint32_t N = 2048, NPAD = 64 * 33, LDS_BYTES = 9 * NPAD * 4, THREADS = 512;
// twiddle file: uniform entries [e][which][9]: D(16,8) e = i < 8; D(4,2) e = 8 + i; middle e = 10 (which 0); U(2,4) e = 11 + i; U(8,16) e = 13 + i;
// per-lane files [which][limb][i]: D(1024,512), D(256,128), D(64,32), U(32,64), U(128,256), U(512,1024)
static constexpr uint32_t E_D16 = 0, E_D4 = 8, E_MID = 10, E_U2 = 11, E_U8 = 13;
static constexpr uint32_t TW_D1024 = 576, TW_D256 = TW_D1024 + 27 * 512, TW_D64 = TW_D256 + 27 * 128, TW_U32 = TW_D64 + 27 * 32, TW_U128 = TW_U32 + 27 * 32,
                          TW_U512 = TW_U128 + 27 * 128, TW_WORDS = TW_U512 + 27 * 512;

KZG_HD uint32_t addr(uint32_t p) { return (p >> 5) * 33u + (p & 31u); }
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
    for (int k = 0; k < 9; k++) v.l[k] = tw[(e * 3 + which) * 9 + k];
    return v;
}
KZG_HD frl tw_v(const uint32_t *tw, uint32_t base, uint32_t m, int which, uint32_t j) {
    frl v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = tw[base + (which * 9 + k) * m + j];
    return v;
}

KZG_HD void down_unit(frl &x0, frl &x1, frl &x2, frl &x3, const frl &wa, const frl &wb, const frl &wc) {
    const frl c0 = frl_add(x0, x2), c1 = frl_add(x1, x3);                              // raw < 4 L, bound 8
    const frl c2 = frl_mul(frl_sub<5, 2>(x0, x2), wa), c3 = frl_mul(frl_sub<5, 2>(x1, x3), wb);   // operands raw < 5 L, bound 9
    frl c1s = c1;
    frl_sweep(c1s);
    x0 = frl_reduce(frl_add(c0, c1));                                                   // raw < 8 L, bound 16 -> normalised, < 1.13 r
    x1 = frl_mul(frl_sub<9>(c0, c1s), wc);                                              // operand raw < 6 L, bound 17
    x2 = frl_add(c2, c3);                                                               // raw < 2 L, bound 4
    x3 = frl_mul(frl_sub<3>(c2, c3), wc);
}
using fr4::unit;

// down (1024, 512): lane t = i, inputs from global memory (canonical), positions i + 512 q
KZG_HD void pass_down_first(uint32_t t, const fr *row, uint32_t *s, const uint32_t *tw) {
    frl x0 = frl_unpack(row[t]), x1 = frl_unpack(row[t + 512]), x2 = frl_unpack(row[t + 1024]), x3 = frl_unpack(row[t + 1536]);
    down_unit(x0, x1, x2, x3, tw_v(tw, TW_D1024, 512, 0, t), tw_v(tw, TW_D1024, 512, 1, t), tw_v(tw, TW_D1024, 512, 2, t));
    put(s, t, x0); put(s, t + 512, x1); put(s, t + 1024, x2); put(s, t + 1536, x3);
}
// down (2 Q, Q) with Q = 128 or 32: a block of 4 Q positions per Q lanes, lanes along a row
template <uint32_t Q> KZG_HD void pass_down_wide(uint32_t t, uint32_t *s, const uint32_t *tw) {
    const uint32_t i = t & (Q - 1), p = 4 * Q * (t / Q) + i, base = (Q == 128 ? TW_D256 : TW_D64);
    frl x0 = get(s, p), x1 = get(s, p + Q), x2 = get(s, p + 2 * Q), x3 = get(s, p + 3 * Q);
    down_unit(x0, x1, x2, x3, tw_v(tw, base, Q, 0, i), tw_v(tw, base, Q, 1, i), tw_v(tw, base, Q, 2, i));
    put(s, p, x0); put(s, p + Q, x1); put(s, p + 2 * Q, x2); put(s, p + 3 * Q, x3);
}
// down (16, 8): lane b = row, wavefront a = i;  down (4, 2): wavefront a = (quarter of the row, i)
template <uint32_t Q> KZG_HD void pass_down_narrow(uint32_t a, uint32_t b, uint32_t *s, const uint32_t *tw) {
    const uint32_t i = a & (Q - 1), p = 32 * b + 4 * Q * (a / Q) + i, e = (Q == 8 ? E_D16 : E_D4) + i;
    frl x0 = get(s, p), x1 = get(s, p + Q), x2 = get(s, p + 2 * Q), x3 = get(s, p + 3 * Q);
    down_unit(x0, x1, x2, x3, tw_u(tw, e, 0), tw_u(tw, e, 1), tw_u(tw, e, 2));
    put(s, p, x0); put(s, p + Q, x1); put(s, p + 2 * Q, x2); put(s, p + 3 * Q, x3);
}
// the two h = 1 stages on the pairs (2 g, 2 g + 1): (x, y) -> (x + y, x - y) -> (s + d u, s - d u), u = ex[n / 2]; two pairs per lane
KZG_HD void pass_middle(uint32_t a, uint32_t b, uint32_t *s, const uint32_t *tw) {
    const frl u = tw_u(tw, E_MID, 0);
    const uint32_t p = 32 * b + 4 * a;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const frl x = get(s, p + 2 * k), y = get(s, p + 2 * k + 1);                       // raw < 2 L, bound 4
        const frl sm = frl_add(x, y);                                                     // raw < 4 L, bound 8
        const frl tq = frl_mul(frl_sub<5, 2>(x, y), u);                                   // normalised, bound 2
        put(s, p + 2 * k, frl_add(sm, tq));                                               // raw < 5 L, bound 10
        put(s, p + 2 * k + 1, frl_sub<3>(sm, tq));                                        // raw < 6 L, bound 11
    }
}
// up (2, 4) and (8, 16): lanes along the rows
template <uint32_t H> KZG_HD void pass_up_narrow(uint32_t a, uint32_t b, uint32_t *s, const uint32_t *tw) {
    const uint32_t i = a & (H - 1), p = 32 * b + 4 * H * (a / H) + i, e = (H == 2 ? E_U2 : E_U8) + i;
    frl x0 = get(s, p), x1 = get(s, p + H), x2 = get(s, p + 2 * H), x3 = get(s, p + 3 * H);
    unit(x0, x1, x2, x3, tw_u(tw, e, 0), tw_u(tw, e, 1), tw_u(tw, e, 2));
    put(s, p, x0); put(s, p + H, x1); put(s, p + 2 * H, x2); put(s, p + 3 * H, x3);
}
// up (32, 64) and (128, 256): lanes along a row
template <uint32_t H> KZG_HD void pass_up_wide(uint32_t t, uint32_t *s, const uint32_t *tw) {
    const uint32_t i = t & (H - 1), p = 4 * H * (t / H) + i, base = (H == 32 ? TW_U32 : TW_U128);
    frl x0 = get(s, p), x1 = get(s, p + H), x2 = get(s, p + 2 * H), x3 = get(s, p + 3 * H);
    unit(x0, x1, x2, x3, tw_v(tw, base, H, 0, i), tw_v(tw, base, H, 1, i), tw_v(tw, base, H, 2, i));
    put(s, p, x0); put(s, p + H, x1); put(s, p + 2 * H, x2); put(s, p + 3 * H, x3);
}
// up (512, 1024): outputs i + 512 q times sc (the image 2^261 of 1 / n), canonical, to global memory
KZG_HD void pass_up_last(uint32_t t, const uint32_t *s, const uint32_t *tw, const frl &sc, fr *row) {
    frl x0 = get(s, t), x1 = get(s, t + 512), x2 = get(s, t + 1024), x3 = get(s, t + 1536);
    unit(x0, x1, x2, x3, tw_v(tw, TW_U512, 512, 0, t), tw_v(tw, TW_U512, 512, 1, t), tw_v(tw, TW_U512, 512, 2, t));
    row[t] = frl_canon_lt2r(frl_mul(x0, sc)); row[t + 512] = frl_canon_lt2r(frl_mul(x1, sc));
    row[t + 1024] = frl_canon_lt2r(frl_mul(x2, sc)); row[t + 1536] = frl_canon_lt2r(frl_mul(x3, sc));
}

// host side: twiddle file from the full-width tables (W + 1 entries each, Kilic images), W >= 2 n
inline void build_twiddles(const fr *expanded, const fr *reversed, uint64_t W, uint32_t *out) {
    (void)W;
    for (uint32_t i = 0; i < TW_WORDS; i++) out[i] = 0;
    auto D = [&](uint32_t h, uint32_t i) { return frl_const_from_kilic(reversed[(uint64_t)i * (N / h)]); };                  // rev[2 i s], s = n / 2h
    auto U = [&](uint32_t h, uint32_t i) { return frl_const_from_kilic(expanded[(uint64_t)(1 + 2 * i) * (N / (2 * h))]); };  // ex[(1 + 2 i) s]
    auto put_u = [&](uint32_t e, int which, const frl &c) { for (int k = 0; k < 9; k++) out[(e * 3 + which) * 9 + k] = c.l[k]; };
    auto put_v = [&](uint32_t base, uint32_t m, int which, uint32_t j, const frl &c) { for (int k = 0; k < 9; k++) out[base + (which * 9 + k) * m + j] = c.l[k]; };
    const uint32_t dh[5] = {1024, 256, 64, 16, 4}, dbase[3] = {TW_D1024, TW_D256, TW_D64};
    for (int pi = 0; pi < 5; pi++) {
        const uint32_t h = dh[pi], q = h / 2;
        for (uint32_t i = 0; i < q; i++) {
            const frl w[3] = {D(h, i), D(h, i + q), D(q, i)};
            for (int which = 0; which < 3; which++) {
                if (pi < 3) put_v(dbase[pi], q, which, i, w[which]);
                else put_u((pi == 3 ? E_D16 : E_D4) + i, which, w[which]);
            }
        }
    }
    put_u(E_MID, 0, U(1, 0));
    const uint32_t uh[5] = {2, 8, 32, 128, 512}, ubase[5] = {0, 0, TW_U32, TW_U128, TW_U512};
    for (int pi = 0; pi < 5; pi++) {
        const uint32_t h = uh[pi];
        for (uint32_t i = 0; i < h; i++) {
            const frl w[3] = {U(h, i), U(2 * h, i), U(2 * h, i + h)};
            for (int which = 0; which < 3; which++) {
                if (pi >= 2) put_v(ubase[pi], h, which, i, w[which]);
                else put_u((pi == 0 ? E_U2 : E_U8) + i, which, w[which]);
            }
        }
    }
}

}  // namespace das2k
}  // namespace kzg
