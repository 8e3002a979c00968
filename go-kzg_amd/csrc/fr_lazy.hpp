// fr_lazy.hpp -- F_r on 9 UNSATURATED 29-bit limbs, lazily reduced: the arithmetic of the LDS-resident transforms
// (k_fr_fft4096_r4, k_das_ext2048_r4 in k_fr.hip; replaces the butterfly loops of fft_fr.go:30-53 and das_extension.go:7-66).
//
// Why 29 bits (the general F_r product of field.hpp uses 30): a 32-bit word then has three spare bits, so sums and
// differences of a few values need no carry sweep at all ("raw" limbs, each < 6 * 2^29), and a 64-bit column of the
// product holds 64 partial products of 2^58: the nine rounds of a product with a raw operand (< 6 * 2^29 per limb)
// accumulate at most 63 of them -- no sweep inside the product either.  Values are only BOUNDED (v < B r, B tracked
// by hand at every use, always < 64 so that v < 2^261 and the top limb stays below 2^29).
//
// Radix: data keeps Kilic's Montgomery image x * 2^256 (byte-identical to the Go slices).  Only the CONSTANT operand
// of a product (a twiddle, the 1/n of the inverse) is pre-scaled to the image w * 2^261: nine reduction rounds of
// 29 bits divide by 2^261, so frl_mul(data, twiddle) is again a Kilic image -- no conversion of the data anywhere.
// r = 1 (mod 2^32), hence -r^-1 = -1 (mod 2^29): the Montgomery factor of a round is a negation, not a product, and
// limb 0 of r is 1: a round is 9 + 8 multiply-adds (153 per product against 162 + 8 for the 30-bit form).
#pragma once
#include "field.hpp"

namespace kzg {

struct frl { uint32_t l[9]; };
static constexpr uint32_t FRL_MASK = 0x1fffffffu;

KZG_HD uint32_t frl_p29(int i) {          // r in 9 limbs of 29 bits
    const uint32_t t[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    return t[i];
}
// limb i of M r, spread so that every limb but the top is >= K (2^29 - 1) (the limbs still sum to M r): a + spread - b has no
// negative limb when the limbs of b are <= K (2^29 - 1) (K = 1: b normalised; K = 2: b a raw sum of two normalised values) and
// b < (M - 1) r: see frl_sub
template <int M, int K = 1> KZG_HD uint32_t frl_spread(int i) {
    uint64_t c = 0; uint32_t v = 0;
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        uint64_t t = (uint64_t)frl_p29(k) * (uint32_t)M + c;
        uint32_t limb = (k < 8) ? (uint32_t)(t & FRL_MASK) : (uint32_t)t;
        c = (k < 8) ? (t >> 29) : 0;
        if (k == i) v = limb;
    }
    if (i == 0) return v + (uint32_t)K * (1u << 29);
    if (i < 8) return v + (uint32_t)K * ((1u << 29) - 1u);
    return v - (uint32_t)K;
}

KZG_HD frl frl_unpack(const fr &a) {      // canonical 8 x 32 -> 9 x 29, normalised
    frl o;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int w = (29 * k) >> 5, sh = (29 * k) & 31;
        uint64_t v = a.l[w];
        if (w + 1 < 8) v |= (uint64_t)a.l[w + 1] << 32;
        o.l[k] = (uint32_t)(v >> sh) & FRL_MASK;
    }
    return o;
}
KZG_HD frl frl_zero() {
    frl o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.l[k] = 0;
    return o;
}
// carry sweep: raw limbs (< 2^32) -> normalised (limbs 0..7 < 2^29; the top limb takes what is left, < 2^29 for v < 2^261)
KZG_HD void frl_sweep(frl &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t t = a.l[i] + c; a.l[i] = t & FRL_MASK; c = t >> 29; }
    a.l[8] += c;
}
KZG_HD frl frl_add(const frl &a, const frl &b) {         // raw: limbs add
    frl o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i];
    return o;
}
// a - b + M r, raw.  Needs the limbs of b <= K (2^29 - 1) and b < (M - 1) r (so that its top limb is <= that of the spread); a raw.
// Limbs grow by < (K + 1) 2^29; the bound grows by M.
template <int M, int K = 1> KZG_HD frl frl_sub(const frl &a, const frl &b) {
    frl o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + frl_spread<M, K>(i) - b.l[i];
    return o;
}
// Montgomery product on 29-bit limbs: r = A B / 2^261 mod r, NORMALISED, value < A B / 2^261 + r.
//   A: raw limbs < 6 * 2^29, any bound Ba;  B: normalised limbs (< 2^29), bound Bb;  needs Ba Bb <= 64 (A B < 2^261 r), result B < 2.
// Column budget: a round adds A[j] B[i] < 6 * 2^58 and m p[j] < 2^58 to a column, nine rounds: 63 * 2^58 < 2^64.
KZG_HD frl frl_mul(const frl &A, const frl &B) {
    uint64_t acc[10];
#pragma unroll
    for (int j = 0; j < 10; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] += (uint64_t)A.l[j] * B.l[i];
        const uint32_t m = (0u - (uint32_t)acc[0]) & FRL_MASK;   // -r^-1 = -1 (mod 2^29)
        acc[1] += (acc[0] + m) >> 29;                             // limb 0 of r is 1; the low 29 bits cancel
#pragma unroll
        for (int j = 1; j < 9; j++) acc[j] += (uint64_t)m * frl_p29(j);
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] = acc[j + 1];
        acc[9] = 0;
    }
    frl o; uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t x = acc[j] + c; o.l[j] = (uint32_t)x & FRL_MASK; c = x >> 29; }
    o.l[8] = (uint32_t)(acc[8] + c);
    return o;
}
// partial reduction: any raw value with bound <= 63 -> normalised, < 1.13 r (same quotient estimate as frl_canon, in limb form)
KZG_HD frl frl_reduce(const frl &araw) {
    frl a = araw;
    frl_sweep(a);
    const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * 0x235u) >> 32);
    frl o; uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {                          // a - q r, limb-wise with a lent 2^29 per limb (wraps correctly mod 2^32 at the top)
        c += (uint64_t)q * frl_p29(j);
        const uint32_t lend = j == 0 ? (1u << 29) : (j < 8 ? (1u << 29) - 1u : 0u - 1u);
        o.l[j] = a.l[j] + lend - ((uint32_t)c & FRL_MASK);
        c >>= 29;
    }
    frl_sweep(o);
    return o;
}
// 9 x 29 normalised, value < 2^256 -> 8 x 32
KZG_HD void frl_pack(uint32_t *t, const frl &a) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int k = (32 * w) / 29, o = (32 * w) % 29;
        uint64_t v = (uint64_t)a.l[k] >> o;
        v |= (uint64_t)a.l[k + 1] << (29 - o);
        if (k + 2 < 9 && 58 - o < 32) v |= (uint64_t)a.l[k + 2] << (58 - o);
        t[w] = (uint32_t)v;
    }
}
// normalised value < 2 r -> canonical (the output of a product)
KZG_HD fr frl_canon_lt2r(const frl &a) {
    uint32_t t[8];
    frl_pack(t, a);
    fr o; reduce_once<FrP>(o, t);
    return o;
}
// any raw value with bound B <= 63 -> canonical.  Quotient estimate from the top limb: with r8 = r >> 232 and D = r8 + 1,
// q = mulhi(x8, floor(2^32 / D)) satisfies x8 / D - 9/8 < q <= x8 / D, so 0 <= x - q r < 1.13 r: one conditional subtraction.
KZG_HD fr frl_canon(const frl &araw) {
    frl a = araw;
    frl_sweep(a);
    const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * 0x235u) >> 32);    // floor(2^32 / 0x73eda8) = 0x235
    // x as 9 saturated words (x < 2^261), y = x - q r in the same form
    uint32_t X[9];
#pragma unroll
    for (int w = 0; w < 9; w++) {
        const int k = (32 * w) / 29, o = (32 * w) % 29;
        uint64_t v = (uint64_t)a.l[k] >> o;
        if (k + 1 < 9) v |= (uint64_t)a.l[k + 1] << (29 - o);
        if (k + 2 < 9 && 58 - o < 32) v |= (uint64_t)a.l[k + 2] << (58 - o);
        X[w] = (uint32_t)v;
    }
    uint32_t t[8]; uint64_t c = 0; uint32_t br = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        c += (uint64_t)q * FrP::mod(w);
        t[w] = subb(X[w], (uint32_t)c, br);
        c >>= 32;
    }
    fr o; reduce_once<FrP>(o, t);          // word 8 of y is zero: y < 2 r < 2^256
    return o;
}
// the constant operand of frl_mul from a Kilic image: w 2^256 -> w 2^261 (canonical), unpacked
KZG_HD frl frl_const_from_kilic(const fr &w) {
    fr k32 = fr_from_u64(32);                  // Kilic image of 2^5
    return frl_unpack(mont_mul_fr30(w, k32));
}

}  // namespace kzg
