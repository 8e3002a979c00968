// g1.hpp -- BLS12-381 G1 group law for gfx950 lanes (one point per lane), y^2 = x^3 + 4.
//
// Replaces bls.AddG1 / SubG1 / MulG1 / ClearG1 / EqualG1 (bls/bls_kilic.go:33-53,106) -> Kilic G1.Add /
// Sub / MulScalar.  Jacobian (X, Y, Z) with Montgomery coordinates, inf <=> Z == 0, byte-identical to
// Kilic's PointG1.  Every add handles the exceptional cases exactly (inf + P, P + P,
// P + (-P)): in FK20 half of every FFT input is the point at infinity and twiddles include 1, so these
// are routine, not rare (SURVEY.md 7 "hard parts").
#pragma once
#include "field.hpp"

namespace kzg {

struct g1j { fp x, y, z; };   // Jacobian, 144 B (= bls.G1Point)
struct g1a { fp x, y; };      // affine, 96 B; (0, 0) encodes inf (not on the curve: 0 != 0 + 4)

KZG_HD bool is_inf(const g1j &p) { return is_zero<FpP>(p.z); }
KZG_HD bool is_inf(const g1a &p) { return is_zero<FpP>(p.x) && is_zero<FpP>(p.y); }
KZG_HD g1j g1_inf() { g1j o; o.x = zero<FpP>(); o.y = one<FpP>(); o.z = zero<FpP>(); return o; }   // Kilic Zero(): (0, 1, 0)
KZG_HD g1a g1a_inf() { g1a o; o.x = zero<FpP>(); o.y = zero<FpP>(); return o; }
KZG_HD g1j g1_neg(const g1j &p) { g1j o = p; o.y = neg<FpP>(p.y); return o; }
KZG_HD g1a g1_neg(const g1a &p) { g1a o = p; o.y = neg<FpP>(p.y); return o; }
KZG_HD g1j to_jac(const g1a &p) {
    g1j o;
    if (is_inf(p)) return g1_inf();
    o.x = p.x; o.y = p.y; o.z = one<FpP>();
    return o;
}

// dbl-2009-l (a = 0): 2M + 5S
KZG_HD g1j g1_dbl(const g1j &p) {
    if (is_inf(p)) return g1_inf();
    fp a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    fp t = add(p.x, b); t = sqr(t); t = sub(t, a); t = sub(t, c);
    fp d = add(t, t);
    fp e = add(add(a, a), a);
    fp f = sqr(e);
    g1j o;
    o.x = sub(f, add(d, d));
    o.z = mul(p.y, p.z); o.z = add(o.z, o.z);
    fp c8 = add(c, c); c8 = add(c8, c8); c8 = add(c8, c8);
    o.y = sub(mul(e, sub(d, o.x)), c8);
    return o;
}

// add-2007-bl: 11M + 5S, exceptional cases handled
KZG_HD g1j g1_add(const g1j &p, const g1j &q) {
    if (is_inf(p)) return q;
    if (is_inf(q)) return p;
    fp z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    fp u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    fp s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    if (equal<FpP>(u1, u2)) {
        if (equal<FpP>(s1, s2)) return g1_dbl(p);
        return g1_inf();
    }
    fp h = sub(u2, u1);
    fp i = add(h, h); i = sqr(i);
    fp j = mul(h, i);
    fp r = sub(s2, s1); r = add(r, r);
    fp v = mul(u1, i);
    g1j o;
    o.x = sub(sub(sub(sqr(r), j), v), v);
    fp t = mul(s1, j); t = add(t, t);
    o.y = sub(mul(r, sub(v, o.x)), t);
    o.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return o;
}

// madd-2007-bl (q affine, Z2 = 1): 7M + 4S, exceptional cases handled
KZG_HD g1j g1_madd(const g1j &p, const g1a &q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return to_jac(q);
    fp z1z1 = sqr(p.z);
    fp u2 = mul(q.x, z1z1);
    fp s2 = mul(mul(q.y, p.z), z1z1);
    if (equal<FpP>(p.x, u2)) {
        if (equal<FpP>(p.y, s2)) return g1_dbl(p);
        return g1_inf();
    }
    fp h = sub(u2, p.x);
    fp hh = sqr(h);
    fp i = add(hh, hh); i = add(i, i);
    fp j = mul(h, i);
    fp r = sub(s2, p.y); r = add(r, r);
    fp v = mul(p.x, i);
    g1j o;
    o.x = sub(sub(sub(sqr(r), j), v), v);
    fp t = mul(p.y, j); t = add(t, t);
    o.y = sub(mul(r, sub(v, o.x)), t);
    o.z = sub(sub(sqr(add(p.z, h)), z1z1), hh);
    return o;
}

// ---------------------------------------------------------------------------------------------
// XYZZ accumulator (x = X / ZZ, y = Y / ZZZ, ZZ^3 == ZZZ^2; inf <=> ZZ == 0) for loops that only ever add AFFINE points
// (the fixed-base table walks): madd-2008-s costs 8M + 2S = 10 products against 7M + 4S = 11 for Jacobian + affine,
// and this multiplier has no cheaper squaring.  Exceptional cases handled as everywhere else.
// ---------------------------------------------------------------------------------------------
struct g1x { fp x, y, zz, zzz; };
KZG_HD g1x g1x_inf() { g1x o; o.x = zero<FpP>(); o.y = one<FpP>(); o.zz = zero<FpP>(); o.zzz = zero<FpP>(); return o; }
KZG_HD bool is_inf(const g1x &p) { return is_zero<FpP>(p.zz); }
KZG_HD g1x g1x_from_jac(const g1j &p) {
    if (is_inf(p)) return g1x_inf();
    g1x o; o.x = p.x; o.y = p.y; o.zz = sqr(p.z); o.zzz = mul(o.zz, p.z);
    return o;
}
KZG_HD g1j g1x_to_jac(const g1x &p) {   // (X ZZ, Y ZZZ, ZZ) is a Jacobian image of the same point
    if (is_inf(p)) return g1_inf();
    g1j o; o.x = mul(p.x, p.zz); o.y = mul(p.y, p.zzz); o.z = p.zz;
    return o;
}
KZG_HD g1x g1x_madd(const g1x &p, const g1a &q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) { g1x o; o.x = q.x; o.y = q.y; o.zz = one<FpP>(); o.zzz = one<FpP>(); return o; }
    fp u2 = mul(q.x, p.zz), s2 = mul(q.y, p.zzz);
    if (equal<FpP>(u2, p.x)) {
        if (equal<FpP>(s2, p.y)) return g1x_from_jac(g1_dbl(to_jac(q)));   // P == Q: double the affine operand
        return g1x_inf();                                                    // P == -Q
    }
    fp pp_ = sub(u2, p.x), r = sub(s2, p.y);
    fp pp = sqr(pp_), ppp = mul(pp_, pp), q_ = mul(p.x, pp);
    g1x o;
    o.x = sub(sub(sub(sqr(r), ppp), q_), q_);
    o.y = sub(mul(r, sub(q_, o.x)), mul(p.y, ppp));
    o.zz = mul(p.zz, pp);
    o.zzz = mul(p.zzz, ppp);
    return o;
}

// Fast path of g1x_madd on unpacked, lazily reduced coordinates (field.hpp: fq).  Bounds (value < B p) are a loop invariant:
//   in : X <= 11, Y <= 5, ZZ <= 2, ZZZ <= 2   (the first point enters with all bounds 1)
//   u2, s2 = 2;  P = u2 - X (M = 12) -> 14;  R = s2 - Y (M = 6) -> 8;  PP, PPP, Q = 2  (products 196, 28, 22 <= 600)
//   X3 = R^2 - PPP - 2 Q : 2 + 3 + 3 + 3 = 11;  Q - X3 (M = 12) -> 14;  R (Q - X3): 8 * 14 = 112 <= 600
//   Y3 = R (Q - X3) + (6 p - Y) PPP in ONE reduction (dot2q: 8 * 14 + 11 * 2 = 134 <= 600) : 2;  ZZ3, ZZZ3 = 2
//                                                                        -> the invariant is reproduced.
// Returns false (and leaves the accumulator untouched) when P == +-Q: the caller takes the generic path for those.
struct g1xq { fq x, y, zz, zzz; };
KZG_HD bool g1x_madd_fast(g1xq &p, const fq &x2, const fq &y2) {
    fq u2 = mulq(x2, p.zz), s2 = mulq(y2, p.zzz);
    fq pp_ = subq<12>(u2, p.x), r = subq<6>(s2, p.y);
    fq pp = sqrq(pp_);
    if (KZG_UNLIKELY(is_zero_mod_p_q(pp))) return false;
    fq ppp = mulq(pp_, pp), q_ = mulq(p.x, pp);
    fq x3 = subq<3>(subq<3>(subq<3>(sqrq(r), ppp), q_), q_);
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    fq y3 = dot2q_inl(r, subq<12>(q_, x3), subq<6>(zero_q, p.y), ppp);
    p.zz = mulq(p.zz, pp);
    p.zzz = mulq(p.zzz, ppp);
    p.x = x3; p.y = y3;
    return true;
}
KZG_HD g1xq g1xq_from_affine(const g1a &q) {
    g1xq o; o.x = unpackq(q.x); o.y = unpackq(q.y); o.zz = unpackq(one<FpP>()); o.zzz = o.zz;
    return o;
}
KZG_HD g1x g1xq_pack(const g1xq &p) { g1x o; o.x = packq(p.x); o.y = packq(p.y); o.zz = packq(p.zz); o.zzz = packq(p.zzz); return o; }
KZG_HD g1xq g1xq_unpack(const g1x &p) { g1xq o; o.x = unpackq(p.x); o.y = unpackq(p.y); o.zz = unpackq(p.zz); o.zzz = unpackq(p.zzz); return o; }
// accumulator with an explicit infinity flag; add() is what the table-walk kernels call per table entry
struct g1x_acc {
    g1xq v; bool inf;
    KZG_HD void init() { inf = true; }
    KZG_HD void add(const g1a &q) {
        // branch weights: the compiler lays the (never taken) generic path out of the straight line of the walk loop: +2.7 % measured
        if (KZG_UNLIKELY(is_inf(q))) return;
        if (KZG_UNLIKELY(inf)) { v = g1xq_from_affine(q); inf = false; return; }
#ifdef KZG_AB_FAKE_AFFINE   // TIMING-ONLY A/B (wrong results): the multiply-adds of a BATCH-AFFINE addition whose shared inversion, prefix-product storage and
        // result storage came for free -- the ceiling of that scheme for the walk (profiles/r05_batch_affine.md): prefix product (1M), two products to unwind
        // the inverse of the denominator (2M), lambda (1M), x3 (1S), y3 (1M) = 5M + 1S against the 8M + 2S (one reduction saved) of the XYZZ mixed addition
        {
            const fq x2 = unpackq(q.x), y2 = unpackq(q.y);
            const fq d = subq<12>(x2, v.x);
            const fq pre = mulq(v.zz, d);
            const fq invd = mulq(v.zzz, pre);
            v.zzz = mulq(v.zzz, d);
            const fq lam = mulq(subq<6>(y2, v.y), invd);
            const fq x3 = subq<3>(subq<12>(sqrq(lam), v.x), x2);
            v.y = subq<6>(mulq(lam, subq<12>(v.x, x3)), v.y);
            v.zz = pre; v.x = x3;
            return;
        }
#endif
#ifdef KZG_AB_FAKE_UNPACK   // TIMING-ONLY A/B (wrong results): what the walk would cost if table entries arrived as 13 limbs (profiles/r04_walk_ab.md)
        fq fx_, fy_;
#pragma unroll
        for (int i_ = 0; i_ < 12; i_++) { fx_.l[i_] = q.x.l[i_]; fy_.l[i_] = q.y.l[i_]; }
        fx_.l[12] = q.x.l[0] >> 8; fy_.l[12] = q.y.l[0] >> 8;
        if (KZG_LIKELY(g1x_madd_fast(v, fx_, fy_))) return;
#else
        if (KZG_LIKELY(g1x_madd_fast(v, unpackq(q.x), unpackq(q.y)))) return;
#endif
        g1x s = g1x_madd(g1xq_pack(v), q);          // P == Q or P == -Q: generic, complete formulas
        if (is_inf(s)) inf = true; else v = g1xq_unpack(s);
    }
    KZG_HD g1j to_jac() const { return inf ? g1_inf() : g1x_to_jac(g1xq_pack(v)); }
};

// XYZZ + XYZZ on the same lazy limbs (add-2008-s, 12M + 2S, the Y3 pair under one reduction: 11.5M + 2S): the reduction trees that
// sum the per-lane accumulators of a table walk.  Both operands obey the accumulator invariant (X, Y, ZZ, ZZZ) <= (11, 5, 2, 2):
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1 : 2 (products <= 22);  P = U2 - U1 (M = 3) : 5;  R = S2 - S1 (M = 3) : 5
//   PP = P^2, PPP = P PP, Q = U1 PP : 2;  X3 = R^2 - PPP - 2 Q : 2 + 3 + 3 + 3 = 11;  Q - X3 (M = 12) : 14
//   Y3 = R (Q - X3) + (3 p - S1) PPP in one reduction (5 * 14 + 3 * 2 = 76 <= 600) : 2;  ZZ3 = (ZZ1 ZZ2) PP, ZZZ3 = (ZZZ1 ZZZ2) PPP : 2
// so the sum obeys the invariant again.  Returns false (a untouched) when P == 0, i.e. the operands are equal or opposite.
KZG_HD bool g1xq_add_fast(g1xq &a, const g1xq &b) {
    fq u1 = mulq(a.x, b.zz), u2 = mulq(b.x, a.zz);
    fq s1 = mulq(a.y, b.zzz), s2 = mulq(b.y, a.zzz);
    fq pp_ = subq<3>(u2, u1), r = subq<3>(s2, s1);
    fq pp = sqrq(pp_);
    if (KZG_UNLIKELY(is_zero_mod_p_q(pp))) return false;
    fq ppp = mulq(pp_, pp), q_ = mulq(u1, pp);
    fq x3 = subq<3>(subq<3>(subq<3>(sqrq(r), ppp), q_), q_);
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    fq y3 = dot2q_inl(r, subq<12>(q_, x3), subq<3>(zero_q, s1), ppp);
    a.zz = mulq(mulq(a.zz, b.zz), pp);
    a.zzz = mulq(mulq(a.zzz, b.zzz), ppp);
    a.x = x3; a.y = y3;
    return true;
}
// accumulator + accumulator with infinity flags; equal / opposite operands take the generic complete formulas
KZG_HD g1j g1_add(const g1j &p, const g1j &q);
KZG_HD void g1x_acc_merge(g1x_acc &a, const g1xq &bv, bool binf) {
    if (binf) return;
    if (a.inf) { a.v = bv; a.inf = false; return; }
    if (KZG_LIKELY(g1xq_add_fast(a.v, bv))) return;
    g1j s = g1_add(g1x_to_jac(g1xq_pack(a.v)), g1x_to_jac(g1xq_pack(bv)));
    if (is_inf(s)) a.inf = true; else a.v = g1xq_unpack(g1x_from_jac(s));
}

KZG_HD g1j g1_sub(const g1j &p, const g1j &q) { return g1_add(p, g1_neg(q)); }

// Projective equality (bls.EqualG1)
KZG_HD bool g1_equal(const g1j &p, const g1j &q) {
    bool pi = is_inf(p), qi = is_inf(q);
    if (pi || qi) return pi && qi;
    fp z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    if (!equal<FpP>(mul(p.x, z2z2), mul(q.x, z1z1))) return false;
    return equal<FpP>(mul(mul(p.y, q.z), z2z2), mul(mul(q.y, p.z), z1z1));
}

// Normalise one point (one F_p inversion).  Output: Z = R (Montgomery one) or Kilic's inf image (0, 1, 0).
KZG_HD g1j g1_normalize(const g1j &p) {
    if (is_inf(p)) return g1_inf();
    fp zi = inv<FpP>(p.z), zi2 = sqr(zi);
    g1j o; o.x = mul(p.x, zi2); o.y = mul(p.y, mul(zi2, zi)); o.z = one<FpP>();
    return o;
}

// 4-bit digit of a standard-form scalar (k.l = 8 x u32), window w in 0..63
KZG_HD uint32_t nibble(const fr &k, int w) { return (k.l[w >> 3] >> ((w & 7) * 4)) & 15u; }

// k * P, k in STANDARD form (the caller does Kilic's FromRed, bls/bls_kilic.go:42-43).
// Fixed 4-bit windows, MSB first: every lane of a wave executes the same dbl/add schedule.
// `tbl` is caller-provided storage for 15 multiples (per-lane: private scratch or LDS).
KZG_HD g1j g1_mul_windowed(const g1j &p, const fr &k, g1j *tbl) {
    tbl[0] = p;
    for (int i = 1; i < 15; i++) tbl[i] = (i & 1) ? g1_dbl(tbl[i >> 1]) : g1_add(tbl[i - 1], p);   // tbl[i] = (i+1) P
    g1j acc = g1_inf();
    for (int w = 63; w >= 0; w--) {
        acc = g1_dbl(g1_dbl(g1_dbl(g1_dbl(acc))));
        uint32_t d = nibble(k, w);
        if (d) acc = g1_add(acc, tbl[d - 1]);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// GLV: phi(x, y) = (beta x, y) acts on G1 as multiplication by lambda = z^2 - 1 (z the BLS12-381 parameter),
// and lambda^2 + lambda + 1 = r, so k = k2 lambda + k1 with k1 = k mod lambda, k2 = k div lambda both < 2^128 and
// non-negative.  k P = k1 P + k2 phi(P) with SHARED doublings: 128 instead of 256.
// `kk` holds k1 in limbs 0..3 and k2 in limbs 4..7 (standard form); phi of a table entry costs one F_p product by beta
// at lookup time.
// ---------------------------------------------------------------------------------------------
KZG_HD fp glv_beta() {   // cube root of unity with phi(G) == lambda G (checked in tests/test_host_arith.py), radix-2^390 Montgomery
    const uint32_t t[12] = {0x9c907181u, 0xef2f7921u, 0xb26574c3u, 0x1bcc91d7u, 0x191c3ebcu, 0x856e7b9au,
                            0x67fd6ffau, 0xbd16b0d2u, 0xeb0c0550u, 0x18c86532u, 0x6567dd7du, 0x09c6d485u};
    fp b;
#pragma unroll
    for (int i = 0; i < 12; i++) b.l[i] = t[i];
    return b;
}
// 5-bit signed digit j of the 128-bit value in limbs l[base .. base + 3] (plus the incoming carry): digits in [-16, 16]
KZG_HD int glv_digit5(const fr &kk, int base, int j, uint32_t &carry) {
    int bit = 5 * j, w = bit >> 5, sh = bit & 31;
    uint64_t v = 0;
    if (w < 4) v = kk.l[base + w];
    if (w + 1 < 4) v |= (uint64_t)kk.l[base + w + 1] << 32;
    uint32_t raw = ((uint32_t)(v >> sh) & 31u) + carry;
    if (raw > 16u) { carry = 1; return (int)raw - 32; }
    carry = 0; return (int)raw;
}
// Signed 5-bit windows on both halves: 26 windows (130 doublings) and at most 52 additions from ONE 16-entry table
// (tbl[i] = (i + 1) P); a negative digit negates Y, the k2 half multiplies X by beta.  Digits are produced LSB first
// (carry propagation) into two packed arrays, then consumed MSB first.
KZG_HD g1j g1_mul_glv(const g1j &p, const fr &kk, g1j *tbl) {
    tbl[0] = p;
    for (int i = 1; i < 16; i++) tbl[i] = (i & 1) ? g1_dbl(tbl[i >> 1]) : g1_add(tbl[i - 1], p);   // tbl[i] = (i+1) P
    int8_t d1[27], d2[27];
    uint32_t c1 = 0, c2 = 0;
    for (int j = 0; j < 27; j++) { d1[j] = (int8_t)glv_digit5(kk, 0, j, c1); d2[j] = (int8_t)glv_digit5(kk, 4, j, c2); }
    const fp beta = glv_beta();
    g1j acc = g1_inf();
    for (int j = 26; j >= 0; j--) {
        acc = g1_dbl(g1_dbl(g1_dbl(g1_dbl(g1_dbl(acc)))));
        int a = d1[j], b = d2[j];
        if (a) { g1j q = tbl[(a < 0 ? -a : a) - 1]; if (a < 0) q.y = neg<FpP>(q.y); acc = g1_add(acc, q); }
        if (b) { g1j q = tbl[(b < 0 ? -b : b) - 1]; if (b < 0) q.y = neg<FpP>(q.y); q.x = mul(q.x, beta); acc = g1_add(acc, q); }
    }
    return acc;
}
// host-side decomposition of a standard-form scalar: out = (k mod lambda, k div lambda)
inline fr glv_decompose(const fr &k) {
    typedef unsigned __int128 u128;
    const u128 lam = ((u128)0xac45a4010001a402ull << 64) | 0x00000000ffffffffull;
    u128 rem = 0, q = 0;
    for (int b = 255; b >= 0; b--) {
        bool top = (rem >> 127) & 1;
        rem = (rem << 1) | ((k.l[b >> 5] >> (b & 31)) & 1u);
        if (top || rem >= lam) { rem -= lam; q |= (b < 128) ? ((u128)1 << b) : 0; }
    }
    fr o;
    for (int i = 0; i < 4; i++) { o.l[i] = (uint32_t)(rem >> (32 * i)); o.l[4 + i] = (uint32_t)(q >> (32 * i)); }
    return o;
}

// Balanced GLV split for VARIABLE scalars, on the device (the bucket MSM): k P = s1 |k1| P + s2 |k2| phi(P) with both magnitudes
// below 2^126.5, so 16 signed 8-bit windows cover each half with no carry out of the top window.
//   a = min(k, r - k) (sign sg), q = round(a / lambda) = floor((a + hl) / lambda), hl = floor(lambda / 2),
//   k1 = a - q lambda in [-hl, hl], k2 = q <= (r / 2 + hl) / lambda < 2^126.5;   neg1 = sg ^ (k1 < 0), neg2 = sg.
// The quotient is a Barrett estimate with mu = floor(2^256 / lambda) (129 bits): floor(floor(a' / 2^127) mu / 2^129) is q - 2 .. q,
// fixed by at most two conditional subtractions.  Checked against Python big integers in tests/test_host_arith.py.
struct glv_halves { uint32_t k1[4], k2[4]; uint32_t neg1, neg2; };
KZG_HD glv_halves glv_split_signed(const fr &k) {
    const uint32_t LAM[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};
    const uint32_t HL[4] = {0x7fffffffu, 0x00000000u, 0x8000d201u, 0x5622d200u};
    const uint32_t MU[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};
    const uint32_t HALF_R[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};
    // sg = k > (r - 1) / 2;  a = sg ? r - k : k
    uint32_t gt = 0, decided = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        uint32_t g = k.l[i] > HALF_R[i], l = k.l[i] < HALF_R[i];
        gt |= g & ~decided; decided |= g | l;
    }
    const uint32_t sg = gt & 1u;
    uint32_t a[9], br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t d = subb(FrP::mod(i), k.l[i], br); a[i] = sg ? d : k.l[i]; }
    // a += hl  (a < 2^254 + 2^127)
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = addc(a[i], i < 4 ? HL[i] : 0u, c);
    a[8] = 0;
    // t = a >> 127 (4 limbs);  prod = t * mu (9 limbs);  q = prod >> 129
    uint32_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = (a[3 + i] >> 31) | (a[4 + i] << 1);
    uint32_t prod[9];
#pragma unroll
    for (int i = 0; i < 9; i++) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t cy = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) { uint64_t x = (uint64_t)t[i] * MU[j] + prod[i + j] + cy; prod[i + j] = (uint32_t)x; cy = x >> 32; }
        prod[i + 5] = (uint32_t)cy;
    }
    uint32_t q[5];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = (prod[4 + i] >> 1) | (prod[5 + i] << 31);
    q[4] = 0;
    // rem = a - q lambda (5 limbs are enough: rem < 3 lambda)
    uint32_t ql[5];
#pragma unroll
    for (int i = 0; i < 5; i++) ql[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t cy = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + j < 5) { uint64_t x = (uint64_t)q[i] * LAM[j] + ql[i + j] + cy; ql[i + j] = (uint32_t)x; cy = x >> 32; }
        }
        if (i + 4 < 5) ql[i + 4] += (uint32_t)cy;
    }
    uint32_t rem[5]; br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) rem[i] = subb(a[i], ql[i], br);
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {   // rem >= lambda: rem -= lambda, q += 1
        uint32_t d[5]; uint32_t b2 = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) d[i] = subb(rem[i], i < 4 ? LAM[i] : 0u, b2);
        const uint32_t ge = b2 ^ 1u;
        uint32_t cc = ge;
#pragma unroll
        for (int i = 0; i < 5; i++) rem[i] = ge ? d[i] : rem[i];
#pragma unroll
        for (int i = 0; i < 4; i++) { uint32_t x = q[i] + cc; cc = x < cc ? 1u : 0u; q[i] = x; }
    }
    // k1 = rem - hl (signed)
    glv_halves o;
    uint32_t d1[4], d2[4], b1 = 0, b2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { d1[i] = subb(rem[i], HL[i], b1); d2[i] = subb(HL[i], rem[i], b2); }
#pragma unroll
    for (int i = 0; i < 4; i++) { o.k1[i] = b1 ? d2[i] : d1[i]; o.k2[i] = q[i]; }
    o.neg1 = sg ^ b1; o.neg2 = sg;
    return o;
}

// ---------------------------------------------------------------------------------------------
// Jacobian arithmetic on unpacked, lazily reduced coordinates (fq) for the GLV scalar multiplication of the G1 FFT.
// Bound invariant (value < B p): every point that lives in the loop has (X, Y, Z) <= (19, 20, 4).
//   dbl (dbl-2009-l with D = 4 X Y^2 instead of 2((X + Y^2)^2 - X^2 - Y^4): same 3M + 4S, no subtractions in D):
//     A = X^2, B = Y^2, C = B^2, S = X B : 2 each (products <= 20^2 = 400 <= 600)
//     D = 4 S : 8;  E = 3 A : 6;  F = E^2 : 2;  X3 = F - 2 D (M = 17) : 19
//     Y3 = E (D - X3) - 8 C : D - X3 (M = 20) : 28, 6 * 28 = 168, then 2 - 16 (M = 17) : 19;  Z3 = 2 Y Z : 4 (20 * 4 = 80)
//   add (add-2007-bl, Z3 = 2 Z1 Z2 H), P1 <= (19, 20, 4), P2 <= (19, 20, 4):
//     Z1Z1, Z2Z2, U1, U2, S1, S2 : 2 (largest product 20 * 4 = 80);  H = U2 - U1 (M = 3) : 5;  I = (2 H)^2 : 2 (100)
//     J = H I : 2;  r = 2 (S2 - S1) : 10;  V = U1 I : 2;  X3 = r^2 - J - 2 V : 2 + 3 + 3 + 3 = 11
//     Y3 = r (V - X3) - 2 S1 J : (M = 12) 14, 10 * 14 = 140, 2 - 4 (M = 5) : 7;  Z3 = 2 (Z1 Z2) H : 4
//   negation: Y -> 20 p - Y (M = 20 >= 19 + 1) : 20;  phi: X -> beta X : 2.
// add returns false when H == 0 (P1 == +-P2); the caller then uses the generic complete formulas.
// Inputs are assumed to lie in G1 (as the reference assumes): Y == 0 cannot occur.
// ---------------------------------------------------------------------------------------------
struct g1jq { fq x, y, z; };
KZG_HD g1jq g1jq_unpack(const g1j &p) { g1jq o; o.x = unpackq(p.x); o.y = unpackq(p.y); o.z = unpackq(p.z); return o; }
KZG_HD g1j g1jq_pack(const g1jq &p) { g1j o; o.x = packq(p.x); o.y = packq(p.y); o.z = packq(p.z); return o; }
KZG_HD g1jq g1jq_dbl(const g1jq &p) {
    fq a = sqrq(p.x), b = sqrq(p.y), c = sqrq(b), s_ = mulq(p.x, b);
    fq d = addq(s_, s_); d = addq(d, d);                   // 4 X Y^2 : 8
    fq e = addq(addq(a, a), a);                            // 6
    fq f = sqrq(e);
    g1jq o;
    o.x = subq<17>(f, addq(d, d));                         // 19
    fq c8 = addq(c, c); c8 = addq(c8, c8); c8 = addq(c8, c8);   // 16
    o.y = subq<17>(mulq(e, subq<20>(d, o.x)), c8);         // 19
    fq yz = mulq(p.y, p.z);
    o.z = addq(yz, yz);                                    // 4
    return o;
}
// the same doubling with the seven products inlined (3.3 k instructions): used by the tight doubling loop of g1_mul_glv_wnaf
KZG_HD g1jq g1jq_dbl_inl(const g1jq &p) {
    fq a = sqrq_inl(p.x), b = sqrq_inl(p.y), s_ = mulq_inl(p.x, b);
    fq d = addq(s_, s_); d = addq(d, d);                   // 8
    fq e = addq(addq(a, a), a);                            // 6
    fq f = sqrq_inl(e);
    g1jq o;
    o.x = subq<17>(f, addq(d, d));                         // 19
    // Y3 = E (D - X3) - 8 B^2 as ONE reduction: E (D - X3) + (17 p - 8 B) B  (6 * 28 + 33 * 2 = 234 <= 600) : 2
    fq b8 = addq(b, b); b8 = addq(b8, b8); b8 = addq(b8, b8);   // 16
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    o.y = dot2q_inl(e, subq<20>(d, o.x), subq<17>(zero_q, b8), b);
    fq yz = mulq_inl(p.y, p.z);
    o.z = addq(yz, yz);                                    // 4
    return o;
}
KZG_HD bool g1jq_add(g1jq &o, const g1jq &p, const g1jq &q) {
    fq z1z1 = sqrq(p.z), z2z2 = sqrq(q.z);
    fq u1 = mulq(p.x, z2z2), u2 = mulq(q.x, z1z1);
    fq s1 = mulq(mulq(p.y, q.z), z2z2), s2 = mulq(mulq(q.y, p.z), z1z1);
    fq h = subq<3>(u2, u1);
    fq h2 = addq(h, h);
    fq i = sqrq(h2);
    if (KZG_UNLIKELY(is_zero_mod_p_q(i))) return false;
    fq j = mulq(h, i);
    fq r = subq<3>(s2, s1); r = addq(r, r);
    fq v = mulq(u1, i);
    fq x3 = subq<3>(subq<3>(subq<3>(sqrq(r), j), v), v);
    fq sj = mulq(s1, j);
    fq y3 = subq<5>(mulq(r, subq<12>(v, x3)), addq(sj, sj));
    fq zz = mulq(mulq(p.z, q.z), h);
    o.x = x3; o.y = y3; o.z = addq(zz, zz);
    return true;
}
// accumulator of the scalar multiplication: unpacked point + explicit infinity flag
struct g1jq_acc {
    g1jq v; bool inf;
    KZG_HD void dbl() { if (!inf) v = g1jq_dbl(v); }
    KZG_HD void add(const g1jq &q) {
        if (inf) { v = q; inf = false; return; }
        g1jq o;
        if (g1jq_add(o, v, q)) { v = o; return; }
        g1j s = g1_add(g1jq_pack(v), g1jq_pack(q));       // P == +-Q: generic complete formulas
        if (is_inf(s)) inf = true; else v = g1jq_unpack(s);
    }
};
// k P via GLV with signed 5-bit windows, all group arithmetic on unpacked lazy coordinates; p must not be inf.
// `tbl` (16 entries, (i + 1) P) lives in the lane's private scratch.
KZG_HD g1j g1_mul_glv_fast(const g1j &p, const fr &kk, g1jq *tbl) {
    tbl[0] = g1jq_unpack(p);
    for (int i = 1; i < 16; i++) {
        if (i & 1) tbl[i] = g1jq_dbl(tbl[i >> 1]);
        else {
            g1jq o;
            if (!g1jq_add(o, tbl[i - 1], tbl[0])) o = g1jq_unpack(g1_add(g1jq_pack(tbl[i - 1]), p));   // cannot happen for points of G1
            tbl[i] = o;
        }
    }
    int8_t d1[27], d2[27];
    uint32_t c1 = 0, c2 = 0;
    for (int j = 0; j < 27; j++) { d1[j] = (int8_t)glv_digit5(kk, 0, j, c1); d2[j] = (int8_t)glv_digit5(kk, 4, j, c2); }
    const fq beta = unpackq(glv_beta());
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    g1jq_acc acc; acc.inf = true;
    for (int j = 26; j >= 0; j--) {
        for (int t = 0; t < 5; t++) acc.dbl();
        int a = d1[j], b = d2[j];
        if (a) { g1jq q = tbl[(a < 0 ? -a : a) - 1]; if (a < 0) q.y = subq<20>(zero_q, q.y); acc.add(q); }
        if (b) { g1jq q = tbl[(b < 0 ? -b : b) - 1]; if (b < 0) q.y = subq<20>(zero_q, q.y); q.x = mulq(q.x, beta); acc.add(q); }
    }
    return acc.inf ? g1_inf() : g1jq_pack(acc.v);
}

// the same regular schedule for a VARIABLE scalar split on the device (glv_split_signed): signs are applied to the digits, so
// s1 |k1| P + s2 |k2| phi(P) costs exactly what g1_mul_glv_fast costs.  Result unpacked (bounds (19, 20, 4)); returns false for inf.
KZG_HD bool g1_mul_glv_signed_q(const g1j &p, const glv_halves &h, g1jq *tbl, g1jq &out) {
    tbl[0] = g1jq_unpack(p);
    for (int i = 1; i < 16; i++) {
        if (i & 1) tbl[i] = g1jq_dbl(tbl[i >> 1]);
        else {
            g1jq o;
            if (!g1jq_add(o, tbl[i - 1], tbl[0])) o = g1jq_unpack(g1_add(g1jq_pack(tbl[i - 1]), p));   // cannot happen for points of G1
            tbl[i] = o;
        }
    }
    fr kk;
#pragma unroll
    for (int i = 0; i < 4; i++) { kk.l[i] = h.k1[i]; kk.l[4 + i] = h.k2[i]; }
    int8_t d1[27], d2[27];
    uint32_t c1 = 0, c2 = 0;
    for (int j = 0; j < 27; j++) { d1[j] = (int8_t)glv_digit5(kk, 0, j, c1); d2[j] = (int8_t)glv_digit5(kk, 4, j, c2); }
    const fq beta = unpackq(glv_beta());
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    g1jq_acc acc; acc.inf = true;
    for (int j = 26; j >= 0; j--) {
        for (int t = 0; t < 5; t++) acc.dbl();
        int a = h.neg1 ? -d1[j] : d1[j], b = h.neg2 ? -d2[j] : d2[j];
        if (a) { g1jq q = tbl[(a < 0 ? -a : a) - 1]; if (a < 0) q.y = subq<20>(zero_q, q.y); acc.add(q); }
        if (b) { g1jq q = tbl[(b < 0 ? -b : b) - 1]; if (b < 0) q.y = subq<20>(zero_q, q.y); q.x = mulq(q.x, beta); acc.add(q); }
    }
    out = acc.v;
    return !acc.inf;
}

// ---- width-5 NAF variant: what the G1 FFT stages run ------------------------------------------------------------------------
// The twiddle of a butterfly is wave-uniform (k_g1_fft_stage orders lanes twiddle-major), so an irregular digit schedule costs no
// divergence: each 128-bit GLV half is recoded into a width-5 NAF (odd digits +-1..+-15, on average one non-zero in six), the
// table holds the 8 odd multiples with Z^2 and Z^3 cached, and the loop runs ~128 doublings + ~43 additions instead of
// 135 + ~54 with a 16-entry table.
KZG_HD int glv_wnaf5(const fr &kk, int base, int8_t *d, int stride) {   // LSB first into d[i * stride], returns the digit count (<= 130)
    uint32_t w0 = kk.l[base], w1 = kk.l[base + 1], w2 = kk.l[base + 2], w3 = kk.l[base + 3], w4 = 0;
    int len = 0;
#pragma nounroll
    for (int i = 0; i < 130; i++) {
        int dig = 0;
        if (w0 & 1u) {
            dig = (int)(w0 & 31u);
            w0 &= ~31u;
            if (dig >= 16) {                                // k - (dig - 32) = (k with the low 5 bits cleared) + 32
                dig -= 32;
                uint32_t t = w0 + 32u, c = t < 32u ? 1u : 0u; w0 = t;
                t = w1 + c; c = t < c ? 1u : 0u; w1 = t;
                t = w2 + c; c = t < c ? 1u : 0u; w2 = t;
                t = w3 + c; c = t < c ? 1u : 0u; w3 = t;
                w4 += c;
            }
            len = i + 1;
        }
        d[i * stride] = (int8_t)dig;
        w0 = (w0 >> 1) | (w1 << 31); w1 = (w1 >> 1) | (w2 << 31); w2 = (w2 >> 1) | (w3 << 31); w3 = (w3 >> 1) | (w4 << 31); w4 >>= 1;
    }
    return len;
}
struct g1jq_t { fq x, y, z, zz, zzz; };                     // table entry: Jacobian point with Z^2, Z^3 cached (bounds 19, 20, 4, 2, 2)
KZG_HD void g1jq_t_make(g1jq_t *o, const g1jq &p) { o->x = p.x; o->y = p.y; o->z = p.z; o->zz = sqrq(p.z); o->zzz = mulq(o->zz, p.z); }
// acc += (+-) (phi?) *t : add-2007-bl with the table entry's Z^2, Z^3 cached (11M + 3S, + 1M for phi).  The entry is read field by
// field where it is used and the sign is applied to S2, so the operand never sits in registers as a whole; beta is materialised
// inside the phi branch (literals) instead of living in 13 registers across the whole loop.  Bounds as in g1jq_add, with
// S2 <= 3 when negated: r <= 12, r (V - X3) <= 12 * 14.  Returns false when H == 0 (P == +-Q).
template <bool INL = false> KZG_HD bool g1jq_add_entry(g1jq &acc, const g1jq_t *t, bool ng, bool phi) {
    auto MQ = [](const fq &a_, const fq &b_) { return INL ? mulq_inl(a_, b_) : mulq(a_, b_); };
    auto SQ = [](const fq &a_) { return INL ? sqrq_inl(a_) : sqrq(a_); };
    fq z1z1 = SQ(acc.z);
    fq u2 = phi ? MQ(MQ(t->x, unpackq(glv_beta())), z1z1) : MQ(t->x, z1z1);
    fq u1 = MQ(acc.x, t->zz);
    fq h = subq<3>(u2, u1);
    fq s2 = MQ(MQ(t->y, acc.z), z1z1);
    if (ng) { fq zero_q;
#pragma unroll
        for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
        s2 = subq<3>(zero_q, s2); }
    fq s1 = MQ(acc.y, t->zzz);
    fq r = subq<3>(s2, s1); r = addq(r, r);
    fq h2 = addq(h, h);
    fq i = SQ(h2);
    if (KZG_UNLIKELY(is_zero_mod_p_q(i))) return false;
    fq zz = MQ(MQ(acc.z, t->z), h);
    fq j = MQ(h, i);
    fq v = MQ(u1, i);
    fq x3 = subq<3>(subq<3>(subq<3>(SQ(r), j), v), v);
    if (INL) {                                             // Y3 = r (V - X3) + (5 p - 2 S1) J in ONE reduction (12 * 14 + 9 * 2 <= 600) : 2
        fq zq;
#pragma unroll
        for (int ii = 0; ii < 13; ii++) zq.l[ii] = 0;
        acc.y = dot2q_inl(r, subq<12>(v, x3), subq<5>(zq, addq(s1, s1)), j);
    } else {
        fq sj = MQ(s1, j);
        acc.y = subq<5>(MQ(r, subq<12>(v, x3)), addq(sj, sj));
    }
    acc.x = x3; acc.z = addq(zz, zz);
    return true;
}
// the entry as a plain point, sign and phi applied (first addition into an empty accumulator, and the P == +-Q fallback)
KZG_HD g1jq g1jq_entry_point(const g1jq_t *t, bool ng, bool phi) {
    g1jq q; q.x = phi ? mulq(t->x, unpackq(glv_beta())) : t->x; q.z = t->z;
    if (ng) { fq zero_q;
#pragma unroll
        for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
        q.y = subq<20>(zero_q, t->y); } else q.y = t->y;
    return q;
}
// P == +-Q (never for the points and digit schedules of the FFT, kept for completeness): generic complete formulas, out of line
// so that the cold path costs the hot loop neither registers nor scratch.  Returns true when the sum is the point at infinity.
KZG_HD_NOINLINE static bool g1jq_add_slow(g1jq *acc, const g1jq *q) {
    g1j sgen = g1_add(g1jq_pack(*acc), g1jq_pack(*q));
    if (is_inf(sgen)) return true;
    *acc = g1jq_unpack(sgen);
    return false;
}
// callers hand over COPIES so that their accumulator never has its address taken (it must stay in registers)
KZG_HD bool g1jq_add_slow_copy(g1jq &acc, const g1jq_t *t, bool ng, bool phi);
// s1 k1 P + s2 k2 phi(P) by bitwise double-and-add on the generic complete formulas: the fallback of the fast schedules (an addition
// that met P == +-Q with an infinite sum: never for points of G1 and the digit schedules used here).  No table: the cold path
// adds three points to a kernel's scratch frame instead of a 2.3 KB window table.
KZG_HD_NOINLINE static void g1_mul_glv_bits_cold(g1j *o, const g1j *p, const uint32_t *k1, const uint32_t *k2, uint32_t neg1, uint32_t neg2) {
    g1j p1 = *p, p2 = *p;
    p2.x = mul(p2.x, glv_beta());
    if (neg1) p1.y = neg<FpP>(p1.y);
    if (neg2) p2.y = neg<FpP>(p2.y);
    g1j acc = g1_inf();
#pragma nounroll
    for (int b = 127; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((k1[b >> 5] >> (b & 31)) & 1u) acc = g1_add(acc, p1);
        if ((k2[b >> 5] >> (b & 31)) & 1u) acc = g1_add(acc, p2);
    }
    *o = acc;
}
KZG_HD void g1_mul_glv_cold(g1j *o, const g1j *p, const fr *kk) { g1_mul_glv_bits_cold(o, p, &kk->l[0], &kk->l[4], 0, 0); }
KZG_HD bool g1jq_add_slow_copy(g1jq &acc, const g1jq_t *t, bool ng, bool phi) {
    g1jq a2 = acc, q2 = g1jq_entry_point(t, ng, phi);
    bool inf = g1jq_add_slow(&a2, &q2);
    acc = a2;
    return inf;
}
// ---- affine table variant (round 2): the 8 odd multiples are normalised with ONE inversion per scalar multiplication (Montgomery's
// trick inside the lane), so that the ~43 additions of the loop are MIXED additions (3S + 6M + one two-product reduction = 3315
// multiply-adds instead of 4329) and a table entry is 104 bytes of scratch instead of 260.
struct g1aq { fq x, y, bx; };                               // affine point on lazy limbs, bounds (2, 2); bx = beta x: the x of phi(P), kept beside
                                                            // x so that the ~half of the additions that take the phi image skip a product
// acc += (+-) (phi?) *t : madd-2004-hmv on lazy limbs.  acc bounds (19, 20, 4) in, (11, 2, 2) out:
//   Z1Z1 = Z1^2 : 2 (16);  U2 = X2 Z1Z1 : 2;  S2 = (Y2 Z1) Z1Z1 : 2 (8, 4);  H = U2 - X1 (M = 20) : 22;  R = +-S2 - Y1 (M = 21) : 23 resp. 24
//   HH = H^2 : 2 (484 <= 600);  HHH = H HH, V = X1 HH : 2 (44, 38);  X3 = R^2 - HHH - 2 V : 2 + 3 + 3 + 3 = 11  (R^2: 576 <= 600)
//   Y3 = R (V - X3) + (21 p - Y1) HHH in ONE reduction (24 * 14 + 41 * 2 = 418 <= 600) : 2;  Z3 = Z1 H : 2 (88)
// Returns false when H == 0 (P == +-Q): the accumulator is untouched and the caller takes the generic path.
template <bool INL = false> KZG_HD bool g1jq_madd_entry(g1jq &acc, const g1aq *t, bool ng, bool phi) {
    auto MQ = [](const fq &a_, const fq &b_) { return INL ? mulq_inl(a_, b_) : mulq(a_, b_); };
    auto SQ = [](const fq &a_) { return INL ? sqrq_inl(a_) : sqrq(a_); };
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    fq z1z1 = SQ(acc.z);
    fq u2 = MQ(phi ? t->bx : t->x, z1z1);
    fq h = subq<20>(u2, acc.x);
    fq hh = SQ(h);
    if (KZG_UNLIKELY(is_zero_mod_p_q(hh))) return false;
    fq s2 = MQ(MQ(t->y, acc.z), z1z1);
    if (ng) s2 = subq<3>(zero_q, s2);                       // - S2 : 3
    fq r = subq<21>(s2, acc.y);                             // 23 / 24
    fq hhh = MQ(h, hh), v = MQ(acc.x, hh);
    fq x3 = subq<3>(subq<3>(subq<3>(SQ(r), hhh), v), v);
    fq z3 = MQ(acc.z, h);
    if (INL) acc.y = dot2q_inl(r, subq<12>(v, x3), subq<21>(zero_q, acc.y), hhh);
    else acc.y = subq<3>(MQ(r, subq<12>(v, x3)), MQ(acc.y, hhh));   // 2 - (20 * 2 -> 2) : 5
    acc.x = x3; acc.z = z3;
    return true;
}
KZG_HD g1jq g1aq_entry_point(const g1aq *t, bool ng, bool phi) {
    g1jq q; q.x = phi ? t->bx : t->x; q.z = unpackq(one<FpP>());
    if (ng) { fq zero_q;
#pragma unroll
        for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
        q.y = subq<3>(zero_q, t->y); } else q.y = t->y;
    return q;
}
// (Reference construction, kept for tests/host: the kernels build the table by the co-Z chain below.)
// the 8 odd multiples P, 3P, .. 15P as affine points: Jacobian multiples (1 doubling + 7 additions) into `jt`, then one inversion of
// the product of their Z's (binary GCD, ~35 product-equivalents) and 3 products per entry to unwind it, 1S + 3M per entry to scale.
// p must be a finite point of G1 (no Z is zero and no addition degenerates for points of G1; the degenerate branch is kept for safety).
KZG_HD void g1_wnaf_table_affine_q(const g1jq &p0, g1aq *tbl, g1jq *jt) {   // p0: any lazy Jacobian image with bounds (19, 20, 4)
    {
        g1jq cur = p0;
        jt[0] = cur;
        g1jq_t p2;
        g1jq_t_make(&p2, g1jq_dbl(cur));
#pragma nounroll
        for (int i = 1; i < 8; i++) {
            if (!g1jq_add_entry(cur, &p2, false, false)) g1jq_add_slow_copy(cur, &p2, false, false);
            jt[i] = cur;
        }
    }
    // prefix products of the Z's go into the (still unused) x slots of the affine table
    fq run = jt[0].z;
#pragma nounroll
    for (int i = 1; i < 8; i++) { tbl[i].x = run; run = mulq(run, jt[i].z); }
    fq inv_all = unpackq(inv<FpP>(packq(run)));             // 1 / (Z_0 .. Z_7), Montgomery domain in and out
#pragma nounroll
    for (int i = 7; i >= 0; i--) {
        fq zi;
        if (i) { zi = mulq(inv_all, tbl[i].x); inv_all = mulq(inv_all, jt[i].z); } else zi = inv_all;
        fq zi2 = sqrq(zi);
        tbl[i].x = mulq(jt[i].x, zi2);
        tbl[i].y = mulq(jt[i].y, mulq(zi2, zi));
        tbl[i].bx = mulq(tbl[i].x, unpackq(glv_beta()));
    }
}
// The same table by co-Z arithmetic (Meloni; Longa-Miri's precomputation scheme): after the initial doubling, P rescaled to the Z of 2P
// is free -- (4 X Y^2, 8 Y^4, 2 Y Z) -- and every further odd multiple is ONE co-Z addition (5M + 2S instead of 10M + 3S) that also
// rescales 2P to the new Z:
//   T = 2P = (X1, Y1), O = (2i - 1)P = (X2, Y2), same Z:   d = X1 - X2, C = d^2, W1 = X1 C, W2 = X2 C, e = Y1 - Y2, D = e^2,
//   A1 = Y1 (W1 - W2);   (2i + 1)P = (D - W1 - W2,  e (W1 - X3) - A1,  Z d);   T <- (W1, A1, Z d).
// The Z's form a chain Z_{i+1} = Z_i d_i, so ONE inversion of the last one unwinds all of them with a product each:
// 70M + 26S + 1 inversion for the 8 affine multiples instead of 118M + 33S + 1 inversion (30.4k multiply-adds instead of 51.4k).
// Bounds: the doubling leaves T <= (19, 19, 4) and O_1 = (8, 16, 4); T's X and Y are brought to 2 by a product with one so that the
// squares fit (d : 2 + 9 = 11, e : 2 + 17 = 19; 361 <= 600); from then on T = (2, 2), O = (8, 5): d : 11, e : 8, W1 - W2 : 5,
// X3 : 2 + 3 + 3 = 8, W1 - X3 : 11, Y3 : 2 - A1 (M = 3) : 5, Z d : 22 resp. 44.  Returns false if a d vanishes (P of order < 16:
// never in G1); the caller then builds the table the slow way.
template <bool INL = false> KZG_HD bool g1_wnaf_table_affine_coz(const g1jq &p0, g1aq *tbl, fq *dz, fq &zc) {
    auto mulq = [](const fq &a_, const fq &b_) { return INL ? mulq_inl(a_, b_) : kzg::mulq(a_, b_); };   // INL: products inlined (the stage kernels: +x % measured)
    auto sqrq = [](const fq &a_) { return INL ? sqrq_inl(a_) : kzg::sqrq(a_); };
    fq one_q = unpackq(one<FpP>());
    fq tx, ty, z;
    {   // DBLU: T = 2 P and O_1 = P on T's Z
        fq a = sqrq(p0.x), b = sqrq(p0.y), c = sqrq(b), s_ = mulq(p0.x, b);
        fq d = addq(s_, s_); d = addq(d, d);                   // 4 X Y^2 : 8   (= X of the rescaled P)
        fq e = addq(addq(a, a), a);                            // 6
        fq f = sqrq(e);
        fq x3 = subq<17>(f, addq(d, d));                       // 19
        fq c8 = addq(c, c); c8 = addq(c8, c8); c8 = addq(c8, c8);   // 8 Y^4 : 16 (= Y of the rescaled P)
        fq y3 = subq<17>(mulq(e, subq<20>(d, x3)), c8);        // 19
        fq yz = mulq(p0.y, p0.z);
        z = addq(yz, yz);                                      // 4
        tx = mulq(x3, one_q); ty = mulq(y3, one_q);            // bounds 2: the squares of the first co-Z addition must fit
        tbl[0].x = d; tbl[0].y = c8;
    }
    bool ok = true;
#pragma nounroll
    for (int i = 0; i < 7; i++) {                              // O_{i+1} = T + O_i, T rescaled
        const fq x2 = tbl[i].x, y2 = tbl[i].y;
        fq d = subq<9>(tx, x2);                                // 11
        fq cc = sqrq(d);
        ok = ok && !is_zero_mod_p_q(cc);
        fq w1 = mulq(tx, cc), w2 = mulq(x2, cc);
        fq e = subq<17>(ty, y2);                               // 19 (first step), 8 afterwards
        fq dd = sqrq(e);
        fq a1 = mulq(ty, subq<3>(w1, w2));
        fq x3 = subq<3>(subq<3>(dd, w1), w2);                  // 8
        fq y3 = subq<3>(mulq(e, subq<9>(w1, x3)), a1);         // 5
        z = mulq(z, d);
        dz[i] = d;
        tx = w1; ty = a1;
        tbl[i + 1].x = x3; tbl[i + 1].y = y3;
    }
    if (KZG_UNLIKELY(!ok)) return false;
    const fq beta_q = unpackq(glv_beta());
#ifdef KZG_WNAF_TABLE_AFFINE                                // A/B builds: the table normalised to affine points of E with one inversion (round 2)
    fq zi = unpackq(inv<FpP>(packq(z)));                       // 1 / Z_8
#pragma nounroll
    for (int i = 7; i >= 0; i--) {
        fq zi2 = sqrq(zi);
        tbl[i].x = mulq(tbl[i].x, zi2);
        tbl[i].y = mulq(tbl[i].y, mulq(zi2, zi));
        tbl[i].bx = mulq(tbl[i].x, beta_q);
        if (i) zi = mulq(zi, dz[i - 1]);                       // 1 / Z_i = (1 / Z_{i+1}) d_i
    }
    zc = one_q;
#else
    // No inversion: entry i sits at Z_{i+1} = Z_1 d_0 .. d_{i-1}; all are brought to the COMMON Z_8 (times (Z_8 / Z_{i+1})^2, ^3) and read as AFFINE
    // points of the isomorphic curve E': y^2 = x^3 + b Z_8^6 -- the doubling and mixed-addition formulas of an a = 0 curve do not contain b, and
    // (x, y) -> (beta x, y) is the same endomorphism there -- so the multiplication runs on E' unchanged and its result (X, Y, Z) is the point
    // (X, Y, Z Z_8) of E: one product at the end (`zc`) instead of a ~38 000-instruction inversion per multiplication.
    fq r = one_q;
    tbl[7].x = mulq(tbl[7].x, one_q); tbl[7].y = mulq(tbl[7].y, one_q);    // bounds (8, 5) -> 2 like the scaled entries
    tbl[7].bx = mulq(tbl[7].x, beta_q);
#pragma nounroll
    for (int i = 6; i >= 0; i--) {
        r = (i == 6) ? dz[6] : mulq(r, dz[i]);                 // Z_8 / Z_{i+1} = d_i .. d_6
        const fq r2 = sqrq(r);
        tbl[i].x = mulq(tbl[i].x, r2);
        tbl[i].y = mulq(tbl[i].y, mulq(r2, r));
        tbl[i].bx = mulq(tbl[i].x, beta_q);
    }
    zc = z;
#endif
    return true;
}
KZG_HD void g1_wnaf_table_affine(const g1j &p, g1aq *tbl, g1jq *jt) { g1_wnaf_table_affine_q(g1jq_unpack(p), tbl, jt); }
KZG_HD bool g1jq_add_slow_copy_a(g1jq &acc, const g1aq *t, bool ng, bool phi);

// p must not be inf.  `tbl` (8 entries, (2 i + 1) P) lives in the lane's private scratch; the digit arrays d1 / d2 (132 entries
// each, element i at [i * stride]) are caller storage (private arrays; an LDS byte column per lane was measured: the 33 KB per
// workgroup cost more in multi-round launches than the scratch reads it saved).
// Loop shape: the accumulator starts from the top non-zero digit (no infinity flag in the loop), runs of zero digits become a
// tight doubling loop, and an addition happens once per non-zero digit.  If an addition ever meets P == +-Q and the sum is the
// point at infinity, the whole product is redone by the generic g1_mul_glv (cold).
// result in `out` (unpacked, bounds (19, 20, 4)) when the function returns 1; 0: the product is the point at infinity; 2: a degenerate
// addition was met and `packed` holds the product computed by the generic path
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD int g1_mul_glv_wnaf_q(const g1j &p, const fr &kk, g1jq_t *tbl, int8_t *d1, int8_t *d2, int stride, g1jq &out, g1j &packed) {
    {
        g1jq cur = g1jq_unpack(p);
        g1jq_t_make(&tbl[0], cur);
        g1jq_t p2;
        g1jq_t_make(&p2, g1jq_dbl(cur));
#pragma nounroll
        for (int i = 1; i < 8; i++) {
            if (!g1jq_add_entry(cur, &p2, false, false)) g1jq_add_slow_copy(cur, &p2, false, false);   // cannot happen for points of G1
            g1jq_t_make(&tbl[i], cur);
        }
    }
    const int n1 = glv_wnaf5(kk, 0, d1, stride), n2 = glv_wnaf5(kk, 4, d2, stride);
    int j = (n1 > n2 ? n1 : n2) - 1;
    if (j < 0) return 0;                                   // k == 0
    g1jq acc;
    bool degenerate = false;
    {   // top position: at least one of the two digits is non-zero there
        const int a = d1[j * stride], b = d2[j * stride];
        if (a) {
            acc = g1jq_entry_point(&tbl[((a < 0 ? -a : a) - 1) >> 1], a < 0, false);
            if (b) {
                const g1jq_t *t = &tbl[((b < 0 ? -b : b) - 1) >> 1];
                if (!g1jq_add_entry(acc, t, b < 0, true)) degenerate = g1jq_add_slow_copy(acc, t, b < 0, true);
            }
        } else acc = g1jq_entry_point(&tbl[((b < 0 ? -b : b) - 1) >> 1], b < 0, true);
        j--;
    }
    int pend = 0;
#pragma nounroll
    for (; j >= 0 && !degenerate; j--) {
        const int a = d1[j * stride], b = d2[j * stride];
        pend++;
        if (!(a | b)) continue;
#pragma nounroll
        for (; pend > 0; pend--) acc = INL_DBL ? g1jq_dbl_inl(acc) : g1jq_dbl(acc);
#pragma nounroll
        for (int half = 0; half < 2; half++) {
            const int dg = half ? b : a;
            if (!dg || degenerate) continue;
            const g1jq_t *t = &tbl[((dg < 0 ? -dg : dg) - 1) >> 1];
            if (KZG_LIKELY(g1jq_add_entry<INL_ADD>(acc, t, dg < 0, half != 0))) continue;
            degenerate = g1jq_add_slow_copy(acc, t, dg < 0, half != 0);
        }
    }
    if (degenerate) { g1j pc = p; fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
#pragma nounroll
    for (; pend > 0; pend--) acc = g1jq_dbl(acc);
    out = acc;
    return 1;
}
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD g1j g1_mul_glv_wnaf(const g1j &p, const fr &kk, g1jq_t *tbl, int8_t *d1, int8_t *d2, int stride) {
    g1jq q; g1j packed;
    int st = g1_mul_glv_wnaf_q<INL_DBL, INL_ADD>(p, kk, tbl, d1, d2, stride, q, packed);
    return st == 0 ? g1_inf() : st == 1 ? g1jq_pack(q) : packed;
}
KZG_HD bool g1jq_add_slow_copy_a(g1jq &acc, const g1aq *t, bool ng, bool phi) {
    g1jq a2 = acc, q2 = g1aq_entry_point(t, ng, phi);
    bool inf = g1jq_add_slow(&a2, &q2);
    acc = a2;
    return inf;
}
// width-5 NAF GLV multiplication with the AFFINE table (what the G1 FFT stages run since round 2).  Same contract as
// g1_mul_glv_wnaf_q; `dz` is scratch for the 7 Z ratios of the co-Z chain (only alive while the table is built).
// (the multiplicand arrives unpacked: the product of a butterfly's difference in the decimation-in-frequency stages needs no pack / unpack)
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD int g1_wnaf_loop_aq(const g1jq &pq, const fr &kk, const g1aq *tbl, const fq &zc, const int8_t *d1, const int8_t *d2, int stride, int n1, int n2, g1jq &out, g1j &packed) {
    int j = (n1 > n2 ? n1 : n2) - 1;
    if (j < 0) return 0;                                   // k == 0
    g1jq acc;
    bool degenerate = false;
    {   // top position: at least one of the two digits is non-zero there
        const int a = d1[j * stride], b = d2[j * stride];
        if (a) {
            acc = g1aq_entry_point(&tbl[((a < 0 ? -a : a) - 1) >> 1], a < 0, false);
            if (b) {
                const g1aq *t = &tbl[((b < 0 ? -b : b) - 1) >> 1];
                if (!g1jq_madd_entry(acc, t, b < 0, true)) degenerate = g1jq_add_slow_copy_a(acc, t, b < 0, true);
            }
        } else acc = g1aq_entry_point(&tbl[((b < 0 ? -b : b) - 1) >> 1], b < 0, true);
        j--;
    }
    int pend = 0;
#pragma nounroll
    for (; j >= 0 && !degenerate; j--) {
        const int a = d1[j * stride], b = d2[j * stride];
        pend++;
        if (!(a | b)) continue;
#pragma nounroll
        for (; pend > 0; pend--) acc = INL_DBL ? g1jq_dbl_inl(acc) : g1jq_dbl(acc);
#pragma nounroll
        for (int half = 0; half < 2; half++) {
            const int dg = half ? b : a;
            if (!dg || degenerate) continue;
            const g1aq *t = &tbl[((dg < 0 ? -dg : dg) - 1) >> 1];
            if (KZG_LIKELY(g1jq_madd_entry<INL_ADD>(acc, t, dg < 0, half != 0))) continue;
            degenerate = g1jq_add_slow_copy_a(acc, t, dg < 0, half != 0);
        }
    }
    if (degenerate) { g1j pc = g1jq_pack(pq); fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
#pragma nounroll
    for (; pend > 0; pend--) acc = g1jq_dbl(acc);
    out = acc;
    out.z = mulq(out.z, zc);                               // back from the table's isomorphic curve
    return 1;
}
// table for the multiplications below (co-Z chain); `dz`: 7 field elements of scratch.  false: a difference of the chain vanished (P of
// order < 16, never in G1) -- the callers then take the generic path for the whole product.
// zc: the factor the multiplication's result Z has to be multiplied with (the table lives on an isomorphic curve: g1_wnaf_table_affine_coz)
KZG_HD bool g1_wnaf_table(const g1jq &pq, g1aq *tbl, fq *dz, fq &zc) {
#ifdef KZG_WNAF_TABLE_NOINLINE                              // A/B builds: the chain's products as calls (-0.2 .. -0.6 % FK20 measured)
    return g1_wnaf_table_affine_coz<false>(pq, tbl, dz, zc);
#else
    return g1_wnaf_table_affine_coz<true>(pq, tbl, dz, zc);
#endif
}
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD int g1_mul_glv_wnaf_aq_q(const g1jq &pq, const fr &kk, g1aq *tbl, fq *dz, int8_t *d1, int8_t *d2, int stride, g1jq &out, g1j &packed) {
    fq zc;
    if (!g1_wnaf_table(pq, tbl, dz, zc)) { g1j pc = g1jq_pack(pq); fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
    const int n1 = glv_wnaf5(kk, 0, d1, stride), n2 = glv_wnaf5(kk, 4, d2, stride);
    return g1_wnaf_loop_aq<INL_DBL, INL_ADD>(pq, kk, tbl, zc, d1, d2, stride, n1, n2, out, packed);
}
// the digit strings come PRECOMPUTED (the twiddles of an FFTSettings are fixed: their width-5 NAF recodings are built once on the
// host with the same glv_wnaf5 and live in HBM): `dg` = 132 bytes for k1 (digit i at [i], the length at [131]) followed by 132 for k2
#define KZG_WNAF_ROW 264
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD int g1_mul_glv_wnaf_aq_pre_q(const g1jq &pq, const fr &kk, g1aq *tbl, fq *dz, const int8_t *dg, g1jq &out, g1j &packed) {
    fq zc;
    if (!g1_wnaf_table(pq, tbl, dz, zc)) { g1j pc = g1jq_pack(pq); fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
    return g1_wnaf_loop_aq<INL_DBL, INL_ADD>(pq, kk, tbl, zc, dg, dg + 132, 1, (int)(uint8_t)dg[131], (int)(uint8_t)dg[132 + 131], out, packed);
}
// ---- regular variant on the same affine table: what lanes with DIFFERENT scalars run (the direct G1 FFT passes, the late stages of
// small batches) ---------------------------------------------------------------------------------------------------------------------
// Signed odd-digit recoding (Joye-Tunstall): with K = |k| | 1 < 2^128,  K = 16^32 + sum_{i < 32} d_i 16^i,  d_i = (((K >> 4 i) & 31) | 1) - 16,
// every digit odd in [-15, 15]: exactly the 8 odd multiples of the width-5 NAF table, one MIXED addition per digit, no zero digits and
// so no data-dependent branch: 128 doublings + 66 mixed additions per multiplication, then -P for an even |k| (K = |k| + 1).  Against the
// 16-entry Jacobian-table schedule of g1_mul_glv_signed_q (135 doublings, ~52 full additions, table 51k): 518k multiply-adds instead
// of 599k, products inlined, 0.8 KB of table instead of 2.5 KB.  A half that is zero is skipped altogether (lane-varying, rare: small
// scalars).  Same contract as g1_wnaf_loop_aq: 1 = `out` holds the product, 0 = infinity, 2 = `packed` holds it (generic path).
KZG_HD void g1_mul_glv_signed_cold(g1j *o, const g1j *p, const glv_halves *h) { g1_mul_glv_bits_cold(o, p, h->k1, h->k2, h->neg1, h->neg2); }
template <bool INL = true> KZG_HD int g1_mul_glv_regular_aq(const g1jq &pq, const glv_halves &h, g1aq *tbl, fq *dz, g1jq &out, g1j &packed) {
    const bool on1 = (h.k1[0] | h.k1[1] | h.k1[2] | h.k1[3]) != 0, on2 = (h.k2[0] | h.k2[1] | h.k2[2] | h.k2[3]) != 0;
    if (!on1 && !on2) return 0;
    fq zc;
    if (!g1_wnaf_table(pq, tbl, dz, zc)) { g1j pc = g1jq_pack(pq); glv_halves hc = h; g1_mul_glv_signed_cold(&packed, &pc, &hc); return 2; }
    const bool n1 = h.neg1 != 0, n2 = h.neg2 != 0;
    // the digit window of step i is bits 4 i .. 4 i + 4: kept at the top of a 129-bit shift register (bit 128 in the fifth word)
    uint32_t a0 = h.k1[0] | 1u, a1 = h.k1[1], a2 = h.k1[2], a3 = h.k1[3], a4 = 0;
    uint32_t b0 = h.k2[0] | 1u, b1 = h.k2[1], b2 = h.k2[2], b3 = h.k2[3], b4 = 0;
    g1jq acc;
    bool degenerate = false;
    if (on1) {                                              // top digits: +1 for each live half
        acc = g1aq_entry_point(&tbl[0], n1, false);
        if (on2 && !g1jq_madd_entry<INL>(acc, &tbl[0], n2, true)) degenerate = g1jq_add_slow_copy_a(acc, &tbl[0], n2, true);
    } else acc = g1aq_entry_point(&tbl[0], n2, true);
#pragma nounroll
    for (int i = 31; i >= 0 && !degenerate; i--) {
#pragma nounroll
        for (int t = 0; t < 4; t++) acc = INL ? g1jq_dbl_inl(acc) : g1jq_dbl(acc);
        const int da = (int)((((a4 & 1u) << 4) | (a3 >> 28)) | 1u) - 16, db = (int)((((b4 & 1u) << 4) | (b3 >> 28)) | 1u) - 16;
        a4 = a3 >> 28; a3 = (a3 << 4) | (a2 >> 28); a2 = (a2 << 4) | (a1 >> 28); a1 = (a1 << 4) | (a0 >> 28); a0 <<= 4;
        b4 = b3 >> 28; b3 = (b3 << 4) | (b2 >> 28); b2 = (b2 << 4) | (b1 >> 28); b1 = (b1 << 4) | (b0 >> 28); b0 <<= 4;
        if (on1) {
            const g1aq *t = &tbl[((da < 0 ? -da : da) - 1) >> 1];
            if (KZG_UNLIKELY(!g1jq_madd_entry<INL>(acc, t, (da < 0) != n1, false))) degenerate = g1jq_add_slow_copy_a(acc, t, (da < 0) != n1, false);
        }
        if (on2 && !degenerate) {
            const g1aq *t = &tbl[((db < 0 ? -db : db) - 1) >> 1];
            if (KZG_UNLIKELY(!g1jq_madd_entry<INL>(acc, t, (db < 0) != n2, true))) degenerate = g1jq_add_slow_copy_a(acc, t, (db < 0) != n2, true);
        }
    }
    // even halves were recoded as |k| + 1: take the extra (+-)P / (+-)phi(P) off again
    if (on1 && !(h.k1[0] & 1u) && !degenerate && !g1jq_madd_entry<INL>(acc, &tbl[0], !n1, false)) degenerate = g1jq_add_slow_copy_a(acc, &tbl[0], !n1, false);
    if (on2 && !(h.k2[0] & 1u) && !degenerate && !g1jq_madd_entry<INL>(acc, &tbl[0], !n2, true)) degenerate = g1jq_add_slow_copy_a(acc, &tbl[0], !n2, true);
    if (degenerate) { g1j pc = g1jq_pack(pq); glv_halves hc = h; g1_mul_glv_signed_cold(&packed, &pc, &hc); return 2; }
    out = acc;
    out.z = mulq(out.z, zc);                               // back from the table's isomorphic curve
    return 1;
}
KZG_HD void glv_wnaf5_row(const fr &kk, int8_t *row) {      // host side of the above
    int n1 = glv_wnaf5(kk, 0, row, 1), n2 = glv_wnaf5(kk, 4, row + 132, 1);
    row[130] = 0; row[132 + 130] = 0;
    row[131] = (int8_t)n1; row[132 + 131] = (int8_t)n2;
}
template <bool INL_DBL = false, bool INL_ADD = false> KZG_HD int g1_mul_glv_wnaf_aq(const g1j &p, const fr &kk, g1aq *tbl, fq *dz, int8_t *d1, int8_t *d2, int stride, g1jq &out, g1j &packed) {
    return g1_mul_glv_wnaf_aq_q<INL_DBL, INL_ADD>(g1jq_unpack(p), kk, tbl, dz, d1, d2, stride, out, packed);
}

// (P + Q, P - Q) for two finite points, sharing everything but r: add-2007-bl twice is 22M + 10S, this is 13M + 5S (both Y3 as
// two products under one reduction).  P <= (1, 1, 1) (a canonical point), Q <= (19, 20, 4).  Bounds: H = U2 - U1 : 5, I = (2H)^2,
// r = 2 (S2 - S1) : 10, r' = -2 (S2 + S1) : 9, X3 : 11, V - X3 : 14, products <= 150.  False when H == 0 (P == +-Q).
KZG_HD bool g1jq_addsub(const g1jq &p, const g1jq &q, g1jq &sum, g1jq &dif) {
    fq z1z1 = sqrq_inl(p.z), z2z2 = sqrq_inl(q.z);
    fq u1 = mulq_inl(p.x, z2z2), u2 = mulq_inl(q.x, z1z1);
    fq s1 = mulq_inl(mulq_inl(p.y, q.z), z2z2), s2 = mulq_inl(mulq_inl(q.y, p.z), z1z1);
    fq h = subq<3>(u2, u1);
    fq h2 = addq(h, h);
    fq i = sqrq_inl(h2);
    if (KZG_UNLIKELY(is_zero_mod_p_q(i))) return false;
    fq j = mulq_inl(h, i), v = mulq_inl(u1, i);
    fq zz = mulq_inl(mulq_inl(p.z, q.z), h);
    fq z3 = addq(zz, zz);                                  // 4
    fq zq;
#pragma unroll
    for (int k = 0; k < 13; k++) zq.l[k] = 0;
    fq n2s1 = subq<5>(zq, addq(s1, s1));                   // 5 p - 2 S1 : 5
    fq r = subq<3>(s2, s1); r = addq(r, r);                // 10
    fq x3 = subq<3>(subq<3>(subq<3>(sqrq_inl(r), j), v), v);
    sum.x = x3; sum.y = dot2q_inl(r, subq<12>(v, x3), n2s1, j); sum.z = z3;
    fq s12 = addq(s2, s1);                                 // 4
    fq rn = subq<9>(zq, addq(s12, s12));                   // - 2 (S2 + S1) : 9
    fq x3n = subq<3>(subq<3>(subq<3>(sqrq_inl(rn), j), v), v);
    dif.x = x3n; dif.y = dot2q_inl(rn, subq<12>(v, x3n), n2s1, j); dif.z = z3;
    return true;
}

// Plain MSB-first double-and-add (no table); used where the scalar is short.
KZG_HD g1j g1_mul_small(const g1j &p, uint32_t k) {
    g1j acc = g1_inf();
    for (int b = 31; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((k >> b) & 1u) acc = g1_add(acc, p);
    }
    return acc;
}

// Kilic image (bls.G1Point as the Go side holds it) <-> device-internal image; inf keeps Z == 0
KZG_HD g1j g1_from_kilic(const g1j &p) {
    if (is_zero<FpP>(p.z)) return g1_inf();
    g1j o; o.x = fp_from_kilic(p.x); o.y = fp_from_kilic(p.y); o.z = fp_from_kilic(p.z);
    return o;
}
KZG_HD g1j g1_to_kilic(const g1j &p) {
    g1j o;
    if (is_inf(p)) { o.x = zero<FpP>(); o.y = fp_kilic_one(); o.z = zero<FpP>(); return o; }   // Kilic Zero(): (0, 1, 0)
    o.x = fp_to_kilic(p.x); o.y = fp_to_kilic(p.y); o.z = fp_to_kilic(p.z);
    return o;
}

}  // namespace kzg
