// g1_quad.hpp -- QUAD-COOPERATIVE scalar multiplication: four adjacent lanes of one wavefront work on ONE multiplication.
//
// For launches with fewer butterflies than a quarter of the resident lanes (a G1 transform stage of at most 8 polynomials of 4096 points: 16 384
// butterflies on a chip that holds 65 536 lanes at one wavefront per SIMD) the stage time is the LATENCY of one scalar multiplication on one
// lane: ~1 370 dependent F_p products.  The products inside a group operation are mostly independent of each other: an XYZZ doubling is 9
// products in 3 dependency levels, an XYZZ + affine addition 10 products in 4.  Here the four lanes of a quad hold replicas of the running
// point and each computes ONE product of a level -- all lanes execute the same instruction stream (one 13-limb product), only their operands
// differ (selected by the lane's role, v_cndmask) -- and the results are exchanged with DPP quad_perm broadcasts (v_mov_b32 ... quad_perm:[r,r,r,r],
// 13 moves per value): no LDS, no barrier, no divergence.  128 doublings + 66 additions = 648 levels instead of ~1 370 products on the chain.
// Replaces the per-butterfly bls.MulG1 of fft_g1.go:49 for those launches; the schedule (signed odd digits of the two GLV halves on the affine
// co-Z table) and all values are those of g1_mul_glv_regular_aq, so the results are the same group elements.
// The table construction and everything outside the digit loop is computed redundantly by the four lanes.
#pragma once
#include "g1.hpp"

namespace kzg {

#if defined(__HIPCC__)
// every lane of the group of L adjacent lanes (a quad, or one of its two pairs) receives the value of the group's lane R
template <int R, int L = 4> __device__ __forceinline__ fq quad_bcast(const fq &v) {
    constexpr int ctrl = L == 4 ? R * 0x55 : (R | (R << 2) | ((2 + R) << 4) | ((2 + R) << 6));   // quad_perm [R, R, R, R] / [R, R, 2 + R, 2 + R]
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)v.l[i], (int)v.l[i], ctrl, 0xf, 0xf, false);
    return o;
}
__device__ __forceinline__ fq quad_sel(uint32_t role, const fq &a0, const fq &a1, const fq &a2, const fq &a3) {
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) { const uint32_t lo = role & 1u ? a1.l[i] : a0.l[i], hi = role & 1u ? a3.l[i] : a2.l[i]; o.l[i] = role & 2u ? hi : lo; }
    return o;
}
__device__ __forceinline__ fq quad_sel2(uint32_t role, const fq &a0, const fq &a1) {   // roles 2, 3 repeat 0, 1 (idle lanes of a two-product level)
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = role & 1u ? a1.l[i] : a0.l[i];
    return o;
}
// One dependency level of up to four independent products, the results on every lane of the group.  L = 4: one product per lane, one round.
// L = 2 (pairs: launches too large for quads but with SIMDs to spare): products 0, 1 in a first round, 2, 3 in a second.
template <int L> __device__ __forceinline__ void coop_mul4(uint32_t role, const fq &a0, const fq &a1, const fq &a2, const fq &a3, const fq &b0, const fq &b1,
                                                           const fq &b2, const fq &b3, fq &p0, fq &p1, fq &p2, fq &p3) {
    if constexpr (L == 4) {
        const fq mine = mulq_inl(quad_sel(role, a0, a1, a2, a3), quad_sel(role, b0, b1, b2, b3));
        p0 = quad_bcast<0>(mine); p1 = quad_bcast<1>(mine); p2 = quad_bcast<2>(mine); p3 = quad_bcast<3>(mine);
    } else {
        const fq m0 = mulq_inl(quad_sel2(role, a0, a1), quad_sel2(role, b0, b1));
        p0 = quad_bcast<0, 2>(m0); p1 = quad_bcast<1, 2>(m0);
        const fq m1 = mulq_inl(quad_sel2(role, a2, a3), quad_sel2(role, b2, b3));
        p2 = quad_bcast<0, 2>(m1); p3 = quad_bcast<1, 2>(m1);
    }
}
template <int L> __device__ __forceinline__ void coop_mul2(uint32_t role, const fq &a0, const fq &a1, const fq &b0, const fq &b1, fq &p0, fq &p1) {
    const fq mine = mulq_inl(quad_sel2(role, a0, a1), quad_sel2(role, b0, b1));
    p0 = quad_bcast<0, L>(mine); p1 = quad_bcast<1, L>(mine);
}
template <int L> __device__ __forceinline__ void coop_sqr2(uint32_t role, const fq &a0, const fq &a1, fq &p0, fq &p1) {
    const fq mine = sqrq_inl(quad_sel2(role, a0, a1));
    p0 = quad_bcast<0, L>(mine); p1 = quad_bcast<1, L>(mine);
}
// p <- 2 p (dbl-2008-s-1, a = 0), bounds (X, Y, ZZ, ZZZ) <= (11, 5, 2, 2) in and out -- the levels of coop_xyzz_dbl (k_msm.hip):
//   U = 2 Y;  L1: V = U^2, XX = X^2;  M = 3 XX;  L2: W = U V, S = X V, ZZ' = V ZZ, MM = M^2;  X' = MM - 2 S;
//   L3: T1 = M (S - X'), T2 = W Y, ZZZ' = W ZZZ;  Y' = T1 - T2
template <int L = 4> __device__ __forceinline__ void quad_xyzz_dbl(g1xq &p, uint32_t role) {
    const fq u = addq(p.y, p.y);                                               // 10
    fq v, xx, w, s_, zz3, mm, t1, t2, zzz3, unused;
    coop_sqr2<L>(role, u, p.x, v, xx);
    const fq m = addq(addq(xx, xx), xx);                                       // 6
    coop_mul4<L>(role, u, p.x, v, m, v, v, p.zz, m, w, s_, zz3, mm);
    const fq x3 = subq<5>(mm, addq(s_, s_));                                   // 7
    coop_mul4<L>(role, m, w, w, w, subq<8>(s_, x3), p.y, p.zzz, p.zzz, t1, t2, zzz3, unused);
    p.x = x3; p.y = subq<3>(t1, t2); p.zz = zz3; p.zzz = zzz3;              // (7, 5, 2, 2)
}
// p <- p + (x2, y2) for an affine point with bounds (2, 3) (madd-2008-s), accumulator bounds as above.  Returns false -- p untouched -- when
// P == +-Q (the caller takes the generic path); the verdict is the same on the four lanes.
//   L1: U2 = x2 ZZ, S2 = y2 ZZZ;  P = U2 - X (14), R = S2 - Y (8);  L2: PP = P^2, RR = R^2;  L3: PPP = P PP, Q = X PP, ZZ' = ZZ PP;
//   X3 = RR - PPP - 2 Q (11);  L4: T1 = R (Q - X3), T2 = (6 p - Y) PPP, ZZZ' = ZZZ PPP;  Y3 = T1 + T2 (4)
template <int L = 4> __device__ __forceinline__ bool quad_xyzz_madd(g1xq &p, const fq &x2, const fq &y2, uint32_t role) {
    fq u2, s2, pp, rr, ppp, q_, zz3, t1, t2, zzz3, unused;
    coop_mul2<L>(role, x2, y2, p.zz, p.zzz, u2, s2);
    const fq pp_ = subq<12>(u2, p.x), r = subq<6>(s2, p.y);
    coop_sqr2<L>(role, pp_, r, pp, rr);
    if (KZG_UNLIKELY(is_zero_mod_p_q(pp))) return false;
    coop_mul4<L>(role, pp_, p.x, p.zz, p.zz, pp, pp, pp, pp, ppp, q_, zz3, unused);
    const fq x3 = subq<3>(subq<3>(subq<3>(rr, ppp), q_), q_);
    fq zero_q;
#pragma unroll
    for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
    coop_mul4<L>(role, r, subq<6>(zero_q, p.y), p.zzz, p.zzz, subq<12>(q_, x3), ppp, ppp, ppp, t1, t2, zzz3, unused);
    p.x = x3; p.y = addq(t1, t2); p.zz = zz3; p.zzz = zzz3;
    return true;
}
// a <- a + b for two XYZZ points (add-2008-s), bounds as in g1xq_add_fast / coop_xyzz_add (k_msm.hip): both operands (11, 5, 2, 2), result the same.
// Returns false (a untouched) when the operands are equal or opposite.
//   L1: U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1;  P = U2 - U1 (5), R = S2 - S1 (5);  L2: PP = P^2, RR = R^2, ZZ12 = ZZ1 ZZ2, ZZZ12 = ZZZ1 ZZZ2;
//   L3: PPP = P PP, Q = U1 PP, ZZ3 = ZZ12 PP;  X3 = RR - PPP - 2 Q (11);  L4: T1 = R (Q - X3), T2 = S1 PPP, ZZZ3 = ZZZ12 PPP;  Y3 = T1 - T2 (5)
template <int L = 4> __device__ __forceinline__ bool quad_xyzz_add(g1xq &a, const g1xq &b, uint32_t role) {
    fq u1, u2, s1, s2, pp, rr, zz12, zzz12, ppp, q_, zz3, t1, t2, zzz3, unused;
    coop_mul4<L>(role, a.x, b.x, a.y, b.y, b.zz, a.zz, b.zzz, a.zzz, u1, u2, s1, s2);
    const fq pp_ = subq<3>(u2, u1), r = subq<3>(s2, s1);
    coop_mul4<L>(role, pp_, r, a.zz, a.zzz, pp_, r, b.zz, b.zzz, pp, rr, zz12, zzz12);
    if (KZG_UNLIKELY(is_zero_mod_p_q(pp))) return false;
    coop_mul4<L>(role, pp_, u1, zz12, zz12, pp, pp, pp, pp, ppp, q_, zz3, unused);
    const fq x3 = subq<3>(subq<3>(subq<3>(rr, ppp), q_), q_);
    coop_mul4<L>(role, r, s1, zzz12, zzz12, subq<12>(q_, x3), ppp, ppp, ppp, t1, t2, zzz3, unused);
    a.x = x3; a.y = subq<3>(t1, t2); a.zz = zz3; a.zzz = zzz3;
    return true;
}
// acc += w for the replicated accumulators of a quad (infinity flags beside the limbs); equal / opposite operands take the generic complete
// formulas, identically on the four lanes.  Lanes with nothing to add (winf) still run the levels on whatever `w` holds and drop the result.
template <int L = 4> __device__ __forceinline__ void quad_acc_add(g1x_acc &acc, const g1xq &w, bool winf, uint32_t role) {
    g1xq sum = acc.v;
    const bool ok = quad_xyzz_add<L>(sum, w, role);
    if (winf) return;
    if (acc.inf) { acc.v = w; acc.inf = false; }
    else if (ok) acc.v = sum;
    else g1x_acc_merge(acc, w, false);
}
// the table entry (+-)(phi?) tbl[i] as the operands of quad_xyzz_madd
__device__ __forceinline__ void quad_entry(const g1aq *t, bool ng, bool phi, fq &x2, fq &y2) {
    x2 = phi ? t->bx : t->x;
    y2 = t->y;
    if (ng) { fq zero_q;
#pragma unroll
        for (int i = 0; i < 13; i++) zero_q.l[i] = 0;
        y2 = subq<3>(zero_q, y2); }
}
// The regular odd-digit GLV multiplication of g1_mul_glv_regular_aq (same digits, same table, same return contract: 1 = `out` holds the product
// as a lazy Jacobian image, 0 = infinity, 2 = `packed` holds it), with the digit loop on a quad.  All four lanes of the quad pass the same
// arguments and receive the same result.
template <int L = 4> __device__ __forceinline__ int g1_mul_glv_regular_quad(const g1jq &pq, const glv_halves &h, g1aq *tbl, fq *dz, g1jq &out, g1j &packed, uint32_t role) {
    const bool on1 = (h.k1[0] | h.k1[1] | h.k1[2] | h.k1[3]) != 0, on2 = (h.k2[0] | h.k2[1] | h.k2[2] | h.k2[3]) != 0;
    if (!on1 && !on2) return 0;
    fq zc;
    if (!g1_wnaf_table(pq, tbl, dz, zc)) { g1j pc = g1jq_pack(pq); glv_halves hc = h; g1_mul_glv_signed_cold(&packed, &pc, &hc); return 2; }
    const bool n1 = h.neg1 != 0, n2 = h.neg2 != 0;
    uint32_t a0 = h.k1[0] | 1u, a1 = h.k1[1], a2 = h.k1[2], a3 = h.k1[3], a4 = 0;
    uint32_t b0 = h.k2[0] | 1u, b1 = h.k2[1], b2 = h.k2[2], b3 = h.k2[3], b4 = 0;
    g1xq acc;
    fq x2, y2;
    bool degenerate = false;
    quad_entry(&tbl[0], on1 ? n1 : n2, !on1, x2, y2);                          // top digits: +1 for each live half
    acc.x = x2; acc.y = y2; acc.zz = unpackq(one<FpP>()); acc.zzz = acc.zz;
    if (on1 && on2) { quad_entry(&tbl[0], n2, true, x2, y2); degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role); }
#pragma nounroll
    for (int i = 31; i >= 0 && !degenerate; i--) {
#pragma nounroll
        for (int t = 0; t < 4; t++) quad_xyzz_dbl<L>(acc, role);
        const int da = (int)((((a4 & 1u) << 4) | (a3 >> 28)) | 1u) - 16, db = (int)((((b4 & 1u) << 4) | (b3 >> 28)) | 1u) - 16;
        a4 = a3 >> 28; a3 = (a3 << 4) | (a2 >> 28); a2 = (a2 << 4) | (a1 >> 28); a1 = (a1 << 4) | (a0 >> 28); a0 <<= 4;
        b4 = b3 >> 28; b3 = (b3 << 4) | (b2 >> 28); b2 = (b2 << 4) | (b1 >> 28); b1 = (b1 << 4) | (b0 >> 28); b0 <<= 4;
        if (on1) {
            quad_entry(&tbl[((da < 0 ? -da : da) - 1) >> 1], (da < 0) != n1, false, x2, y2);
            degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role);
        }
        if (on2 && !degenerate) {
            quad_entry(&tbl[((db < 0 ? -db : db) - 1) >> 1], (db < 0) != n2, true, x2, y2);
            degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role);
        }
    }
    // even halves were recoded as |k| + 1: take the extra (+-)P / (+-)phi(P) off again
    if (on1 && !(h.k1[0] & 1u) && !degenerate) { quad_entry(&tbl[0], !n1, false, x2, y2); degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role); }
    if (on2 && !(h.k2[0] & 1u) && !degenerate) { quad_entry(&tbl[0], !n2, true, x2, y2); degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role); }
    if (degenerate) { g1j pc = g1jq_pack(pq); glv_halves hc = h; g1_mul_glv_signed_cold(&packed, &pc, &hc); return 2; }   // (a degenerate addition never ends at infinity silently)
    // XYZZ -> the Jacobian image (X ZZ, Y ZZZ, ZZ): two more products, one level
    coop_mul2<L>(role, acc.x, acc.y, acc.zz, acc.zzz, out.x, out.y);
    out.z = mulq_inl(acc.zz, zc);                          // ... and back from the table's isomorphic curve
    return 1;
}
// The width-5 NAF schedule of g1_wnaf_loop_aq on a quad, digits from the twiddle's precomputed row (KZG_WNAF_ROW bytes: 132 for k1 -- digit i at
// [i], the length at [131] -- then 132 for k2): 128 doublings + ~43 additions = ~556 levels.  For launches whose wavefronts hold ONE twiddle (the
// zero-digit runs are data-dependent branches).  Same return contract.
template <int L = 4> __device__ __forceinline__ int g1_mul_glv_wnaf_quad(const g1jq &pq, const fr &kk, g1aq *tbl, fq *dz, const int8_t *dg, g1jq &out, g1j &packed, uint32_t role) {
    fq zc;
    if (!g1_wnaf_table(pq, tbl, dz, zc)) { g1j pc = g1jq_pack(pq); fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
    const int8_t *d1 = dg, *d2 = dg + 132;
    const int n1 = (int)(uint8_t)dg[131], n2 = (int)(uint8_t)dg[132 + 131];
    int j = (n1 > n2 ? n1 : n2) - 1;
    if (j < 0) return 0;
    g1xq acc; fq x2, y2;
    bool degenerate = false;
    {
        const int a = d1[j], b = d2[j];
        const int first = a ? a : b;
        quad_entry(&tbl[((first < 0 ? -first : first) - 1) >> 1], first < 0, a == 0, x2, y2);
        acc.x = x2; acc.y = y2; acc.zz = unpackq(one<FpP>()); acc.zzz = acc.zz;
        if (a && b) { quad_entry(&tbl[((b < 0 ? -b : b) - 1) >> 1], b < 0, true, x2, y2); degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role); }
        j--;
    }
    int pend = 0;
#pragma nounroll
    for (; j >= 0 && !degenerate; j--) {
        const int a = d1[j], b = d2[j];
        pend++;
        if (!(a | b)) continue;
#pragma nounroll
        for (; pend > 0; pend--) quad_xyzz_dbl<L>(acc, role);
#pragma nounroll
        for (int half = 0; half < 2; half++) {
            const int dgt = half ? b : a;
            if (!dgt || degenerate) continue;
            quad_entry(&tbl[((dgt < 0 ? -dgt : dgt) - 1) >> 1], dgt < 0, half != 0, x2, y2);
            degenerate = !quad_xyzz_madd<L>(acc, x2, y2, role);
        }
    }
    if (degenerate) { g1j pc = g1jq_pack(pq); fr kc = kk; g1_mul_glv_cold(&packed, &pc, &kc); return 2; }
#pragma nounroll
    for (; pend > 0; pend--) quad_xyzz_dbl<L>(acc, role);
    coop_mul2<L>(role, acc.x, acc.y, acc.zz, acc.zzz, out.x, out.y);
    out.z = mulq_inl(acc.zz, zc);                          // ... and back from the table's isomorphic curve
    return 1;
}
#endif

}  // namespace kzg
