// coop_inv.hpp -- a WAVE-COOPERATIVE F_p inversion for the latency paths (round 6).
//
// Every affine result of the path (bls.LinCombG1's return value normalised, bls/bls_kilic.go:132-150; the proofs of fk20_single.go:122-134) ends in one F_p inversion.
// inv() of field.hpp is one lane's work: Pornin's binary GCD, 26 rounds x (30 dependent 64-bit steps + a 13-limb matrix application) = ~38 k dependent instructions,
// ~100 us whatever else the machine does -- 0.11 of a lone commitment's 0.30 ms.  Nothing in it is wide: a wavefront that waits for it idles 63 lanes.
//
// This form spreads the WIDE half over lanes and shrinks the SERIAL half:
//   * the numbers live limb-per-lane: lane j of a row of 16 holds limb j (30 bits, signed, centred: |limb| <= 2^29 + 2) of f, g, d, e.  Applying a round's 2 x 2
//     matrix to (f, g) and to (d, e) is 4 + 6 v_mad_i64_i32 per lane and TWO carry hand-overs to the neighbour lane (DPP row shifts) instead of four 13-limb carry
//     chains: ~70 instructions per round instead of ~650.  Limbs are never fully normalised between rounds -- a centred signed limb below 2^30 in magnitude has a
//     unique representation of zero (sum_j w_j 2^(30 j) = 0 forces w_0 = 0 mod 2^30, hence w_0 = 0, and so on upwards), so "g == 0" is one ballot, and the low 30
//     bits of f and g -- all the next round needs -- are exact in limb 0 whatever the redundancy above.
//   * the serial half is Bernstein-Yang's divsteps ("safegcd", the form of libsecp256k1's modinv32: eta = -delta, 30 divsteps per round on the LOW 30 bits only --
//     no window of top bits, no comparison of 64-bit approximations, no negation of rows), in its variable-time form: trailing zeros of g are skipped with one
//     count-trailing-zeros, up to six low bits of g are cancelled per iteration with w = f g (f f - 2) mod 2^k (f (f f - 2) = -1 / f mod 64).  The operands are
//     wave-uniform (read from lane 0), so the loop runs on the scalar unit with real branches: 7.7 iterations per round on average (measured over 3 000 random
//     elements: 26-28 rounds, 27 on average -- the loop runs until g == 0, no iteration bound is assumed).
//   * (d, e) <- M (d, e) / 2^30 mod p adds the multiple of p that clears the low limb (md, me in (-2^29, 2^29], computed on the scalar unit from lane 0's limbs);
//     no sign-dependent correction keeps d, e small: they grow by at most p / 2 + |.| per round, |d| < 32 p after 30 rounds, inside the 390 bits of 13 limbs; ONE
//     final quotient estimate (single precision, from the two top limbs) brings d into (-p, p).
//   Result: x R' -> x^-1 R' (0 -> 0), the same canonical value inv<FpP>() returns -- bit-exactness of every caller is unaffected by which of the two ran.
//
// wave_inv_fp() must be called by ALL 64 lanes of a wavefront in converged control flow; the operand is taken from lane `src` (wave-uniform), the result is
// wave-uniform.  wave_inv_any() is the drop-in for call sites where some lanes of a wavefront hold operands: up to KZG_COOP_INV_MAX of them are served one after the
// other by the cooperative form, more than that run the lane form in parallel as before.
// The scalar pieces (divsteps, carry split, final canonicalisation) are plain C++ shared with tests/host/host_emul.cpp, which replays the lane protocol over arrays.
#pragma once
#include "field.hpp"

namespace kzg {
namespace cinv {

static constexpr uint32_t M30 = 0x3fffffffu;

KZG_HD int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }           // the low 30 bits, sign-extended (v_bfe_i32)
KZG_HD uint32_t limb30(const uint32_t *w, int nwords, int k) {                  // bits [30 k, 30 k + 30) of a little-endian word array
    const int i = (30 * k) >> 5, sh = (30 * k) & 31;
    uint64_t v = i < nwords ? w[i] : 0u;
    if (i + 1 < nwords) v |= (uint64_t)w[i + 1] << 32;
    return (uint32_t)(v >> sh) & M30;
}
template <class F> KZG_HD uint32_t r2_limb30(int k) {                            // R^2 mod m (R the field's Montgomery radix) in 30-bit limbs
    uint32_t w[F::N];
#pragma unroll
    for (int i = 0; i < F::N; i++) w[i] = F::r2(i);
    return limb30(w, F::N, k);
}
// -m^-1 mod 2^30 is F::INV30; the rounds need the multiple of m that CLEARS the low limb: md = -cd m^-1 = cd INV30 (mod 2^30)

// Up to 30 divsteps on the low 30 bits of f (odd) and g, variable time.  On return (u v; q r) is the transition matrix scaled by 2^30: M (f, g) = 2^30 (f', g').
// |u| + |v| <= 2^30, |q| + |r| <= 2^30.  eta = -delta.
KZG_HD void divsteps30_var(int32_t &eta, uint32_t f, uint32_t g, int32_t &u, int32_t &v, int32_t &q, int32_t &r) {
    uint32_t uu = 1, vv = 0, qq = 0, rr = 1;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));               // the sentinel bit stops the count at the steps that are left
        g >>= zeros; uu <<= zeros; vv <<= zeros; eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {                                                          // delta > 0 and g odd: (f, g) <- (g, -f)
            uint32_t t;
            eta = -eta;
            t = f; f = g; g = 0u - t;
            t = uu; uu = qq; qq = 0u - t;
            t = vv; vv = rr; rr = 0u - t;
        }
        const int limit = eta + 1 < i ? eta + 1 : i;                            // bits of g that can be cancelled before eta changes sign / the round ends
        const uint32_t m = (0xffffffffu >> (32 - limit)) & 63u;
        const uint32_t w = (f * g * (f * f - 2u)) & m;                          // -g / f mod 2^min(limit, 6)
        g += f * w; qq += uu * w; rr += vv * w;
    }
    u = (int32_t)uu; v = (int32_t)vv; q = (int32_t)qq; r = (int32_t)rr;
}

// t = lo + 2^30 hi with lo the centred low 30 bits
KZG_HD void split64(int64_t t, int32_t &lo, int32_t &hi) {
    lo = sext30((uint32_t)t);
    hi = (int32_t)(t >> 30) + (int32_t)(((uint32_t)t >> 29) & 1u);              // (t - lo) >> 30; fits: |t| < 2^61
}
KZG_HD void split32(int32_t t, int32_t &lo, int32_t &hi) {
    lo = sext30((uint32_t)t);
    hi = (t - lo) >> 30;
}
// the multiples of p that clear the low limb of u d + v e and q d + r e: centred, from the low limbs of d and e
template <class F> KZG_HD void de_multipliers(int32_t u, int32_t v, int32_t q, int32_t r, int32_t d0, int32_t e0, int32_t &md, int32_t &me) {
    const uint32_t cd = (uint32_t)u * (uint32_t)d0 + (uint32_t)v * (uint32_t)e0, ce = (uint32_t)q * (uint32_t)d0 + (uint32_t)r * (uint32_t)e0;
    md = sext30(cd * F::INV30); me = sext30(ce * F::INV30);
}
// the quotient estimate of the final reduction: round(d / m) from the two top limbs, single precision (|d| < 32 m: |error| < 2^-10)
template <class F> KZG_HD int32_t final_quotient(int32_t d_top, int32_t d_next) {
    const float D = (float)d_top * 1073741824.0f + (float)d_next;
    const float inv_mt = (float)(1.0 / ((double)F::p30(F::N30 - 1) * 1073741824.0 + (double)F::p30(F::N30 - 2)));
    return (int32_t)__builtin_rintf(D * inv_mt);
}
// centred limbs w[0 .. L-1] of a value in (-m, m) -> the canonical image in [0, m)
template <class F> KZG_HD felem<F> canonical_from_centred(const int32_t *w) {
    constexpr int L = F::N30;
    uint32_t a[L];
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const int32_t s = w[j] + c;
        if (j < L - 1) { a[j] = (uint32_t)s & M30; c = s >> 30; } else a[j] = (uint32_t)s;     // top limb keeps the sign
    }
    const uint32_t neg = (uint32_t)((int32_t)a[L - 1] >> 31);                   // all ones: add p once
    uint32_t k = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const uint32_t s = a[j] + (F::p30(j) & neg) + k;
        if (j < L - 1) { a[j] = s & M30; k = s >> 30; } else a[j] = s;
    }
    felem<F> out;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const int k0 = (32 * i) / 30, o = (32 * i) % 30;
        uint64_t acc = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) if (k0 + t < L) acc |= (uint64_t)a[k0 + t] << (30 * t);
        out.l[i] = (uint32_t)(acc >> o);
    }
    return out;
}

#if defined(__HIPCC__)
__device__ __forceinline__ int32_t from_next_lane(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0x101, 0xf, 0xf, true); }   // row_shl:1: lane i <- lane i + 1 (0 at the row's end)
__device__ __forceinline__ int32_t from_prev_lane(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); }   // row_shr:1: lane i <- lane i - 1 (0 at the row's start)
// (sum of the 13 lanes' t_j 2^(30 j)) / 2^30 as centred limbs, one per lane: two hand-overs, no carry chain
__device__ __forceinline__ int32_t carry_div30(int64_t t) {
    int32_t lo, hi, lo2, hi2;
    split64(t, lo, hi);
    const int32_t v = from_next_lane(lo) + hi;                                  // limb j of the quotient before the second hand-over: lo(t_{j+1}) + hi(t_j)
    split32(v, lo2, hi2);
    return lo2 + from_prev_lane(hi2);
}
// the same without the division (the final d - q p)
__device__ __forceinline__ int32_t carry_keep(int64_t t) {
    int32_t lo, hi, lo2, hi2;
    split64(t, lo, hi);
    const int32_t v = lo + from_prev_lane(hi);
    split32(v, lo2, hi2);
    return lo2 + from_prev_lane(hi2);
}
#endif

}  // namespace cinv

#if defined(__HIPCC__)
#ifndef KZG_COOP_INV_MAX
#define KZG_COOP_INV_MAX 3      // a wavefront serves up to this many of its lanes' operands cooperatively, one after the other (~20-25 us each); more: the lane form
#endif
// x R -> x^-1 R (0 -> 0; R the field's Montgomery radix) of lane `src`'s x, computed by the whole wavefront; every lane receives the result.  ALL 64 lanes must be active.
// F = FpP (13 limbs, R' = 2^390) or FrP (9 limbs, R = 2^256: 270 bits hold |d| < 32 r just as 390 hold 32 p).
template <class F> __device__ __noinline__ felem<F> wave_inv(const felem<F> &x, uint32_t src) {
    using namespace cinv;
    constexpr int L = F::N30;
    uint32_t xs[F::N];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) { xs[i] = (uint32_t)__builtin_amdgcn_readlane((int)x.l[i], (int)src); any |= xs[i]; }
    if (any == 0) return zero<F>();
    const uint32_t j = threadIdx.x & 15u;                                       // this lane's limb (the lanes above L - 1 of a row hold zeros; rows 1..3 replicate row 0)
    int32_t f = 0, g = 0, d = 0, e = 0, pj = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
        const uint32_t gk = limb30(xs, F::N, k);                                // wave-uniform
        if (j == (uint32_t)k) { g = (int32_t)gk; pj = (int32_t)F::p30(k); e = (int32_t)r2_limb30<F>(k); }
    }
    f = pj;
    int32_t eta = -1;
    while (__builtin_amdgcn_ballot_w64(g != 0) != 0) {
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane(f, 0), g0 = (uint32_t)__builtin_amdgcn_readlane(g, 0);
        int32_t u, v, q, r;
        divsteps30_var(eta, f0, g0, u, v, q, r);                                // scalar unit: everything in it is wave-uniform
        const int64_t tf = (int64_t)u * f + (int64_t)v * g, tg = (int64_t)q * f + (int64_t)r * g;
        int32_t md, me;
        de_multipliers<F>(u, v, q, r, __builtin_amdgcn_readlane(d, 0), __builtin_amdgcn_readlane(e, 0), md, me);
        const int64_t td = ((int64_t)u * d + (int64_t)v * e) + (int64_t)md * pj, te = ((int64_t)q * d + (int64_t)r * e) + (int64_t)me * pj;
        f = carry_div30(tf); g = carry_div30(tg);
        d = carry_div30(td); e = carry_div30(te);
    }
    // f = +-1 (limb 0, exactly: the representation is unique below 2^30): the inverse is sign(f) d mod m
    if (__builtin_amdgcn_readlane(f, 0) < 0) d = -d;
    const int32_t qe = final_quotient<F>(__builtin_amdgcn_readlane(d, L - 1), __builtin_amdgcn_readlane(d, L - 2));
    d = carry_keep((int64_t)d - (int64_t)qe * pj);                              // in (-m, m)
    int32_t w[L];
#pragma unroll
    for (int k = 0; k < L; k++) w[k] = __builtin_amdgcn_readlane(d, k);
    return canonical_from_centred<F>(w);
}
__device__ __forceinline__ fp wave_inv_fp(const fp &x, uint32_t src) { return wave_inv<FpP>(x, src); }
// Inverses of one non-zero value per lane of a WORKGROUP of 2^LOG lanes (LOG <= 10) with ONE inversion: a product tree in LDS (heap layout: node k has children 2 k and
// 2 k + 1, the lanes' values are the leaves at [2^LOG, 2^(LOG+1))), the root inverted by the first wavefront cooperatively, then the inverses pushed down
// (inv(left) = inv(node) right, inv(right) = inv(node) left).  2 LOG barriers + one cooperative inversion instead of one lane-form inversion per lane: the quotient
// kernel of eth.ComputeKZGProof has 1 024 lanes, i.e. 16 wavefronts on one CU that each spent a full binary GCD.  `tree` holds 2^(LOG+1) elements.  Every lane calls.
// `side` runs on the SECOND wavefront while the first one inverts (an independent chain that would otherwise cost every lane its issue slots: z^n in the quotient kernel).
struct no_side_job { __device__ void operator()() const {} };
template <class F, int LOG, class Side = no_side_job> __device__ __forceinline__ felem<F> block_batch_inverse(const felem<F> &v, felem<F> *tree, uint32_t tid, Side side = Side()) {
    constexpr uint32_t LANES = 1u << LOG;
    tree[LANES + tid] = v;
    __syncthreads();
#pragma nounroll
    for (uint32_t off = LANES >> 1; off >= 1; off >>= 1) {
        if (tid < off) tree[off + tid] = mul(tree[2 * (off + tid)], tree[2 * (off + tid) + 1]);
        __syncthreads();
    }
    if (tid < 64) {                                                            // the first wavefront, whole
        const felem<F> ri = wave_inv<F>(tree[1], 0);
        if (tid == 0) tree[1] = ri;
    } else if (tid < 128) side();
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 1; off < LANES; off <<= 1) {
        if (tid < off) {
            const uint32_t node = off + tid;
            const felem<F> in = tree[node], l = tree[2 * node], r = tree[2 * node + 1];
            tree[2 * node] = mul(in, r); tree[2 * node + 1] = mul(in, l);
        }
        __syncthreads();
    }
    return tree[LANES + tid];
}
// Call-site form: lanes with need == true hold an operand; returns its inverse to each of them (other lanes: unspecified).  ALL 64 lanes must call it together.
__device__ __forceinline__ fp wave_inv_any(const fp &x, bool need) {
    uint64_t mask = __builtin_amdgcn_ballot_w64(need);
    fp out = zero<FpP>();
    if (mask == 0) return out;
    if (__builtin_popcountll(mask) > KZG_COOP_INV_MAX) {                        // many operands: one lane each, in parallel (inv() is branch-free and lane-uniform)
        if (need) out = inv<FpP>(x);
        return out;
    }
    const uint32_t lane = threadIdx.x & 63u;
    while (mask) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        const fp y = wave_inv_fp(x, src);
        if (lane == src) out = y;
    }
    return out;
}
#endif

}  // namespace kzg
