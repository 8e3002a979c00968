// field.hpp -- BLS12-381 prime-field arithmetic for gfx950 lanes (one field element per lane).
//
// Replaces the per-element bls.MulModFr / AddModFr / SubModFr calls of the reference's default backend
// (bls/bignum_kilic.go:95-111 -> Kilic Fr.RedMul/Add/Sub) and Kilic's `fe` (F_p) arithmetic underneath
// bls.AddG1 / MulG1 (bls/bls_kilic.go:41-53).  Representation is byte-identical to the Kilic memory
// images the Go API hands over: little-endian limbs, Montgomery form with R = 2^256
// (F_r, 8 x u32) and R = 2^384 (F_p, 12 x u32); a u64-limbed Go value reinterpreted as u32 pairs.
//
// CDNA4 notes: 32-bit VALU, v_mad_u64_u32 is the widest multiplier, so limbs are 32 bit and every
// partial product is one v_mad_u64_u32 (a*b + 64-bit addend never overflows: (2^32-1)^2 + 2(2^32-1) < 2^64).
// Everything is plain C++ so the same source is compiled for the host by the unit tests (tests/host).
#pragma once
#include <stdint.h>
// branch weights for paths that exist for completeness only (P == +-Q in a table walk, an infinite operand of a butterfly): the compiler
// moves them out of the straight line of the hot loops
#define KZG_LIKELY(x) __builtin_expect(!!(x), 1)
#define KZG_UNLIKELY(x) __builtin_expect(!!(x), 0)

#if defined(__HIPCC__)
#define KZG_HD __host__ __device__ __forceinline__
#define KZG_HD_NOINLINE __host__ __device__ __noinline__
#else
#define KZG_HD inline __attribute__((always_inline))
#define KZG_HD_NOINLINE __attribute__((noinline))
#endif

namespace kzg {

// ---------------------------------------------------------------------------------------------
// field parameter packs.  mod(i) is written so that, after unrolling, every use folds to a literal.
// ---------------------------------------------------------------------------------------------
struct FpP {   // F_p, p = 0x1a0111ea...aaab (381 bit)
    // DEVICE-INTERNAL Montgomery radix is R' = 2^390 (13 limbs of 30 bits, see mont_mul_fp30 below), NOT Kilic's
    // 2^384: every G1 buffer that lives on the device is in the R' domain; the C ABI converts at the boundary
    // (fp_from_kilic / fp_to_kilic).  Storage stays 12 x u32 saturated, values canonical (< p).
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;     // -p^-1 mod 2^32 (generic CIOS, unused for F_p on the device)
    static constexpr uint32_t INV30 = 0x3ffcfffdu;   // -p^-1 mod 2^30
    static constexpr int N30 = 13, BITS = 381, GCD_ROUNDS = 26;   // inv(): 26 x 30 >= 2 * 381 - 1 binary-GCD steps
    KZG_HD static uint32_t mod(int i) {
        const uint32_t t[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    KZG_HD static uint32_t p30(int i) {   // p in 13 limbs of 30 bits
        const uint32_t t[13] = {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x034a83dau, 0x112bf673u,
                                0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x001a0111u};
        return t[i];
    }
    KZG_HD static uint32_t one(int i) {   // R' mod p = 2^390 mod p
        const uint32_t t[12] = {0x00d1ff2eu, 0x46760000u, 0x9b4800acu, 0x84b80337u, 0xe882431cu, 0x0dd9a7e0u,
                                0xb683dcf8u, 0xc26c26d0u, 0x63c4a5eeu, 0x29f14576u, 0x7f3e804bu, 0x015de996u};
        return t[i];
    }
    KZG_HD static uint32_t r2(int i) {    // R'^2 mod p
        const uint32_t t[12] = {0x4510070fu, 0xaec641c3u, 0xa0132243u, 0x6ea66ec3u, 0x1df507afu, 0x5efee07bu,
                                0xeed21b14u, 0x41442921u, 0x2d32f70au, 0x97900177u, 0x4acd918cu, 0x0f696ee0u};
        return t[i];
    }
    KZG_HD static uint32_t kilic_in(int i) {   // 2^396 mod p: mul(x_kilic, .) = x * 2^6 = R'-domain image
        const uint32_t t[12] = {0x3480cb7fu, 0x6f830000u, 0xbe042b12u, 0xd1fccdeau, 0x3c7de4b4u, 0x40d78057u,
                                0xc66805c5u, 0x6da3d19eu, 0x2746752au, 0x9afe6676u, 0x23205efbu, 0x09772fe1u};
        return t[i];
    }
    KZG_HD static uint32_t kilic_one(int i) {  // 2^384 mod p: Kilic's Montgomery one; mul(x', .) = x' / 2^6 = Kilic image
        const uint32_t t[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
};
struct FrP {   // F_r, r = 0x73eda753...00000001 (255 bit), bls/globals.go:9
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;   // -r^-1 mod 2^32
    static constexpr uint32_t INV30 = 0x3fffffffu; // -r^-1 mod 2^30
    static constexpr int N30 = 9, BITS = 255, GCD_ROUNDS = 18;    // inv(): 18 x 30 >= 2 * 255 - 1 binary-GCD steps
    KZG_HD static uint32_t p30(int i) {            // r in 9 limbs of 30 bits
        const uint32_t t[9] = {0x00000001u, 0x3ffffffcu, 0x3fe5bfefu, 0x2f6900bfu, 0x21d80553u, 0x27602026u, 0x17d48333u, 0x29d4ca67u, 0x000073edu};
        return t[i];
    }
    KZG_HD static uint32_t mod(int i) {
        const uint32_t t[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return t[i];
    }
    KZG_HD static uint32_t one(int i) {
        const uint32_t t[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return t[i];
    }
    KZG_HD static uint32_t r2(int i) {
        const uint32_t t[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return t[i];
    }
};

template <class F> struct alignas(16) felem { uint32_t l[F::N]; };
using fp = felem<FpP>;
using fr = felem<FrP>;

// ---------------------------------------------------------------------------------------------
// limb helpers
// ---------------------------------------------------------------------------------------------
#if defined(__clang__)
KZG_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t &c) {   // a + b + c, carry out in c  (v_addc_co_u32 chain)
    unsigned co; uint32_t r = __builtin_addc(a, b, c, &co); c = co; return r;
}
KZG_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t &br) {  // a - b - br, borrow out in br  (v_subb_co_u32 chain)
    unsigned bo; uint32_t r = __builtin_subc(a, b, br, &bo); br = bo; return r;
}
#else
KZG_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t &c) {
    uint64_t x = (uint64_t)a + b + c; c = (uint32_t)(x >> 32); return (uint32_t)x;
}
KZG_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t &br) {
    uint64_t x = (uint64_t)a - b - br; br = (uint32_t)(x >> 63); return (uint32_t)x;
}
#endif

template <class F> KZG_HD bool is_zero(const felem<F> &a) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) v |= a.l[i];
    return v == 0;
}
template <class F> KZG_HD bool equal(const felem<F> &a, const felem<F> &b) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) v |= a.l[i] ^ b.l[i];
    return v == 0;
}
template <class F> KZG_HD felem<F> zero() {
    felem<F> o;
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = 0;
    return o;
}
template <class F> KZG_HD felem<F> one() {
    felem<F> o;
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = F::one(i);
    return o;
}
// o = (t >= p) ? t - p : t, for t < 2p
template <class F> KZG_HD void reduce_once(felem<F> &o, const uint32_t *t) {
    uint32_t d[F::N]; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) d[i] = subb(t[i], F::mod(i), br);
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = br ? t[i] : d[i];
}
template <class F> KZG_HD felem<F> add(const felem<F> &a, const felem<F> &b) {
    uint32_t t[F::N]; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) t[i] = addc(a.l[i], b.l[i], c);
    felem<F> o; reduce_once<F>(o, t);   // both moduli leave >= 1 spare bit, so no carry out of the top limb
    return o;
}
template <class F> KZG_HD felem<F> sub(const felem<F> &a, const felem<F> &b) {
    felem<F> o; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = subb(a.l[i], b.l[i], br);
    uint32_t mask = 0u - br, c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = addc(o.l[i], F::mod(i) & mask, c);
    return o;
}
template <class F> KZG_HD felem<F> neg(const felem<F> &a) {
    felem<F> o; uint32_t br = 0;
    bool z = is_zero<F>(a);
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = subb(F::mod(i), a.l[i], br);
#pragma unroll
    for (int i = 0; i < F::N; i++) o.l[i] = z ? 0u : o.l[i];
    return o;
}
template <class F> KZG_HD felem<F> dbl(const felem<F> &a) { return add<F>(a, a); }

// ---------------------------------------------------------------------------------------------
// Montgomery product (CIOS), o = a * b / R mod p.   One v_mad_u64_u32 per partial product.
// ---------------------------------------------------------------------------------------------
template <class F> KZG_HD felem<F> mont_mul_inl(const felem<F> &a, const felem<F> &b) {
    constexpr int N = F::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)x; c = x >> 32;
        }
        uint64_t x = (uint64_t)t[N] + c;
        t[N] = (uint32_t)x; t[N + 1] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * F::INV;
        x = (uint64_t)m * F::mod(0) + t[0]; c = x >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            x = (uint64_t)m * F::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)x; c = x >> 32;
        }
        x = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)x; t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    felem<F> o; reduce_once<F>(o, t);   // t < 2p < 2^(32N): t[N] == 0
    return o;
}

// ---------------------------------------------------------------------------------------------
// F_p Montgomery product with UNSATURATED 30-bit limbs (R' = 2^390).
// Measured on gfx950 (tools/microbench.hip): v_mad_u64_u32, v_addc_co_u32 and v_lshl_add_u64 all issue at half rate,
// so in a saturated 32-bit CIOS the carry handling costs as much as the 288 multiplies themselves.  With 13 limbs of
// 30 bits every partial product (< 2^60) is accumulated by ONE v_mad_u64_u32 into a 64-bit column and columns never
// carry into each other inside a round: 338 mads + ~130 cheap ops instead of 288 mads + ~900 carry/move ops.
// A column holds at most 14 products between sweeps (14 * 2^60 < 2^64): one carry sweep after round 7 keeps it exact.
// Inputs and output are canonical (< p) in 12 x u32 storage.
// ---------------------------------------------------------------------------------------------
KZG_HD void unpack30(uint32_t *o, const fp &a) {
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int w = (30 * k) >> 5, sh = (30 * k) & 31;
        uint64_t v = a.l[w];
        if (w + 1 < 12) v |= (uint64_t)a.l[w + 1] << 32;
        o[k] = (uint32_t)(v >> sh) & 0x3fffffffu;
    }
}
// core: r = A * B / 2^390 mod p on 13 x 30-bit limbs (limbs < 2^30), r normalised (limbs < 2^30), value < A B / 2^390 + p
#ifdef KZG_COUNT_OPS
static uint64_t g_count_mul = 0, g_count_sqr = 0;          // host-only instrumentation (tests/host): products per group operation
#define KZG_COUNT(x) (++(x))
#else
#define KZG_COUNT(x)
#endif
KZG_HD void mont_core30(uint32_t *r, const uint32_t *A, const uint32_t *B) {
    KZG_COUNT(g_count_mul);
    uint64_t acc[14];
#pragma unroll
    for (int j = 0; j < 14; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] += (uint64_t)A[j] * B[i];
        uint32_t m = ((uint32_t)acc[0] * FpP::INV30) & 0x3fffffffu;
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] += (uint64_t)m * FpP::p30(j);
        acc[1] += acc[0] >> 30;                       // low 30 bits of acc[0] are zero by construction of m
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] = acc[j + 1];
        acc[13] = 0;
        if (i == 6) {                                 // carry sweep: keeps every column below 2^64
#pragma unroll
            for (int j = 0; j < 12; j++) { acc[j + 1] += acc[j] >> 30; acc[j] &= 0x3fffffffull; }
        }
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) { uint64_t x = acc[j] + c; r[j] = (uint32_t)x & 0x3fffffffu; c = x >> 30; }
}
// (A B + C D) / 2^390 mod p with ONE reduction: r normalised, value < (A B + C D) / 2^390 + p.  507 multiplies instead of 676
// for two products.  Every round adds three products per column, so the columns are swept every fourth round.
KZG_HD void mont_core30_dot2(uint32_t *r, const uint32_t *A, const uint32_t *B, const uint32_t *C, const uint32_t *D) {
    KZG_COUNT(g_count_mul);
    uint64_t acc[14];
#pragma unroll
    for (int j = 0; j < 14; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] += (uint64_t)A[j] * B[i];
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] += (uint64_t)C[j] * D[i];
        uint32_t m = ((uint32_t)acc[0] * FpP::INV30) & 0x3fffffffu;
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] += (uint64_t)m * FpP::p30(j);
        acc[1] += acc[0] >> 30;
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] = acc[j + 1];
        acc[13] = 0;
        if ((i & 3) == 3) {                           // <= 12 products of 2^60 (+ one swept carry) per column between sweeps
#pragma unroll
            for (int j = 0; j < 12; j++) { acc[j + 1] += acc[j] >> 30; acc[j] &= 0x3fffffffull; }
        }
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) { uint64_t x = acc[j] + c; r[j] = (uint32_t)x & 0x3fffffffu; c = x >> 30; }
}
// squaring: the 78 cross products are formed once with a doubled operand (2 A[j] < 2^31, product < 2^61; a column holds at
// most 6 of them + one square: < 2^64), all 26 columns are swept, then the 13 reduction rounds add at most 13 products of
// 2^60 per column.  91 + 169 = 260 multiplies instead of 338.
KZG_HD void mont_sqr_core30(uint32_t *r, const uint32_t *A) {
    KZG_COUNT(g_count_sqr);
    uint64_t T[26];
    uint32_t A2[13];
#pragma unroll
    for (int j = 0; j < 26; j++) T[j] = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) A2[j] = A[j] << 1;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        T[2 * i] += (uint64_t)A[i] * A[i];
#pragma unroll
        for (int j = i + 1; j < 13; j++) T[i + j] += (uint64_t)A2[j] * A[i];
    }
#pragma unroll
    for (int c = 0; c < 25; c++) { T[c + 1] += T[c] >> 30; T[c] &= 0x3fffffffull; }
#pragma unroll
    for (int i = 0; i < 13; i++) {
        uint32_t m = ((uint32_t)T[i] * FpP::INV30) & 0x3fffffffu;
#pragma unroll
        for (int j = 0; j < 13; j++) T[i + j] += (uint64_t)m * FpP::p30(j);
        T[i + 1] += T[i] >> 30;
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) { uint64_t x = T[13 + j] + c; r[j] = (uint32_t)x & 0x3fffffffu; c = x >> 30; }
}
KZG_HD void pack30(uint32_t *t, const uint32_t *r) {   // 13 normalised limbs (value < 2^384) -> 12 words
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int k = (32 * w) / 30, o = (32 * w) % 30;
        uint64_t v = (uint64_t)r[k] >> o;
        v |= (uint64_t)r[k + 1] << (30 - o);
        if (k + 2 < 13) v |= (uint64_t)r[k + 2] << (60 - o);
        t[w] = (uint32_t)v;
    }
}
KZG_HD fp mont_mul_fp30(const fp &a, const fp &b) {
    uint32_t A[13], B[13], r[13];
    unpack30(A, a); unpack30(B, b);
    mont_core30(r, A, B);
    // value = a b / R' + (multiple of p) < 2p < 2^382: fits 12 words; repack and subtract p once if needed
    uint32_t t[12];
    pack30(t, r);
    fp out; reduce_once<FpP>(out, t);
    return out;
}

KZG_HD fp mont_sqr_fp30(const fp &a) {
    uint32_t A[13], r[13];
    unpack30(A, a);
    mont_sqr_core30(r, A);
    uint32_t t[12];
    pack30(t, r);
    fp out; reduce_once<FpP>(out, t);
    return out;
}

// ---------------------------------------------------------------------------------------------
// fq: UNPACKED, LAZILY REDUCED F_p element for hot loops (the fixed-base table walks).  13 limbs of 30 bits,
// always normalised (limbs 0..11 < 2^30), value only bounded: v < B p with B tracked by hand at every use
// (see g1x_madd_fast in g1.hpp).  Rules:
//   mulq(a, b): needs Ba * Bb <= 600 (a b < 2^390 p), result B = 2.     No pack / unpack / final subtraction.
//   addq(a, b): limb-wise add + carry sweep, B = Ba + Bb.
//   subq<M>(a, b): a + M p - b with M p spread so that no limb goes negative (needs M >= Bb + 1), B = Ba + M.
// ---------------------------------------------------------------------------------------------
struct fq { uint32_t l[13]; };
KZG_HD fq unpackq(const fp &a) { fq o; unpack30(o.l, a); return o; }
KZG_HD void sweepq(fq &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { uint32_t t = a.l[i] + c; a.l[i] = t & 0x3fffffffu; c = t >> 30; }
    a.l[12] += c;
}
#if defined(KZG_MULQ_NOINLINE) && defined(__clang__)
// out-of-line form for kernels with many call sites (the G1 FFT stage has ~65: inlined they are ~240 KB of code against a
// 64 KB instruction cache); operands travel as 13-wide vectors in VGPRs
typedef uint32_t u32x13 __attribute__((ext_vector_type(13)));
KZG_HD_NOINLINE static u32x13 mulq_call(u32x13 av, u32x13 bv) {
    uint32_t A[13], B[13], r[13];
#pragma unroll
    for (int i = 0; i < 13; i++) { A[i] = av[i]; B[i] = bv[i]; }
    mont_core30(r, A, B);
    u32x13 o;
#pragma unroll
    for (int i = 0; i < 13; i++) o[i] = r[i];
    return o;
}
KZG_HD fq mulq(const fq &a, const fq &b) {
    u32x13 av, bv;
#pragma unroll
    for (int i = 0; i < 13; i++) { av[i] = a.l[i]; bv[i] = b.l[i]; }
    u32x13 r = mulq_call(av, bv);
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = r[i];
    return o;
}
KZG_HD_NOINLINE static u32x13 sqrq_call(u32x13 av) {
    uint32_t A[13], r[13];
#pragma unroll
    for (int i = 0; i < 13; i++) A[i] = av[i];
    mont_sqr_core30(r, A);
    u32x13 o;
#pragma unroll
    for (int i = 0; i < 13; i++) o[i] = r[i];
    return o;
}
KZG_HD fq sqrq(const fq &a) {
    u32x13 av;
#pragma unroll
    for (int i = 0; i < 13; i++) av[i] = a.l[i];
    u32x13 r = sqrq_call(av);
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = r[i];
    return o;
}
#else
KZG_HD fq mulq(const fq &a, const fq &b) { fq o; mont_core30(o.l, a.l, b.l); return o; }
KZG_HD fq sqrq(const fq &a) { fq o; mont_sqr_core30(o.l, a.l); return o; }
#endif
// always-inline forms, for the one loop of a call-based kernel that is worth the code size (the doubling loop of the GLV
// multiplication: the call ABI's 39 register moves per product are 8 % of its instructions)
KZG_HD fq mulq_inl(const fq &a, const fq &b) { fq o; mont_core30(o.l, a.l, b.l); return o; }
KZG_HD fq sqrq_inl(const fq &a) { fq o; mont_sqr_core30(o.l, a.l); return o; }
// a b + c d with one reduction: needs Ba Bb + Bc Bd <= 600, result B = 2
KZG_HD fq dot2q_inl(const fq &a, const fq &b, const fq &c, const fq &d) { fq o; mont_core30_dot2(o.l, a.l, b.l, c.l, d.l); return o; }
KZG_HD fq addq(const fq &a, const fq &b) {
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = a.l[i] + b.l[i];
    sweepq(o);
    return o;
}
// limb i of M * p in 30-bit limbs, spread so that every limb but the top is >= 2^30 - 1 (sum of the limbs is still M p)
template <int M> KZG_HD uint32_t mp_spread(int i) {
    uint64_t c = 0; uint32_t v = 0;
#pragma unroll
    for (int k = 0; k <= 12; k++) {
        uint64_t t = (uint64_t)FpP::p30(k) * (uint32_t)M + c;
        uint32_t limb = (k < 12) ? (uint32_t)(t & 0x3fffffffu) : (uint32_t)t;
        c = (k < 12) ? (t >> 30) : 0;
        if (k == i) v = limb;
    }
    if (i == 0) return v + (1u << 30);
    if (i < 12) return v + (1u << 30) - 1u;
    return v - 1u;
}
template <int M> KZG_HD fq subq(const fq &a, const fq &b) {
    fq o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = a.l[i] + mp_spread<M>(i) - b.l[i];
    sweepq(o);
    return o;
}
// v ≡ 0 (mod p) for a normalised v < 2p, i.e. v in {0, p}
KZG_HD bool is_zero_mod_p_q(const fq &a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { z |= a.l[i]; e |= a.l[i] ^ FpP::p30(i); }
    return z == 0 || e == 0;
}
// any bound B <= 600 -> canonical packed value
KZG_HD fp packq(const fq &a) {
    fq one_;
#pragma unroll
    for (int i = 0; i < 13; i++) one_.l[i] = 0;
    fp o1 = one<FpP>(); unpack30(one_.l, o1);
    fq r = mulq(a, one_);            // a * R' / R' = a, now < 2p
    uint32_t t[12];
    pack30(t, r.l);
    fp out; reduce_once<FpP>(out, t);
    return out;
}

// ---------------------------------------------------------------------------------------------
// F_r Montgomery product on 9 unsaturated 30-bit limbs, SAME radix as Kilic's images (R = 2^256): eight reduction rounds
// of 30 bits and a last one of 16 bits (8 * 30 + 16 = 256), so Fr arrays stay byte-identical to the Go slices with no
// conversion anywhere.  162 multiplies + ~180 cheap instructions against ~430 for the saturated 32-bit CIOS
// (mont_mul_inl), where two of every three instructions were carry handling.  A column holds at most 10 products of
// 2^60 between sweeps (one sweep after round 4).  Inputs canonical (< r), output canonical.
// ---------------------------------------------------------------------------------------------
KZG_HD void unpack30_fr(uint32_t *o, const fr &a) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int w = (30 * k) >> 5, sh = (30 * k) & 31;
        uint64_t v = a.l[w];
        if (w + 1 < 8) v |= (uint64_t)a.l[w + 1] << 32;
        o[k] = (uint32_t)(v >> sh) & 0x3fffffffu;
    }
}
KZG_HD fr mont_mul_fr30(const fr &a, const fr &b) {
    uint32_t A[9], B[9];
    unpack30_fr(A, a); unpack30_fr(B, b);
    uint64_t acc[10];
#pragma unroll
    for (int j = 0; j < 10; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] += (uint64_t)A[j] * B[i];
        if (i < 8) {
            uint32_t m = ((uint32_t)acc[0] * FrP::INV30) & 0x3fffffffu;
#pragma unroll
            for (int j = 0; j < 9; j++) acc[j] += (uint64_t)m * FrP::p30(j);
            acc[1] += acc[0] >> 30;                       // low 30 bits are zero by construction of m
#pragma unroll
            for (int j = 0; j < 9; j++) acc[j] = acc[j + 1];
            acc[9] = 0;
            if (i == 3) {
#pragma unroll
                for (int j = 0; j < 8; j++) { acc[j + 1] += acc[j] >> 30; acc[j] &= 0x3fffffffull; }
            }
        }
    }
    // last 16 bits of the radix: m16 r clears the low 16 bits (r = 1 mod 2^16, so -r^-1 = 0xffff)
    uint32_t m16 = (0u - (uint32_t)acc[0]) & 0xffffu;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[j] += (uint64_t)m16 * FrP::p30(j);
    // normalise, then shift right by 16 while repacking into 8 x 32-bit words: value < 2 r < 2^256
    uint32_t L[10]; uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { uint64_t x = acc[j] + c; L[j] = (uint32_t)x & 0x3fffffffu; c = x >> 30; }
    L[9] = (uint32_t)c;
    uint32_t t[9];
#pragma unroll
    for (int w = 0; w < 9; w++) {
        const int bit = 32 * w + 16, k = bit / 30, o = bit % 30;
        uint64_t v = 0;
        if (k < 10) v = (uint64_t)L[k] >> o;
        if (k + 1 < 10) v |= (uint64_t)L[k + 1] << (30 - o);
        if (k + 2 < 10) v |= (uint64_t)L[k + 2] << (60 - o);
        t[w] = (uint32_t)v;
    }
    fr out; reduce_once<FrP>(out, t);
    return out;
}

#if defined(KZG_FP_MUL_NOINLINE) && defined(__clang__)
// Out-of-line F_p product: keeps a Jacobian add at ~2 KB of code instead of ~100 KB (the I-cache is 64 KB).
// Operands and result travel as 12-wide vectors so the AMDGPU calling convention keeps all 24 + 12 dwords in
// VGPRs; passing the `fp` structs by value spilled the second operand through scratch on every call.
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));
KZG_HD_NOINLINE static u32x12 fp_mul_call(u32x12 av, u32x12 bv) {
    fp a, b;
#pragma unroll
    for (int i = 0; i < 12; i++) { a.l[i] = av[i]; b.l[i] = bv[i]; }
    fp o = mont_mul_fp30(a, b);
    u32x12 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = o.l[i];
    return r;
}
KZG_HD_NOINLINE static u32x12 fp_sqr_call(u32x12 av) {
    fp a;
#pragma unroll
    for (int i = 0; i < 12; i++) a.l[i] = av[i];
    fp o = mont_sqr_fp30(a);
    u32x12 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = o.l[i];
    return r;
}
KZG_HD fp sqr(const fp &a) {
    u32x12 av;
#pragma unroll
    for (int i = 0; i < 12; i++) av[i] = a.l[i];
    u32x12 r = fp_sqr_call(av);
    fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = r[i];
    return o;
}
KZG_HD fp mul(const fp &a, const fp &b) {
    u32x12 av, bv;
#pragma unroll
    for (int i = 0; i < 12; i++) { av[i] = a.l[i]; bv[i] = b.l[i]; }
    u32x12 r = fp_mul_call(av, bv);
    fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = r[i];
    return o;
}
#else
KZG_HD fp mul(const fp &a, const fp &b) { return mont_mul_fp30(a, b); }
KZG_HD fp sqr(const fp &a) { return mont_sqr_fp30(a); }
#endif
KZG_HD fr mul(const fr &a, const fr &b) { return mont_mul_fr30(a, b); }
KZG_HD fr sqr(const fr &a) { return mul(a, a); }
KZG_HD fp add(const fp &a, const fp &b) { return add<FpP>(a, b); }
KZG_HD fp sub(const fp &a, const fp &b) { return sub<FpP>(a, b); }
KZG_HD fr add(const fr &a, const fr &b) { return add<FrP>(a, b); }
KZG_HD fr sub(const fr &a, const fr &b) { return sub<FrP>(a, b); }

template <class F> KZG_HD felem<F> from_mont(const felem<F> &a) {   // Kilic FromRed()
    felem<F> o1 = zero<F>(); o1.l[0] = 1;
    return mul(a, o1);
}
template <class F> KZG_HD felem<F> to_mont(const felem<F> &a) {
    felem<F> r2;
#pragma unroll
    for (int i = 0; i < F::N; i++) r2.l[i] = F::r2(i);
    return mul(a, r2);
}
// a^(mod-2): inversion by Fermat (0 -> 0).  Plain square-and-multiply over the bits of the modulus.  Kept as the check of
// inv() in tests/host; the device paths use the binary GCD below (a Fermat chain is ~570 dependent products: 0.7 ms of pure
// latency at the end of every commitment batch).
template <class F> KZG_HD felem<F> inv_fermat(const felem<F> &a) {
    uint32_t ex[F::N]; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) ex[i] = subb(F::mod(i), i == 0 ? 2u : 0u, br);   // r's low limb is 1: borrow
    felem<F> acc = one<F>();
    for (int i = F::N - 1; i >= 0; i--) {
        uint32_t e = ex[i];
        for (int b = 31; b >= 0; b--) {
            acc = sqr(acc);
            if ((e >> b) & 1u) acc = mul(acc, a);
        }
    }
    return acc;
}

// Hides what the compiler knows about the range of a 32-bit value.  The limbs of inv()'s u, v are masked to 30 bits, so clang turns
// (int64)f * (int64)u[j] into an UNSIGNED 32 x 32 product plus a sign correction (two v_mad_u64_u32 and two moves per product);
// with the range hidden every product is one v_mad_i64_i32: the round's matrix application 952 -> 647 instructions.  (With the
// select-based steps below a round is 1 457 instead of 1 882 instructions.  Measured effect on the latency of one inversion on one lane
// of an otherwise idle wavefront: none, 100-110 us before and after -- that path waits on dependent results, not on issue.)
KZG_HD int32_t opaque_i32(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}
// Montgomery-domain inverse (x R -> x^-1 R, 0 -> 0) by the binary GCD with 64-bit approximations (Pornin, "Optimized Binary GCD
// for Modular Inversion", ePrint 2020/972), restated for 30-bit limbs:
//   a = y, b = p, u = R^2 mod p, v = 0, invariants a = u y / R^2, b = v y / R^2 (mod p);  every round runs 30 steps of
//   "a odd: (a < b ? swap), a -= b;  a /= 2" on approximations (top 34 bits | low 30 bits) of a and b, collecting the
//   step matrix (f0 g0; f1 g1), |f|+|g| <= 2^30, then applies it exactly:  (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30
//   (negating a row whose result is negative) and (u, v) <- the same combination / 2^30 mod p (one Montgomery step, so the
//   invariant keeps no stray power of two).  2 BITS - 1 steps reach a = 0, b = 1, i.e. v = y^-1 R^2 = x^-1 R.
// Branch-free and lane-uniform: ~30 k VALU instructions for F_p against ~250 k for the Fermat chain.
template <class F> KZG_HD felem<F> inv(const felem<F> &x) {
    constexpr int L = F::N30;
    constexpr uint32_t MASK = 0x3fffffffu;
    uint32_t a[L], b[L];
    int32_t u[L], v[L];                                   // limbs 0..L-2 in [0, 2^30), top limb signed; |u|, |v| < 2^BITS
#pragma unroll
    for (int k = 0; k < L; k++) {
        const int w = (30 * k) >> 5, sh = (30 * k) & 31;
        uint64_t xv = w < F::N ? x.l[w] : 0u, rv = w < F::N ? F::r2(w) : 0u;
        if (w + 1 < F::N) { xv |= (uint64_t)x.l[w + 1] << 32; rv |= (uint64_t)F::r2(w + 1) << 32; }
        a[k] = (uint32_t)(xv >> sh) & MASK;
        u[k] = (int32_t)((uint32_t)(rv >> sh) & MASK);
        b[k] = F::p30(k);
        v[k] = 0;
    }
#pragma nounroll
    for (int round = 0; round < F::GCD_ROUNDS; round++) {
        // window (limb j, j-1, j-2) at the highest limb j >= 2 where a or b is non-zero
        uint32_t ah = a[L - 1], am = a[L - 2], al = a[L - 3], bh = b[L - 1], bm = b[L - 2], bl = b[L - 3], upper = 0;
#pragma unroll
        for (int i = L - 4; i >= 0; i--) {
            upper |= a[i + 3] | b[i + 3];
            const bool z = (ah | bh) == 0;
            ah = z ? am : ah; am = z ? al : am; al = z ? a[i] : al;
            bh = z ? bm : bh; bm = z ? bl : bm; bl = z ? b[i] : bl;
        }
        const uint64_t A2 = (uint64_t)ah << 30 | am, B2 = (uint64_t)bh << 30 | bm;
        const int s = __builtin_clzll(A2 | B2 | 1ull);    // >= 4
        const uint64_t XA = (A2 << s) | (s <= 30 ? (uint64_t)(al >> (30 - s)) : (uint64_t)al << (s - 30));
        const uint64_t XB = (B2 << s) | (s <= 30 ? (uint64_t)(bl >> (30 - s)) : (uint64_t)bl << (s - 30));
        const bool exact = upper == 0 && s >= 30;         // both below 2^64: the window is (2, 1, 0) and holds the exact values
        uint64_t xa = exact ? ((uint64_t)a[0] | (uint64_t)a[1] << 30 | (uint64_t)a[2] << 60) : ((XA >> 30) << 30 | a[0]);
        uint64_t xb = exact ? ((uint64_t)b[0] | (uint64_t)b[1] << 30 | (uint64_t)b[2] << 60) : ((XB >> 30) << 30 | b[0]);
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma nounroll
        for (int i = 0; i < 30; i++) {
            // selects on lane conditions (v_cndmask on an SGPR-pair mask) instead of xor-masks: 22 instead of 28 instructions per step
            const bool odd = (xa & 1ull) != 0, sw = odd && xa < xb;
            const uint64_t sa = sw ? xb : xa, sb = sw ? xa : xb;
            const int32_t sf0 = sw ? f1 : f0, sf1 = sw ? f0 : f1, sg0 = sw ? g1 : g0, sg1 = sw ? g0 : g1;
            xa = (odd ? sa - sb : sa) >> 1; xb = sb;
            f0 = odd ? sf0 - sf1 : sf0; g0 = odd ? sg0 - sg1 : sg0;
            f1 = sf1 << 1; g1 = sg1 << 1;
        }
        // (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30, exact division; a negative row is negated (with its factors)
        int64_t ca = 0, cb = 0;
        uint32_t na[L], nb[L];
#pragma unroll
        for (int j = 0; j < L; j++) {
            // (every limb of a and b is below 2^31: as int32 operands each product is ONE v_mad_i64_i32, the carry its addend)
            const int64_t ta = (ca + (int64_t)f0 * (int64_t)(int32_t)a[j]) + (int64_t)g0 * (int64_t)(int32_t)b[j];
            const int64_t tb = (cb + (int64_t)f1 * (int64_t)(int32_t)a[j]) + (int64_t)g1 * (int64_t)(int32_t)b[j];
            if (j > 0) { na[j - 1] = (uint32_t)ta & MASK; nb[j - 1] = (uint32_t)tb & MASK; }
            ca = ta >> 30; cb = tb >> 30;
        }
        const uint32_t nga = (uint32_t)(ca >> 63), ngb = (uint32_t)(cb >> 63);    // all ones when negative
        na[L - 1] = (uint32_t)ca; nb[L - 1] = (uint32_t)cb;
        uint32_t c1 = nga & 1u, c2 = ngb & 1u;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const uint32_t sa = (na[j] ^ (j < L - 1 ? (nga & MASK) : nga)) + c1, sb = (nb[j] ^ (j < L - 1 ? (ngb & MASK) : ngb)) + c2;
            if (j < L - 1) { a[j] = sa & MASK; c1 = sa >> 30; b[j] = sb & MASK; c2 = sb >> 30; }
            else { a[j] = sa; b[j] = sb; }
        }
        f0 = (f0 ^ (int32_t)nga) - (int32_t)nga; g0 = (g0 ^ (int32_t)nga) - (int32_t)nga;
        f1 = (f1 ^ (int32_t)ngb) - (int32_t)ngb; g1 = (g1 ^ (int32_t)ngb) - (int32_t)ngb;
        // (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^30 mod p: add the multiple of p that clears the low limb, shift one limb
        const uint32_t mu = (((uint32_t)f0 * (uint32_t)u[0] + (uint32_t)g0 * (uint32_t)v[0]) * F::INV30) & MASK;
        const uint32_t mv = (((uint32_t)f1 * (uint32_t)u[0] + (uint32_t)g1 * (uint32_t)v[0]) * F::INV30) & MASK;
        int64_t cu = 0, cv = 0;
        int32_t nu[L], nv[L];
        const int32_t smu = opaque_i32((int32_t)mu), smv = opaque_i32((int32_t)mv);
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int32_t uj = opaque_i32(u[j]), vj = opaque_i32(v[j]);
            const int64_t tu = ((cu + (int64_t)f0 * (int64_t)uj) + (int64_t)g0 * (int64_t)vj) + (int64_t)smu * (int64_t)(int32_t)F::p30(j);
            const int64_t tv = ((cv + (int64_t)f1 * (int64_t)uj) + (int64_t)g1 * (int64_t)vj) + (int64_t)smv * (int64_t)(int32_t)F::p30(j);
            if (j > 0) { nu[j - 1] = (int32_t)((uint32_t)tu & MASK); nv[j - 1] = (int32_t)((uint32_t)tv & MASK); }
            cu = tu >> 30; cv = tv >> 30;
        }
        nu[L - 1] = (int32_t)cu; nv[L - 1] = (int32_t)cv;
        // results lie in (-2^BITS, 2^BITS + p): subtract p once when >= 2^BITS, keeping |u|, |v| < 2^BITS
        constexpr int32_t TOPBIT = 1 << (F::BITS - 30 * (L - 1));
        const uint32_t su = nu[L - 1] >= TOPBIT ? 0xffffffffu : 0u, sv = nv[L - 1] >= TOPBIT ? 0xffffffffu : 0u;
        int32_t bu = 0, bv = 0;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int32_t du = nu[j] - (int32_t)(F::p30(j) & su) + bu, dv = nv[j] - (int32_t)(F::p30(j) & sv) + bv;
            if (j < L - 1) { u[j] = du & (int32_t)MASK; bu = du >> 30; v[j] = dv & (int32_t)MASK; bv = dv >> 30; }
            else { u[j] = du; v[j] = dv; }
        }
    }
    // v in (-2^BITS, 2^BITS), 2^BITS < 2 p: add p while negative (at most twice), subtract p once if still >= p
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const uint32_t sel = v[L - 1] < 0 ? 0xffffffffu : 0u;
        int32_t c = 0;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int32_t d = v[j] + (int32_t)(F::p30(j) & sel) + c;
            if (j < L - 1) { v[j] = d & (int32_t)MASK; c = d >> 30; } else v[j] = d;
        }
    }
    int32_t d[L], c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const int32_t t = v[j] - (int32_t)F::p30(j) + c;
        if (j < L - 1) { d[j] = t & (int32_t)MASK; c = t >> 30; } else d[j] = t;
    }
    const bool ge = d[L - 1] >= 0;
    felem<F> out;
#pragma unroll
    for (int w = 0; w < F::N; w++) {
        const int k = (32 * w) / 30, o = (32 * w) % 30;
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (k + q < L) acc |= (uint64_t)(uint32_t)(ge ? d[k + q] : v[k + q]) << (30 * q);
        out.l[w] = (uint32_t)(acc >> o);
    }
    return out;
}
KZG_HD fr fr_from_u64(uint64_t v) {   // bls.AsFr (bls/bignum_kilic.go:61-65)
    fr t = zero<FrP>(); t.l[0] = (uint32_t)v; t.l[1] = (uint32_t)(v >> 32);
    return to_mont<FrP>(t);
}

// Kilic memory image (R = 2^384) <-> device-internal image (R' = 2^390) of an F_p element
KZG_HD fp fp_from_kilic(const fp &a) {
    fp c;
#pragma unroll
    for (int i = 0; i < 12; i++) c.l[i] = FpP::kilic_in(i);
    return mul(a, c);
}
KZG_HD fp fp_to_kilic(const fp &a) {
    fp c;
#pragma unroll
    for (int i = 0; i < 12; i++) c.l[i] = FpP::kilic_one(i);
    return mul(a, c);
}
KZG_HD fp fp_kilic_one() {
    fp c;
#pragma unroll
    for (int i = 0; i < 12; i++) c.l[i] = FpP::kilic_one(i);
    return c;
}

}  // namespace kzg
