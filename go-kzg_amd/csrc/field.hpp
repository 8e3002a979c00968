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

#if defined(__HIPCC__) || defined(__CUDACC__)
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
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;   // -p^-1 mod 2^32
    KZG_HD static uint32_t mod(int i) {
        const uint32_t t[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    KZG_HD static uint32_t one(int i) {   // R mod p
        const uint32_t t[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
    KZG_HD static uint32_t r2(int i) {    // R^2 mod p
        const uint32_t t[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return t[i];
    }
};
struct FrP {   // F_r, r = 0x73eda753...00000001 (255 bit), bls/globals.go:9
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;   // -r^-1 mod 2^32
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

#if defined(KZG_FP_MUL_NOINLINE) && defined(__clang__)
// Out-of-line F_p product: keeps a Jacobian add at ~2 KB of code instead of ~100 KB (the I-cache is 64 KB).
// Operands and result travel as 12-wide vectors so the AMDGPU calling convention keeps all 24 + 12 dwords in
// VGPRs; passing the `fp` structs by value spilled the second operand through scratch on every call.
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));
KZG_HD_NOINLINE static u32x12 fp_mul_call(u32x12 av, u32x12 bv) {
    fp a, b;
#pragma unroll
    for (int i = 0; i < 12; i++) { a.l[i] = av[i]; b.l[i] = bv[i]; }
    fp o = mont_mul_inl<FpP>(a, b);
    u32x12 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = o.l[i];
    return r;
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
KZG_HD fp mul(const fp &a, const fp &b) { return mont_mul_inl<FpP>(a, b); }
#endif
KZG_HD fr mul(const fr &a, const fr &b) { return mont_mul_inl<FrP>(a, b); }
KZG_HD fp sqr(const fp &a) { return mul(a, a); }
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
// a^(mod-2): inversion by Fermat (0 -> 0).  Plain square-and-multiply over the bits of the modulus.
template <class F> KZG_HD felem<F> inv(const felem<F> &a) {
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
KZG_HD fr fr_from_u64(uint64_t v) {   // bls.AsFr (bls/bignum_kilic.go:61-65)
    fr t = zero<FrP>(); t.l[0] = (uint32_t)v; t.l[1] = (uint32_t)(v >> 32);
    return to_mont<FrP>(t);
}

}  // namespace kzg
