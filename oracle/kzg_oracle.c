/*
 * kzg_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of go-kzg's commitment / proof hot
 * path (reference: protolambda/go-kzg @ /root/reference), used as the parity oracle for the HIP
 * library and as the `cpu_baseline` leg of bench.py.  Nothing in the product path (go-kzg_amd/,
 * include/) may include, link or call this file.
 *
 * Parity pin: the reference cannot be built here (no Go toolchain; its arithmetic dependency
 * github.com/kilic/bls12-381 v0.1.1-0.20220929213557-ca162e8a70f4 (go.mod:8) is not vendored), so
 * this file is a restatement.  It is pinned by the reference's own known-answer data:
 *   TestInvFFT (fft_fr_test.go:32-71), TestDASFFTExtension (das_extension_test.go:11-40),
 *   TestPointCompression (bls/bls_test.go:11-23), bls.Scale2RootOfUnity (bls/globals.go:27-60) and
 *   eth/trusted_setup.json (setup_G1 = [1337^i]G1, setup_G1_lagrange = FFTG1(setup_G1, inv)),
 * see tests/test_oracle_kat.py and tests/golden/.
 *
 * Memory images are the Kilic ones the Go API hands over:
 *   Fr  = 4 x u64 little-endian limbs, Montgomery form, R = 2^256 mod r   (bls/bignum_kilic.go:21-23)
 *   G1  = 3 x 6 x u64 (X, Y, Z) Jacobian, each coordinate Montgomery R = 2^384 mod p, inf <=> Z = 0
 *         (bls/bls_kilic.go:30-35)
 *
 * Control flow follows the reference file by file (each function cites the lines it restates);
 * the field / curve arithmetic underneath restates the published Kilic algorithms (Montgomery
 * CIOS, add-2007-bl / dbl-2009-l Jacobian formulas, MSB-first double-and-add MulScalar, bucket
 * MultiExp with window ceil(ln n)) -- [from memory; source not on disk].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct { u64 l[4]; } fr_t;
typedef struct { u64 l[6]; } fp_t;
typedef struct { fp_t x, y, z; } g1_t;

#define KO_OK 0
#define KO_ERR_TOO_WIDE 1      /* "got %d values but only have %d roots of unity"  fft_fr.go:57-59,78-80 */
#define KO_ERR_NOT_POW2 2      /* "got %d values but not a power of two"           fft_fr.go:81-83       */
#define KO_ERR_LEN_MISMATCH 3  /* panic sites: bls_kilic.go:133-135, fk20_single.go:60-62               */
#define KO_ERR_UPPER_HALF 4    /* "bad input, second half should be zeroed" fk20_single.go:150-154       */
#define KO_ERR_BAD_ARG 5
#define KO_ERR_BAD_POINT 6

/* ------------------------------------------------------------------------------------------------
 * constants (computed from p, r; checked in tests/test_oracle_kat.py against bls/globals.go)
 * ---------------------------------------------------------------------------------------------- */
static const fp_t FP_P   = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const fp_t FP_ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const fp_t FP_R2  = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
static const u64 FP_INV = 0x89f3fffcfffcfffdULL;
static const fp_t FP_B   = {{0xaa270000000cfff3ULL, 0x53cc0032fc34000aULL, 0x478fe97a6b0a807fULL, 0xb1d37ebee6ba24d7ULL, 0x8ec9733bbf78ab2fULL, 0x09d645513d83de7eULL}}; /* 4 (curve b) */
static const fp_t FP_HALF_PM1 = {{0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL, 0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL}}; /* (p-1)/2, standard form */
/* G1 generator (decimals in-tree at bls/bls_hbls.go:23-24), Montgomery form */
static const fp_t G1_GX  = {{0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL}};
static const fp_t G1_GY  = {{0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL}};

static const fr_t FR_R   = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}}; /* bls/globals.go:9 */
static const fr_t FR_ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const fr_t FR_R2  = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const u64 FR_INV = 0xfffffffeffffffffULL;

/* ------------------------------------------------------------------------------------------------
 * generic Montgomery arithmetic on N x u64 limbs
 * ---------------------------------------------------------------------------------------------- */
static inline int limbs_geq(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline u64 limbs_add(u64 *o, const u64 *a, const u64 *b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; i++) { u128 t = (u128)a[i] + b[i] + c; o[i] = (u64)t; c = (u64)(t >> 64); }
    return c;
}
static inline u64 limbs_sub(u64 *o, const u64 *a, const u64 *b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) { u128 t = (u128)a[i] - b[i] - br; o[i] = (u64)t; br = (u64)(t >> 64) & 1; }
    return br;
}
static inline int limbs_is_zero(const u64 *a, int n) { u64 v = 0; for (int i = 0; i < n; i++) v |= a[i]; return v == 0; }

static inline void mod_add(u64 *o, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 t[6]; u64 c = limbs_add(t, a, b, n);
    if (c || limbs_geq(t, m, n)) limbs_sub(o, t, m, n); else memcpy(o, t, 8 * n);
}
static inline void mod_sub(u64 *o, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 t[6]; u64 br = limbs_sub(t, a, b, n);
    if (br) limbs_add(o, t, m, n); else memcpy(o, t, 8 * n);
}
/* CIOS Montgomery product, o = a*b/R mod m */
static inline void mont_mul(u64 *o, const u64 *a, const u64 *b, const u64 *m, u64 inv, int n) {
    u64 t[8] = {0};
    for (int i = 0; i < n; i++) {
        u64 c = 0;
        for (int j = 0; j < n; j++) { u128 x = (u128)a[j] * b[i] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
        u128 x = (u128)t[n] + c; t[n] = (u64)x; t[n + 1] = (u64)(x >> 64);
        u64 q = t[0] * inv;
        x = (u128)q * m[0] + t[0]; c = (u64)(x >> 64);
        for (int j = 1; j < n; j++) { x = (u128)q * m[j] + t[j] + c; t[j - 1] = (u64)x; c = (u64)(x >> 64); }
        x = (u128)t[n] + c; t[n - 1] = (u64)x; t[n] = t[n + 1] + (u64)(x >> 64);
    }
    if (t[n] || limbs_geq(t, m, n)) limbs_sub(o, t, m, n); else memcpy(o, t, 8 * n);
}

/* ---- Fp ---- */
static inline void fp_mul(fp_t *o, const fp_t *a, const fp_t *b) { mont_mul(o->l, a->l, b->l, FP_P.l, FP_INV, 6); }
static inline void fp_sqr(fp_t *o, const fp_t *a) { mont_mul(o->l, a->l, a->l, FP_P.l, FP_INV, 6); }
static inline void fp_add(fp_t *o, const fp_t *a, const fp_t *b) { mod_add(o->l, a->l, b->l, FP_P.l, 6); }
static inline void fp_sub(fp_t *o, const fp_t *a, const fp_t *b) { mod_sub(o->l, a->l, b->l, FP_P.l, 6); }
static inline void fp_dbl(fp_t *o, const fp_t *a) { mod_add(o->l, a->l, a->l, FP_P.l, 6); }
static inline int fp_is_zero(const fp_t *a) { return limbs_is_zero(a->l, 6); }
static inline int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof(fp_t)) == 0; }
static inline void fp_neg(fp_t *o, const fp_t *a) { if (fp_is_zero(a)) *o = *a; else limbs_sub(o->l, FP_P.l, a->l, 6); }
static void fp_pow(fp_t *o, const fp_t *a, const u64 *e, int nlimbs) {
    fp_t acc = FP_ONE;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, a);
    }
    *o = acc;
}
static void fp_inv(fp_t *o, const fp_t *a) { /* Fermat: a^(p-2); a = 0 -> 0 */
    u64 e[6]; memcpy(e, FP_P.l, 48); e[0] -= 2;
    fp_pow(o, a, e, 6);
}
static void fp_from_mont(fp_t *o, const fp_t *a) { fp_t one = {{1, 0, 0, 0, 0, 0}}; fp_mul(o, a, &one); }
static void fp_to_mont(fp_t *o, const fp_t *a) { fp_mul(o, a, &FP_R2); }
/* sqrt for p = 3 mod 4: a^((p+1)/4); returns 1 if a is a square */
static int fp_sqrt(fp_t *o, const fp_t *a) {
    u64 e[6]; u64 one[6] = {1, 0, 0, 0, 0, 0};
    limbs_add(e, FP_P.l, one, 6);                       /* p + 1 (no overflow: p < 2^381) */
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
    fp_t s, chk; fp_pow(&s, a, e, 6); fp_sqr(&chk, &s);
    *o = s; return fp_eq(&chk, a);
}

/* ---- Fr ---- (bls/bignum_kilic.go:95-118: Add, Sub, RedMul, RedInverse on mont-red values) */
static inline void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { mont_mul(o->l, a->l, b->l, FR_R.l, FR_INV, 4); }
static inline void fr_add(fr_t *o, const fr_t *a, const fr_t *b) { mod_add(o->l, a->l, b->l, FR_R.l, 4); }
static inline void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { mod_sub(o->l, a->l, b->l, FR_R.l, 4); }
static inline int fr_is_zero(const fr_t *a) { return limbs_is_zero(a->l, 4); }
static inline int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof(fr_t)) == 0; }
static void fr_pow(fr_t *o, const fr_t *a, const u64 *e, int nlimbs) {
    fr_t acc = FR_ONE;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fr_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
    }
    *o = acc;
}
static void fr_inv(fr_t *o, const fr_t *a) { u64 e[4]; memcpy(e, FR_R.l, 32); e[0] -= 2; fr_pow(o, a, e, 4); }
static void fr_from_mont(fr_t *o, const fr_t *a) { fr_t one = {{1, 0, 0, 0}}; fr_mul(o, a, &one); }
static void fr_to_mont(fr_t *o, const fr_t *a) { fr_mul(o, a, &FR_R2); }
static void fr_from_u64(fr_t *o, u64 v) { fr_t t = {{v, 0, 0, 0}}; fr_to_mont(o, &t); }   /* bls.AsFr bignum_kilic.go:61-65 */

/* ------------------------------------------------------------------------------------------------
 * G1 (Jacobian, a = 0, b = 4)
 * ---------------------------------------------------------------------------------------------- */
static inline int g1_is_inf(const g1_t *p) { return fp_is_zero(&p->z); }
static inline void g1_set_inf(g1_t *p) { memset(p, 0, sizeof *p); p->y = FP_ONE; }   /* Kilic Zero(): (0, 1, 0) */

static void g1_dbl(g1_t *o, const g1_t *p) {      /* dbl-2009-l */
    if (g1_is_inf(p)) { g1_set_inf(o); return; }
    fp_t a, b, c, d, e, f, t, x3, y3, z3;
    fp_sqr(&a, &p->x); fp_sqr(&b, &p->y); fp_sqr(&c, &b);
    fp_add(&t, &p->x, &b); fp_sqr(&t, &t); fp_sub(&t, &t, &a); fp_sub(&t, &t, &c); fp_dbl(&d, &t);
    fp_dbl(&e, &a); fp_add(&e, &e, &a);
    fp_sqr(&f, &e);
    fp_dbl(&t, &d); fp_sub(&x3, &f, &t);
    fp_mul(&z3, &p->y, &p->z); fp_dbl(&z3, &z3);
    fp_sub(&t, &d, &x3); fp_mul(&y3, &e, &t);
    fp_dbl(&c, &c); fp_dbl(&c, &c); fp_dbl(&c, &c); fp_sub(&y3, &y3, &c);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1_add(g1_t *o, const g1_t *p, const g1_t *q) {   /* add-2007-bl with the exceptional cases */
    if (g1_is_inf(p)) { *o = *q; return; }
    if (g1_is_inf(q)) { *o = *p; return; }
    fp_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&u1, &p->x, &z2z2); fp_mul(&u2, &q->x, &z1z1);
    fp_mul(&s1, &p->y, &q->z); fp_mul(&s1, &s1, &z2z2);
    fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
    if (fp_eq(&u1, &u2)) {
        if (fp_eq(&s1, &s2)) { g1_dbl(o, p); return; }
        g1_set_inf(o); return;
    }
    fp_sub(&h, &u2, &u1);
    fp_dbl(&i, &h); fp_sqr(&i, &i);
    fp_mul(&j, &h, &i);
    fp_sub(&rr, &s2, &s1); fp_dbl(&rr, &rr);
    fp_mul(&v, &u1, &i);
    fp_sqr(&x3, &rr); fp_sub(&x3, &x3, &j); fp_sub(&x3, &x3, &v); fp_sub(&x3, &x3, &v);
    fp_sub(&t, &v, &x3); fp_mul(&y3, &rr, &t);
    fp_mul(&t, &s1, &j); fp_dbl(&t, &t); fp_sub(&y3, &y3, &t);
    fp_add(&z3, &p->z, &q->z); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &z1z1); fp_sub(&z3, &z3, &z2z2); fp_mul(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1_neg(g1_t *o, const g1_t *p) { o->x = p->x; o->z = p->z; fp_neg(&o->y, &p->y); }
static void g1_sub(g1_t *o, const g1_t *p, const g1_t *q) { g1_t n; g1_neg(&n, q); g1_add(o, p, &n); }
/* bls.MulG1 (bls/bls_kilic.go:41-45): scalar leaves Montgomery form (FromRed), then MSB-first double-and-add */
static void g1_mul(g1_t *o, const g1_t *p, const fr_t *k_mont) {
    fr_t k; fr_from_mont(&k, k_mont);
    g1_t acc; g1_set_inf(&acc);
    int top = 255;
    while (top >= 0 && !((k.l[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        g1_dbl(&acc, &acc);
        if ((k.l[i / 64] >> (i % 64)) & 1) g1_add(&acc, &acc, p);
    }
    *o = acc;
}
static void g1_affine(g1_t *o, const g1_t *p) {   /* Z -> 1 (Montgomery one); inf -> (0,1,0) */
    if (g1_is_inf(p)) { g1_set_inf(o); return; }
    fp_t zi, zi2, zi3; fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&o->x, &p->x, &zi2); fp_mul(&o->y, &p->y, &zi3); o->z = FP_ONE;
}
static int g1_equal(const g1_t *p, const g1_t *q) {   /* bls.EqualG1 bls_kilic.go:106-108 (projective equality) */
    int pi = g1_is_inf(p), qi = g1_is_inf(q);
    if (pi || qi) return pi && qi;
    fp_t z1z1, z2z2, a, b;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&a, &p->x, &z2z2); fp_mul(&b, &q->x, &z1z1);
    if (!fp_eq(&a, &b)) return 0;
    fp_mul(&a, &p->y, &q->z); fp_mul(&a, &a, &z2z2);
    fp_mul(&b, &q->y, &p->z); fp_mul(&b, &b, &z1z1);
    return fp_eq(&a, &b);
}

/* ------------------------------------------------------------------------------------------------
 * exported: scalar / point primitives  (the per-op bls.* functions the tests need)
 * ---------------------------------------------------------------------------------------------- */
#define API __attribute__((visibility("default")))

/* bls.ValidFr (bls/bignum_all.go:12-35) + FrFrom32 (bls/bignum_kilic.go:33-44): 32 LE bytes -> mont */
API int ko_fr_from_le32(fr_t *o, const uint8_t *b) {
    fr_t t; memcpy(&t, b, 32);
    if (limbs_geq(t.l, FR_R.l, 4)) return 0;
    fr_to_mont(o, &t); return 1;
}
API void ko_fr_to_le32(uint8_t *b, const fr_t *a) { fr_t t; fr_from_mont(&t, a); memcpy(b, &t, 32); }   /* FrTo32 :46-55 */
API void ko_fr_from_u64(fr_t *o, u64 v) { fr_from_u64(o, v); }
API void ko_fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { fr_mul(o, a, b); }
API void ko_fr_add(fr_t *o, const fr_t *a, const fr_t *b) { fr_add(o, a, b); }
API void ko_fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { fr_sub(o, a, b); }
API void ko_fr_inv(fr_t *o, const fr_t *a) { fr_inv(o, a); }
API void ko_g1_generator(g1_t *o) { o->x = G1_GX; o->y = G1_GY; o->z = FP_ONE; }
API void ko_g1_zero(g1_t *o) { g1_set_inf(o); }
API void ko_g1_add(g1_t *o, const g1_t *a, const g1_t *b) { g1_add(o, a, b); }
API void ko_g1_sub(g1_t *o, const g1_t *a, const g1_t *b) { g1_sub(o, a, b); }
API void ko_g1_dbl(g1_t *o, const g1_t *a) { g1_dbl(o, a); }
API void ko_g1_mul(g1_t *o, const g1_t *a, const fr_t *k) { g1_mul(o, a, k); }
API int ko_g1_equal(const g1_t *a, const g1_t *b) { return g1_equal(a, b); }
API void ko_g1_affine_batch(g1_t *o, const g1_t *a, u64 n) { for (u64 i = 0; i < n; i++) g1_affine(&o[i], &a[i]); }

/* ZCash 48-byte compressed form (bls.ToCompressedG1, bls/bls_kilic.go:114-116; format SURVEY App. A) */
API void ko_g1_to_compressed(uint8_t *out, const g1_t *p) {
    memset(out, 0, 48);
    if (g1_is_inf(p)) { out[0] = 0xc0; return; }
    g1_t a; g1_affine(&a, p);
    fp_t x, y; fp_from_mont(&x, &a.x); fp_from_mont(&y, &a.y);
    for (int i = 0; i < 48; i++) out[i] = (uint8_t)(x.l[(47 - i) / 8] >> (8 * ((47 - i) % 8)));
    out[0] |= 0x80;
    /* y > (p-1)/2 */
    int gt = 0;
    for (int i = 5; i >= 0; i--) { if (y.l[i] > FP_HALF_PM1.l[i]) { gt = 1; break; } if (y.l[i] < FP_HALF_PM1.l[i]) break; }
    if (gt) out[0] |= 0x20;
}
API void ko_g1_to_compressed_batch(uint8_t *out, const g1_t *p, u64 n) { for (u64 i = 0; i < n; i++) ko_g1_to_compressed(out + 48 * i, &p[i]); }
/* bls.FromCompressedG1 (bls/bls_kilic.go:118-121). No subgroup check (setup inputs are trusted). */
API int ko_g1_from_compressed(g1_t *o, const uint8_t *in) {
    if (!(in[0] & 0x80)) return KO_ERR_BAD_POINT;
    if (in[0] & 0x40) {
        if (in[0] & 0x3f) return KO_ERR_BAD_POINT;
        for (int i = 1; i < 48; i++) if (in[i]) return KO_ERR_BAD_POINT;
        g1_set_inf(o); return KO_OK;
    }
    fp_t x; memset(&x, 0, sizeof x);
    for (int i = 0; i < 48; i++) { uint8_t b = in[i]; if (i == 0) b &= 0x1f; x.l[(47 - i) / 8] |= (u64)b << (8 * ((47 - i) % 8)); }
    if (limbs_geq(x.l, FP_P.l, 6)) return KO_ERR_BAD_POINT;
    fp_t xm, y2, y; fp_to_mont(&xm, &x);
    fp_sqr(&y2, &xm); fp_mul(&y2, &y2, &xm); fp_add(&y2, &y2, &FP_B);
    if (!fp_sqrt(&y, &y2)) return KO_ERR_BAD_POINT;
    fp_t ys; fp_from_mont(&ys, &y);
    int gt = 0;
    for (int i = 5; i >= 0; i--) { if (ys.l[i] > FP_HALF_PM1.l[i]) { gt = 1; break; } if (ys.l[i] < FP_HALF_PM1.l[i]) break; }
    if (gt != !!(in[0] & 0x20)) fp_neg(&y, &y);
    o->x = xm; o->y = y; o->z = FP_ONE;
    /* Kilic G1.FromCompressed rejects curve points outside the order-r subgroup ("point is not on correct subgroup")
     * [restated from memory of kilic/bls12-381 g1.go]: [r]P must be the point at infinity */
    g1_t t; g1_set_inf(&t);
    for (int i = 254; i >= 0; i--) { g1_dbl(&t, &t); if ((FR_R.l[i / 64] >> (i % 64)) & 1) g1_add(&t, &t, o); }
    if (!g1_is_inf(&t)) return KO_ERR_BAD_POINT;
    return KO_OK;
}
API int ko_g1_from_compressed_batch(g1_t *o, const uint8_t *in, u64 n) {
    for (u64 i = 0; i < n; i++) { int s = ko_g1_from_compressed(&o[i], in + 48 * i); if (s) return s; }
    return KO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * bls.LinCombG1 (bls/bls_kilic.go:132-150) -> Kilic G1.MultiExp [restated from memory]:
 * scalars leave Montgomery form (FromRed, :141-147); window c = 3 if n < 32 else ceil(ln n);
 * per window: every point is added into bucket[digit-1]; buckets are folded by a running sum;
 * windows are combined MSB-first with c doublings each.  Empty input -> inf (bls_test.go:69-78).
 * ---------------------------------------------------------------------------------------------- */
API int ko_lincomb_g1(g1_t *out, const g1_t *pts, const fr_t *scalars, u64 n) {
    g1_set_inf(out);
    if (n == 0) return KO_OK;
    int c = 3;
    if (n >= 32) c = (int)ceil(log((double)n));
    int nb = (1 << c) - 1;
    int nwin = 255 / c + 1;
    fr_t *ks = malloc(n * sizeof(fr_t));
    g1_t *bucket = malloc((size_t)nb * sizeof(g1_t));
    g1_t *wins = malloc((size_t)nwin * sizeof(g1_t));
    for (u64 i = 0; i < n; i++) fr_from_mont(&ks[i], &scalars[i]);
    for (int w = 0; w < nwin; w++) {
        for (int b = 0; b < nb; b++) g1_set_inf(&bucket[b]);
        int sh = w * c;
        for (u64 i = 0; i < n; i++) {
            int limb = sh / 64, off = sh % 64;
            u64 d = ks[i].l[limb] >> off;
            if (off + c > 64 && limb < 3) d |= ks[i].l[limb + 1] << (64 - off);
            d &= (u64)nb;
            if (d) g1_add(&bucket[d - 1], &bucket[d - 1], &pts[i]);
        }
        g1_t acc, sum; g1_set_inf(&acc); g1_set_inf(&sum);
        for (int b = nb - 1; b >= 0; b--) { g1_add(&sum, &sum, &bucket[b]); g1_add(&acc, &acc, &sum); }
        wins[w] = acc;
    }
    g1_t acc; g1_set_inf(&acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int j = 0; j < c; j++) g1_dbl(&acc, &acc);
        g1_add(&acc, &acc, &wins[w]);
    }
    *out = acc;
    free(ks); free(bucket); free(wins);
    return KO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * FFTSettings (fft.go:34-61)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    u64 max_width;
    fr_t root_of_unity;
    fr_t *expanded;   /* max_width + 1 entries: w^i, first and last = 1  (fft.go:21-32) */
    fr_t *reversed;   /* expanded reversed                                (fft.go:49-54) */
} ko_fft_t;

/* bls.Scale2RootOfUnity[k] = 7^((r-1)/2^k) (bls/globals.go:24-60; checked against the decimals in tests) */
API void ko_scale2_root_of_unity(fr_t *o, unsigned k) {
    u64 e[4]; memcpy(e, FR_R.l, 32); e[0] -= 1;
    for (unsigned s = 0; s < k; s++) for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
    fr_t seven; fr_from_u64(&seven, 7);
    fr_pow(o, &seven, e, 4);
}
API ko_fft_t *ko_fft_settings_new(unsigned max_scale) {
    if (max_scale > 31) return NULL;
    ko_fft_t *fs = calloc(1, sizeof *fs);
    fs->max_width = 1ULL << max_scale;
    ko_scale2_root_of_unity(&fs->root_of_unity, max_scale);
    fs->expanded = malloc((fs->max_width + 1) * sizeof(fr_t));
    fs->reversed = malloc((fs->max_width + 1) * sizeof(fr_t));
    fs->expanded[0] = FR_ONE;
    for (u64 i = 1; i <= fs->max_width; i++) fr_mul(&fs->expanded[i], &fs->expanded[i - 1], &fs->root_of_unity);
    for (u64 i = 0; i <= fs->max_width; i++) fs->reversed[i] = fs->expanded[fs->max_width - i];
    return fs;
}
API void ko_fft_settings_free(ko_fft_t *fs) { if (fs) { free(fs->expanded); free(fs->reversed); free(fs); } }
API u64 ko_fft_max_width(const ko_fft_t *fs) { return fs->max_width; }
API const fr_t *ko_fft_expanded_roots(const ko_fft_t *fs) { return fs->expanded; }
API const fr_t *ko_fft_reverse_roots(const ko_fft_t *fs) { return fs->reversed; }

static int is_pow2(u64 v) { return (v & (v - 1)) == 0; }   /* bls.IsPowerOfTwo, globals.go:72-74 (true for 0) */

/* simpleFT (fft_fr.go:8-28) */
static void simple_ft(const fr_t *vals, u64 off, u64 stride, const fr_t *roots, u64 rstride, fr_t *out, u64 l) {
    for (u64 i = 0; i < l; i++) {
        fr_t v, last;
        fr_mul(&v, &vals[off], &roots[0]); last = v;
        for (u64 j = 1; j < l; j++) {
            fr_mul(&v, &vals[off + j * stride], &roots[((i * j) % l) * rstride]);
            fr_add(&last, &last, &v);
        }
        out[i] = last;
    }
}
/* _fft (fft_fr.go:30-53) */
static void fft_rec(const fr_t *vals, u64 off, u64 stride, const fr_t *roots, u64 rstride, fr_t *out, u64 l) {
    if (l <= 4) { simple_ft(vals, off, stride, roots, rstride, out, l); return; }
    u64 half = l >> 1;
    fft_rec(vals, off, stride << 1, roots, rstride << 1, out, half);
    fft_rec(vals, off + stride, stride << 1, roots, rstride << 1, out + half, half);
    for (u64 i = 0; i < half; i++) {
        fr_t x = out[i], y = out[i + half], yr;
        fr_mul(&yr, &y, &roots[i * rstride]);
        fr_add(&out[i], &x, &yr);
        fr_sub(&out[i + half], &x, &yr);
    }
}
/* InplaceFFT (fft_fr.go:76-105) */
API int ko_inplace_fft(const ko_fft_t *fs, const fr_t *vals, fr_t *out, u64 n, int inv) {
    if (n > fs->max_width) return KO_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KO_ERR_NOT_POW2;
    if (n == 0) return KO_OK;
    u64 stride = fs->max_width / n;
    if (inv) {
        fr_t inv_len; fr_from_u64(&inv_len, n); fr_inv(&inv_len, &inv_len);
        fft_rec(vals, 0, 1, fs->reversed, stride, out, n);
        for (u64 i = 0; i < n; i++) fr_mul(&out[i], &out[i], &inv_len);
    } else {
        fft_rec(vals, 0, 1, fs->expanded, stride, out, n);
    }
    return KO_OK;
}
static u64 next_pow2(u64 v) { if (v == 0) return 1; u64 p = 1; while (p < v) p <<= 1; return p; }   /* fft.go:11-16 */
/* FFT (fft_fr.go:55-74): zero-pads to the next power of two. *out_n receives the padded length;
 * out must have room for next_pow2(n) elements. */
API int ko_fft_fr(const ko_fft_t *fs, const fr_t *vals, u64 n, int inv, fr_t *out, u64 *out_n) {
    if (n > fs->max_width) return KO_ERR_TOO_WIDE;
    u64 np = next_pow2(n);
    fr_t *copy = calloc(np, sizeof(fr_t));
    memcpy(copy, vals, n * sizeof(fr_t));
    int s = ko_inplace_fft(fs, copy, out, np, inv);
    free(copy);
    if (out_n) *out_n = np;
    return s;
}

/* simpleFTG1 / _fftG1 / FFTG1 (fft_g1.go:11-94) */
static void simple_ft_g1(const g1_t *vals, u64 off, u64 stride, const fr_t *roots, u64 rstride, g1_t *out, u64 l) {
    for (u64 i = 0; i < l; i++) {
        g1_t v, last;
        g1_mul(&v, &vals[off], &roots[0]); last = v;
        for (u64 j = 1; j < l; j++) {
            g1_mul(&v, &vals[off + j * stride], &roots[((i * j) % l) * rstride]);
            g1_add(&last, &last, &v);
        }
        out[i] = last;
    }
}
static void fft_g1_rec(const g1_t *vals, u64 off, u64 stride, const fr_t *roots, u64 rstride, g1_t *out, u64 l) {
    if (l <= 4) { simple_ft_g1(vals, off, stride, roots, rstride, out, l); return; }
    u64 half = l >> 1;
    fft_g1_rec(vals, off, stride << 1, roots, rstride << 1, out, half);
    fft_g1_rec(vals, off + stride, stride << 1, roots, rstride << 1, out + half, half);
    for (u64 i = 0; i < half; i++) {
        g1_t x = out[i], y = out[i + half], yr;
        g1_mul(&yr, &y, &roots[i * rstride]);
        g1_add(&out[i], &x, &yr);
        g1_sub(&out[i + half], &x, &yr);
    }
}
API int ko_fft_g1(const ko_fft_t *fs, const g1_t *vals, u64 n, int inv, g1_t *out) {
    if (n > fs->max_width) return KO_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KO_ERR_NOT_POW2;
    if (n == 0) return KO_ERR_BAD_ARG;   /* the reference divides by zero here (fft_g1.go:76); rejected */
    u64 stride = fs->max_width / n;
    if (inv) {
        fr_t inv_len; fr_from_u64(&inv_len, n); fr_inv(&inv_len, &inv_len);
        fft_g1_rec(vals, 0, 1, fs->reversed, stride, out, n);
        for (u64 i = 0; i < n; i++) g1_mul(&out[i], &out[i], &inv_len);
    } else {
        fft_g1_rec(vals, 0, 1, fs->expanded, stride, out, n);
    }
    return KO_OK;
}

/* dASFFTExtension / DASFFTExtension (das_extension.go:7-84), in place */
static void das_rec(const ko_fft_t *fs, fr_t *ab, u64 len, u64 ds) {
    if (len == 2) {
        fr_t x, y, t;
        fr_add(&x, &ab[0], &ab[1]); fr_sub(&y, &ab[0], &ab[1]);
        fr_mul(&t, &y, &fs->expanded[ds]);
        fr_add(&ab[0], &x, &t); fr_sub(&ab[1], &x, &t);
        return;
    }
    u64 hh = len >> 1;
    fr_t *h0 = ab, *h1 = ab + hh;
    for (u64 i = 0; i < hh; i++) {
        fr_t t1, t2;
        fr_add(&t1, &h0[i], &h1[i]); fr_sub(&t2, &h0[i], &h1[i]);
        fr_mul(&h1[i], &t2, &fs->reversed[i * 2 * ds]);
        h0[i] = t1;
    }
    das_rec(fs, h0, hh, ds << 1);
    das_rec(fs, h1, hh, ds << 1);
    for (u64 i = 0; i < hh; i++) {
        fr_t x = h0[i], y = h1[i], yr;
        fr_mul(&yr, &y, &fs->expanded[(1 + 2 * i) * ds]);
        fr_add(&h0[i], &x, &yr); fr_sub(&h1[i], &x, &yr);
    }
}
API int ko_das_fft_extension(const ko_fft_t *fs, fr_t *vals, u64 n) {
    if (n * 2 > fs->max_width) return KO_ERR_TOO_WIDE;   /* panic das_extension.go:72-74 */
    if (n < 2 || !is_pow2(n)) return KO_ERR_BAD_ARG;     /* "bad usage" :22-24 */
    das_rec(fs, vals, n, 1);
    fr_t inv_len; fr_from_u64(&inv_len, n); fr_inv(&inv_len, &inv_len);
    for (u64 i = 0; i < n; i++) fr_mul(&vals[i], &vals[i], &inv_len);
    return KO_OK;
}

/* reverseBitOrder (reverse_bit_order.go:74-101) */
static uint32_t rev_bits_limited(uint32_t length, uint32_t v) {
    unsigned bits = 0; while ((1u << (bits + 1)) <= length) bits++;   /* bitIndex = floor(log2) :25-53 */
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) if (v & (1u << i)) r |= 1u << (bits - 1 - i);
    return r;
}
API uint32_t ko_reverse_bits_limited(uint32_t length, uint32_t v) { return rev_bits_limited(length, v); }
API void ko_reverse_bit_order_fr(fr_t *v, u64 n) {
    for (uint32_t i = 0; i < n; i++) { uint32_t r = rev_bits_limited((uint32_t)n, i); if (r > i) { fr_t t = v[i]; v[i] = v[r]; v[r] = t; } }
}
API void ko_reverse_bit_order_g1(g1_t *v, u64 n) {
    for (uint32_t i = 0; i < n; i++) { uint32_t r = rev_bits_limited((uint32_t)n, i); if (r > i) { g1_t t = v[i]; v[i] = v[r]; v[r] = t; } }
}

/* GenerateTestingSetup (setup.go:9-26), G1 half only (G2 is verifier-side, out of scope) */
API void ko_generate_testing_setup_g1(const fr_t *secret, u64 n, g1_t *out) {
    fr_t spow = FR_ONE; g1_t gen; ko_g1_generator(&gen);
    for (u64 i = 0; i < n; i++) { g1_mul(&out[i], &gen, &spow); fr_mul(&spow, &spow, secret); }
}

/* ------------------------------------------------------------------------------------------------
 * KZG single proofs (kzg_single_proofs.go, poly.go)
 * ---------------------------------------------------------------------------------------------- */
/* polyLongDiv (poly.go:14-40) incl. the per-step inversion of the divisor's leading coefficient */
static fr_t *poly_long_div(const fr_t *dividend, u64 na, const fr_t *divisor, u64 nb, u64 *nout) {
    fr_t *a = malloc(na * sizeof(fr_t)); memcpy(a, dividend, na * sizeof(fr_t));
    long apos = (long)na - 1, bpos = (long)nb - 1, diff = apos - bpos;
    fr_t *out = calloc(diff >= 0 ? diff + 1 : 1, sizeof(fr_t));
    *nout = diff >= 0 ? (u64)diff + 1 : 0;
    while (diff >= 0) {
        fr_t inv, quot; fr_inv(&inv, &divisor[bpos]); fr_mul(&quot, &inv, &a[apos]);   /* polyFactorDiv :6-11 */
        out[diff] = quot;
        for (long i = bpos; i >= 0; i--) { fr_t t; fr_mul(&t, &quot, &divisor[i]); fr_sub(&a[diff + i], &a[diff + i], &t); }
        apos--; diff--;
    }
    free(a); return out;
}
/* CommitToPoly (kzg_single_proofs.go:17-19) */
API int ko_commit_to_poly(const g1_t *secret_g1, u64 n_setup, const fr_t *coeffs, u64 n, g1_t *out) {
    if (n > n_setup) return KO_ERR_LEN_MISMATCH;   /* Go slice bounds panic */
    return ko_lincomb_g1(out, secret_g1, coeffs, n);
}
/* ComputeProofSingle (kzg_single_proofs.go:36-54) */
API int ko_compute_proof_single(const g1_t *secret_g1, u64 n_setup, const fr_t *poly, u64 n, u64 x, g1_t *out) {
    if (n < 2) return KO_ERR_BAD_ARG;
    fr_t divisor[2], xf, zero; memset(&zero, 0, sizeof zero);
    fr_from_u64(&xf, x); fr_sub(&divisor[0], &zero, &xf); divisor[1] = FR_ONE;
    u64 nq; fr_t *q = poly_long_div(poly, n, divisor, 2, &nq);
    int s = nq > n_setup ? KO_ERR_LEN_MISMATCH : ko_lincomb_g1(out, secret_g1, q, nq);
    free(q); return s;
}
API void ko_poly_quotient_linear(const fr_t *poly, u64 n, u64 x, fr_t *q_out) {   /* the quotient itself, for tests */
    fr_t divisor[2], xf, zero; memset(&zero, 0, sizeof zero);
    fr_from_u64(&xf, x); fr_sub(&divisor[0], &zero, &xf); divisor[1] = FR_ONE;
    u64 nq; fr_t *q = poly_long_div(poly, n, divisor, 2, &nq);
    memcpy(q_out, q, nq * sizeof(fr_t)); free(q);
}

/* ComputeProofMulti (kzg_multi_proofs.go:13-43).  NOTE the reference never initialises xPowN to ONE (:20-24), so the loop
 * leaves it at zero and the divisor is X^n, not X^n - x^n; restated faithfully. */
API int ko_compute_proof_multi(const g1_t *secret_g1, u64 n_setup, const fr_t *poly, u64 len, u64 x, u64 n, g1_t *out) {
    if (len < n + 1) return KO_ERR_BAD_ARG;
    fr_t *divisor = calloc(n + 1, sizeof(fr_t));
    fr_t xf, xpown, tmp, zero; memset(&zero, 0, sizeof zero); memset(&xpown, 0, sizeof xpown);
    fr_from_u64(&xf, x);
    for (u64 i = 0; i < n; i++) { fr_mul(&tmp, &xpown, &xf); xpown = tmp; }
    fr_sub(&divisor[0], &zero, &xpown);
    divisor[n] = FR_ONE;
    u64 nq; fr_t *q = poly_long_div(poly, len, divisor, n + 1, &nq);
    int s = nq > n_setup ? KO_ERR_LEN_MISMATCH : ko_lincomb_g1(out, secret_g1, q, nq);
    free(q); free(divisor); return s;
}
/* prover-side part of CheckProofMulti (kzg_multi_proofs.go:47-75): interpolation polynomial on the coset x * <w_n>
 * (IFFT of ys, coefficient i divided by x^i), its commitment [I(s)]_1, and x^n.  The pairing itself is out of scope. */
API int ko_check_proof_multi_interpolation(const ko_fft_t *fs, const g1_t *secret_g1, u64 n_setup, const fr_t *ys, u64 n, const fr_t *x,
                                           g1_t *is1, fr_t *xpow_out) {
    fr_t *ip = malloc(next_pow2(n) * sizeof(fr_t)); u64 np;
    int s = ko_fft_fr(fs, ys, n, 1, ip, &np);
    if (s) { free(ip); return s; }
    fr_t xpow = FR_ONE, tmp;
    for (u64 i = 0; i < np; i++) { fr_inv(&tmp, &xpow); fr_mul(&ip[i], &ip[i], &tmp); fr_mul(&xpow, &xpow, x); }
    if (xpow_out) *xpow_out = xpow;
    s = np > n_setup ? KO_ERR_LEN_MISMATCH : ko_lincomb_g1(is1, secret_g1, ip, np);
    free(ip); return s;
}

/* ------------------------------------------------------------------------------------------------
 * FK20 (kzg.go:38-116, fk20_single.go, fk20_multi.go)
 * ---------------------------------------------------------------------------------------------- */
/* toeplitzPart1 (fk20_single.go:40-56) */
static int toeplitz_part1(const ko_fft_t *fs, const g1_t *x, u64 n, g1_t *out /* 2n */) {
    g1_t *ext = malloc(2 * n * sizeof(g1_t));
    memcpy(ext, x, n * sizeof(g1_t));
    for (u64 i = n; i < 2 * n; i++) g1_set_inf(&ext[i]);
    int s = ko_fft_g1(fs, ext, 2 * n, 0, out);
    free(ext); return s;
}
/* ToeplitzPart2 (fk20_single.go:59-77) */
API int ko_toeplitz_part2(const ko_fft_t *fs, const fr_t *coeffs, const g1_t *x_ext_fft, u64 n, g1_t *h_ext_fft) {
    fr_t *cf = malloc(n * sizeof(fr_t)); u64 np;
    int s = ko_fft_fr(fs, coeffs, n, 0, cf, &np);
    if (!s) for (u64 i = 0; i < n; i++) g1_mul(&h_ext_fft[i], &x_ext_fft[i], &cf[i]);
    free(cf); return s;
}
/* ToeplitzPart3 (fk20_single.go:80-87): out has n entries, the caller keeps the first n/2 */
API int ko_toeplitz_part3(const ko_fft_t *fs, const g1_t *h_ext_fft, u64 n, g1_t *out) { return ko_fft_g1(fs, h_ext_fft, n, 1, out); }
/* toeplitzCoeffsStepStrided (fk20_single.go:89-103); toeplitzCoeffsStep (:106-119) == offset 0, stride 1 */
static void toeplitz_coeffs_strided(const fr_t *poly, u64 n, u64 offset, u64 stride, fr_t *out /* 2k */) {
    u64 k = n / stride, k2 = 2 * k;
    memset(out, 0, k2 * sizeof(fr_t));
    out[0] = poly[n - 1 - offset];
    for (u64 i = k + 2, j = 2 * stride - offset - 1; i < k2; i++, j += stride) out[i] = poly[j];
}
API void ko_toeplitz_coeffs_step_strided(const fr_t *poly, u64 n, u64 offset, u64 stride, fr_t *out) { toeplitz_coeffs_strided(poly, n, offset, stride, out); }

typedef struct { const ko_fft_t *fs; u64 n2; g1_t *x_ext_fft; } ko_fk20s_t;
/* NewFK20SingleSettings (kzg.go:43-64) */
API ko_fk20s_t *ko_fk20_single_new(const ko_fft_t *fs, const g1_t *secret_g1, u64 n_setup, u64 n2, int *status) {
    int st = KO_OK;
    if (n2 > fs->max_width) st = KO_ERR_TOO_WIDE; else if (!is_pow2(n2)) st = KO_ERR_NOT_POW2; else if (n2 < 2) st = KO_ERR_BAD_ARG;
    else if (n_setup < fs->max_width) st = KO_ERR_LEN_MISMATCH;   /* NewKZGSettings kzg.go:25-27 */
    if (status) *status = st;
    if (st) return NULL;
    u64 n = n2 / 2;
    ko_fk20s_t *fk = calloc(1, sizeof *fk); fk->fs = fs; fk->n2 = n2;
    g1_t *x = malloc(n * sizeof(g1_t));
    for (u64 i = 0; i + 1 < n; i++) x[i] = secret_g1[n - 2 - i];
    g1_set_inf(&x[n - 1]);
    fk->x_ext_fft = malloc(n2 * sizeof(g1_t));
    toeplitz_part1(fs, x, n, fk->x_ext_fft);
    free(x); return fk;
}
API void ko_fk20_single_free(ko_fk20s_t *fk) { if (fk) { free(fk->x_ext_fft); free(fk); } }
API const g1_t *ko_fk20_single_x_ext_fft(const ko_fk20s_t *fk) { return fk->x_ext_fft; }
/* FK20Single (fk20_single.go:122-134): n coefficients -> n proofs. Needs 2n <= max_width. */
API int ko_fk20_single(const ko_fk20s_t *fk, const fr_t *poly, u64 n, g1_t *out) {
    if (2 * n != fk->n2) return KO_ERR_LEN_MISMATCH;
    fr_t *tc = malloc(2 * n * sizeof(fr_t)); g1_t *h = malloc(2 * n * sizeof(g1_t)), *h2 = malloc(2 * n * sizeof(g1_t));
    toeplitz_coeffs_strided(poly, n, 0, 1, tc);
    int s = ko_toeplitz_part2(fk->fs, tc, fk->x_ext_fft, 2 * n, h);
    if (!s) s = ko_toeplitz_part3(fk->fs, h, 2 * n, h2);
    if (!s) s = ko_fft_g1(fk->fs, h2, n, 0, out);
    free(tc); free(h); free(h2); return s;
}
/* FK20SingleDAOptimized (fk20_single.go:139-172): n2 coefficients (upper half zero) -> n2 proofs */
API int ko_fk20_single_da_optimized(const ko_fk20s_t *fk, const fr_t *poly, u64 n2, g1_t *out) {
    if (n2 > fk->fs->max_width) return KO_ERR_TOO_WIDE;
    if (!is_pow2(n2)) return KO_ERR_NOT_POW2;
    if (n2 != fk->n2) return KO_ERR_LEN_MISMATCH;
    u64 n = n2 / 2;
    for (u64 i = n; i < n2; i++) if (!fr_is_zero(&poly[i])) return KO_ERR_UPPER_HALF;
    fr_t *tc = malloc(n2 * sizeof(fr_t)); g1_t *h = malloc(n2 * sizeof(g1_t)), *h2 = malloc(n2 * sizeof(g1_t));
    toeplitz_coeffs_strided(poly, n, 0, 1, tc);
    int s = ko_toeplitz_part2(fk->fs, tc, fk->x_ext_fft, n2, h);
    if (!s) s = ko_toeplitz_part3(fk->fs, h, n2, h2);
    if (!s) { for (u64 i = n; i < n2; i++) g1_set_inf(&h2[i]); s = ko_fft_g1(fk->fs, h2, n2, 0, out); }
    free(tc); free(h); free(h2); return s;
}
/* DAUsingFK20 (fk20_single.go:176-196): n coefficients -> 2n proofs in reverse-bit order */
API int ko_da_using_fk20(const ko_fk20s_t *fk, const fr_t *poly, u64 n, g1_t *out) {
    if (n > fk->fs->max_width / 2) return KO_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KO_ERR_NOT_POW2;
    fr_t *ext = calloc(2 * n, sizeof(fr_t)); memcpy(ext, poly, n * sizeof(fr_t));
    int s = ko_fk20_single_da_optimized(fk, ext, 2 * n, out);
    if (!s) ko_reverse_bit_order_g1(out, 2 * n);
    free(ext); return s;
}

typedef struct { const ko_fft_t *fs; u64 n2, chunk_len; g1_t **files; } ko_fk20m_t;
/* NewFK20MultiSettings (kzg.go:73-116) */
API ko_fk20m_t *ko_fk20_multi_new(const ko_fft_t *fs, const g1_t *secret_g1, u64 n_setup, u64 n2, u64 chunk_len, int *status) {
    int st = KO_OK;
    if (n2 > fs->max_width) st = KO_ERR_TOO_WIDE; else if (!is_pow2(n2)) st = KO_ERR_NOT_POW2; else if (n2 < 2) st = KO_ERR_BAD_ARG;
    else if (chunk_len > n2 / 2 || chunk_len < 1) st = KO_ERR_BAD_ARG; else if (!is_pow2(chunk_len)) st = KO_ERR_NOT_POW2;
    else if (n_setup < fs->max_width) st = KO_ERR_LEN_MISMATCH;
    if (status) *status = st;
    if (st) return NULL;
    u64 n = n2 / 2, k = n / chunk_len;
    ko_fk20m_t *fk = calloc(1, sizeof *fk); fk->fs = fs; fk->n2 = n2; fk->chunk_len = chunk_len;
    fk->files = calloc(chunk_len, sizeof(g1_t *));
    g1_t *x = malloc(k * sizeof(g1_t));
    for (u64 off = 0; off < chunk_len; off++) {
        u64 start = n - chunk_len - 1 - off;
        for (u64 i = 0, j = start; i + 1 < k; i++, j -= chunk_len) x[i] = secret_g1[j];
        g1_set_inf(&x[k - 1]);
        fk->files[off] = malloc(2 * k * sizeof(g1_t));
        toeplitz_part1(fs, x, k, fk->files[off]);
    }
    free(x); return fk;
}
API void ko_fk20_multi_free(ko_fk20m_t *fk) { if (fk) { for (u64 i = 0; i < fk->chunk_len; i++) free(fk->files[i]); free(fk->files); free(fk); } }
API const g1_t *ko_fk20_multi_file(const ko_fk20m_t *fk, u64 i) { return fk->files[i]; }
/* shared body of FK20Multi (fk20_multi.go:25-52) and FK20MultiDAOptimized (:58-109) */
static int fk20_multi_body(const ko_fk20m_t *fk, const fr_t *poly, u64 n, int da, g1_t *out) {
    u64 l = fk->chunk_len, k = n / l, k2 = 2 * k;
    g1_t *hext = malloc(k2 * sizeof(g1_t)), *file = malloc(k2 * sizeof(g1_t)), *h = malloc(k2 * sizeof(g1_t));
    fr_t *tc = malloc(k2 * sizeof(fr_t));
    for (u64 j = 0; j < k2; j++) g1_set_inf(&hext[j]);
    int s = KO_OK;
    for (u64 i = 0; i < l && !s; i++) {
        toeplitz_coeffs_strided(poly, n, i, l, tc);
        s = ko_toeplitz_part2(fk->fs, tc, fk->files[i], k2, file);
        if (!s) for (u64 j = 0; j < k2; j++) g1_add(&hext[j], &hext[j], &file[j]);
    }
    if (!s) s = ko_toeplitz_part3(fk->fs, hext, k2, h);
    if (!s) {
        if (da) { for (u64 i = k; i < k2; i++) g1_set_inf(&h[i]); s = ko_fft_g1(fk->fs, h, k2, 0, out); }
        else s = ko_fft_g1(fk->fs, h, k, 0, out);
    }
    free(hext); free(file); free(h); free(tc); return s;
}
/* FK20Multi: n coefficients -> k = n/l proofs. NOTE the reference sizes hExtFFT as 2n (fk20_multi.go:33)
 * while ToeplitzPart2 returns 2k entries; for l > 1 its loop `for j < n2` would index past the file, so the
 * reference function only runs for l == 1.  Restated here with the evident intent (2k). */
API int ko_fk20_multi(const ko_fk20m_t *fk, const fr_t *poly, u64 n, g1_t *out) {
    if (fk->fs->max_width < 2 * n) return KO_ERR_TOO_WIDE;
    if (2 * n != fk->n2) return KO_ERR_LEN_MISMATCH;
    return fk20_multi_body(fk, poly, n, 0, out);
}
API int ko_fk20_multi_da_optimized(const ko_fk20m_t *fk, const fr_t *poly, u64 n2, g1_t *out) {
    if (fk->fs->max_width < n2) return KO_ERR_TOO_WIDE;
    if (n2 != fk->n2) return KO_ERR_LEN_MISMATCH;
    u64 n = n2 / 2;
    for (u64 i = n; i < n2; i++) if (!fr_is_zero(&poly[i])) return KO_ERR_UPPER_HALF;
    return fk20_multi_body(fk, poly, n, 1, out);
}
/* DAUsingFK20Multi (fk20_multi.go:113-133): n coefficients -> 2k proofs in reverse-bit order */
API int ko_da_using_fk20_multi(const ko_fk20m_t *fk, const fr_t *poly, u64 n, g1_t *out) {
    if (n > fk->fs->max_width / 2) return KO_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KO_ERR_NOT_POW2;
    fr_t *ext = calloc(2 * n, sizeof(fr_t)); memcpy(ext, poly, n * sizeof(fr_t));
    int s = ko_fk20_multi_da_optimized(fk, ext, 2 * n, out);
    if (!s) ko_reverse_bit_order_g1(out, 2 * (n / fk->chunk_len));
    free(ext); return s;
}

/* ------------------------------------------------------------------------------------------------
 * Erasure recovery: zero_poly.go, recover_from_samples.go
 * ---------------------------------------------------------------------------------------------- */
/* makeZeroPolyMulLeaf (zero_poly.go:17-44) */
static void zero_poly_leaf(const ko_fft_t *fs, fr_t *dst, u64 dst_len, const u64 *indices, u64 n_idx, u64 stride) {
    fr_t zero; memset(&zero, 0, sizeof zero);
    for (u64 i = n_idx + 1; i < dst_len; i++) dst[i] = zero;
    dst[n_idx] = FR_ONE;
    for (u64 i = 0; i < n_idx; i++) {
        fr_t neg; fr_sub(&neg, &zero, &fs->expanded[indices[i] * stride]);
        dst[i] = neg;
        if (i > 0) {
            fr_add(&dst[i], &dst[i], &dst[i - 1]);
            for (u64 j = i - 1; j > 0; j--) { fr_mul(&dst[j], &dst[j], &neg); fr_add(&dst[j], &dst[j], &dst[j - 1]); }
            fr_mul(&dst[0], &dst[0], &neg);
        }
    }
}
/* reduceLeaves (zero_poly.go:58-111): product of the polynomials ps[0..cnt) via FFT of size n = dst length */
static u64 reduce_leaves(const ko_fft_t *fs, fr_t *scratch, fr_t *dst, u64 n, fr_t **ps, const u64 *lens, u64 cnt) {
    u64 out_degree = 0;
    for (u64 i = 0; i < cnt; i++) out_degree += lens[i] - 1;
    fr_t *padded = scratch, *mul_eval = scratch + n, *p_eval = scratch + 2 * n;
    u64 last = cnt - 1;
    memset(padded, 0, n * sizeof(fr_t)); memcpy(padded, ps[last], lens[last] * sizeof(fr_t));
    ko_inplace_fft(fs, padded, mul_eval, n, 0);
    for (u64 i = 0; i < last; i++) {
        /* the reference copies ps[i] over the previous padded buffer without clearing the tail (zero_poly.go:93-96); all leaves
         * it multiplies have equal length except the last one, which it loads first, so the tail is already zero */
        memset(padded, 0, n * sizeof(fr_t)); memcpy(padded, ps[i], lens[i] * sizeof(fr_t));
        ko_inplace_fft(fs, padded, p_eval, n, 0);
        for (u64 j = 0; j < n; j++) fr_mul(&mul_eval[j], &mul_eval[j], &p_eval[j]);
    }
    ko_inplace_fft(fs, mul_eval, dst, n, 1);
    return out_degree + 1;
}
/* ZeroPolyViaMultiplication (zero_poly.go:116-217): zero_eval and zero_poly have `length` entries each */
API int ko_zero_poly_via_multiplication(const ko_fft_t *fs, const u64 *missing, u64 n_missing, u64 length, fr_t *zero_eval, fr_t *zero_poly) {
    memset(zero_eval, 0, length * sizeof(fr_t)); memset(zero_poly, 0, length * sizeof(fr_t));
    if (n_missing == 0) return KO_OK;
    if (length > fs->max_width) return KO_ERR_TOO_WIDE;
    if (!is_pow2(length)) return KO_ERR_NOT_POW2;
    u64 stride = fs->max_width / length, per_leaf_poly = 64, per_leaf = 63;
    if (n_missing <= per_leaf) {
        if (n_missing + 1 > length) return KO_ERR_BAD_ARG;
        zero_poly_leaf(fs, zero_poly, n_missing + 1, missing, n_missing, stride);
        return ko_inplace_fft(fs, zero_poly, zero_eval, length, 0);
    }
    u64 leaf_count = (n_missing + per_leaf - 1) / per_leaf;
    u64 n = next_pow2(leaf_count * per_leaf_poly);
    fr_t *out = calloc(n, sizeof(fr_t)), *scratch = calloc(3 * n, sizeof(fr_t));
    fr_t **leaves = calloc(leaf_count, sizeof(fr_t *)); u64 *lens = calloc(leaf_count, sizeof(u64));
    for (u64 i = 0, off = 0; i < leaf_count; i++, off += per_leaf) {
        u64 end = off + per_leaf > n_missing ? n_missing : off + per_leaf;
        leaves[i] = out + i * per_leaf_poly; lens[i] = per_leaf_poly;
        zero_poly_leaf(fs, leaves[i], per_leaf_poly, missing + off, end - off, stride);
    }
    u64 nleaves = leaf_count;
    fr_t *tmp = calloc(n, sizeof(fr_t));
    while (nleaves > 1) {
        u64 reduced = (nleaves + 3) / 4, leaf_size = next_pow2(lens[0]);
        for (u64 i = 0; i < reduced; i++) {
            u64 start = i * 4, end = start + 4, out_end = end * leaf_size;
            if (out_end > n) out_end = n;
            if (end > nleaves) end = nleaves;
            fr_t *dst = out + start * leaf_size; u64 dst_n = out_end - start * leaf_size;
            if (end > start + 1) {
                /* inputs live inside dst: multiply into tmp, then copy back (the reference reads them before the final IFFT writes) */
                u64 len = reduce_leaves(fs, scratch, tmp, dst_n, leaves + start, lens + start, end - start);
                memcpy(dst, tmp, dst_n * sizeof(fr_t));
                leaves[i] = dst; lens[i] = len;
            } else { leaves[i] = dst; lens[i] = dst_n < lens[start] ? dst_n : lens[start]; if (leaves[start] != dst) memmove(dst, leaves[start], lens[i] * sizeof(fr_t)); }
        }
        nleaves = reduced;
    }
    int st = KO_OK;
    if (lens[0] > length) st = KO_ERR_BAD_ARG;   /* "expected output smaller or equal to input length" */
    else { memcpy(zero_poly, leaves[0], lens[0] * sizeof(fr_t)); st = ko_inplace_fft(fs, zero_poly, zero_eval, length, 0); }
    free(out); free(scratch); free(leaves); free(lens); free(tmp);
    return st;
}
/* ShiftPoly / UnshiftPoly (recover_from_samples.go:9-40): poly[i] *= 5^-i  /  5^i */
static void shift_poly(fr_t *poly, u64 n, int unshift) {
    fr_t f, pw = FR_ONE; fr_from_u64(&f, 5);
    if (!unshift) fr_inv(&f, &f);
    for (u64 i = 0; i < n; i++) { fr_mul(&poly[i], &poly[i], &pw); fr_mul(&pw, &pw, &f); }
}
/* RecoverPolyFromSamples (recover_from_samples.go:42-109) with ZeroPolyViaMultiplication; present[i] == 0 <=> samples[i] == nil */
API int ko_recover_poly_from_samples(const ko_fft_t *fs, const fr_t *samples, const uint8_t *present, u64 n, fr_t *out) {
    if (!is_pow2(n)) return KO_ERR_NOT_POW2;
    if (n > fs->max_width) return KO_ERR_TOO_WIDE;
    u64 *missing = malloc(n * sizeof(u64)), nm = 0;
    for (u64 i = 0; i < n; i++) if (!present[i]) missing[nm++] = i;
    fr_t *zeval = malloc(n * sizeof(fr_t)), *zpoly = malloc(n * sizeof(fr_t)), *e = calloc(n, sizeof(fr_t)), *a = malloc(n * sizeof(fr_t)),
         *b = malloc(n * sizeof(fr_t)), *c = malloc(n * sizeof(fr_t));
    int st = ko_zero_poly_via_multiplication(fs, missing, nm, n, zeval, zpoly);
    if (!st) {
        for (u64 i = 0; i < n; i++) if (present[i]) fr_mul(&e[i], &samples[i], &zeval[i]);
        ko_inplace_fft(fs, e, a, n, 1);                  /* polyWithZero */
        shift_poly(a, n, 0); shift_poly(zpoly, n, 0);
        ko_inplace_fft(fs, a, b, n, 0);                  /* evalShiftedPolyWithZero */
        ko_inplace_fft(fs, zpoly, c, n, 0);              /* evalShiftedZeroPoly */
        for (u64 i = 0; i < n; i++) { fr_t inv; fr_inv(&inv, &c[i]); fr_mul(&b[i], &inv, &b[i]); }   /* DivModFr */
        ko_inplace_fft(fs, b, a, n, 1);                  /* shiftedReconstructedPoly */
        shift_poly(a, n, 1);
        ko_inplace_fft(fs, a, out, n, 0);                /* reconstructedData */
        for (u64 i = 0; i < n; i++) if (present[i] && !fr_eq(&out[i], &samples[i])) { st = KO_ERR_BAD_ARG; break; }
    }
    free(missing); free(zeval); free(zpoly); free(e); free(a); free(b); free(c);
    return st;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs: splitmix64 stream -> uniform field elements, Montgomery form
 * ---------------------------------------------------------------------------------------------- */
API void ko_synthetic_blob(u64 seed, u64 n, fr_t *out) {
    u64 state = seed;
    for (u64 i = 0; i < n; i++) {
        u64 w[8] = {0};
        for (int k = 0; k < 4; k++) {
            state += 0x9E3779B97F4A7C15ULL; u64 z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; w[k] = z ^ (z >> 31);
        }
        /* reduce the 256-bit value mod r: at most 4 conditional subtractions of r<<k would be needed; 2^256 < 3r... use repeated subtraction (value < 2^256 < 2.3 r... r ~ 0.45 * 2^256) */
        fr_t v; memcpy(&v, w, 32);
        while (limbs_geq(v.l, FR_R.l, 4)) limbs_sub(v.l, v.l, FR_R.l, 4);
        fr_to_mont(&out[i], &v);
    }
}
