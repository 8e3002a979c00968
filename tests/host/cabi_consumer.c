/*
 * cabi_consumer.c -- a compiled, non-Python consumer of libkzg_hip.so, written the way cgo would bind it: it
 * includes ONLY include/kzg_hip.h (strict C99), links -lkzg_hip, and passes plain pointers and sizes.
 * (The reference's own precedent for a cgo-bound backend is bls/bls_hbls.go:143-149; Go itself is absent from this
 * image, so this program stands where the Go shim's compiled form would.)
 *
 * Flow (the reference's test data: kzg_single_proofs_test.go:33-64, fk20_single_test.go:11-41):
 *   NewFFTSettings -> GenerateTestingSetup -> NewKZGSettings -> CommitToPoly (vector A) -> ComputeProofSingle(x = 17) (vector B)
 *   -> NewFK20SingleSettings -> DAUsingFK20 (vector C, positions 0 / 18 / 31) -> status codes 1..6 -> package eth (aggregate proof, status 11) -> the multi-device handle on {0, 0} -> frees.
 * Expected values are SURVEY.md 8(c) vectors A-C (tests/golden/derived_vectors.json), compared as 48-byte compressed hex.
 * Test infrastructure: built and run by tests/test_cabi.py (-m gpu); prints one line per check and exits non-zero on a mismatch.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kzg_hip.h"

static int failures = 0;

static void check(int cond, const char *what) {
    printf("%s %s\n", cond ? "ok  " : "FAIL", what);
    if (!cond) failures++;
}

static void expect_status(int got, int want, const char *what) {
    char msg[256];
    snprintf(msg, sizeof msg, "%s -> status %d (want %d)", what, got, want);
    check(got == want, msg);
}

static void hex48(const uint8_t *in, char *out) {
    static const char d[] = "0123456789abcdef";
    int i;
    for (i = 0; i < 48; i++) {
        out[2 * i] = d[in[i] >> 4];
        out[2 * i + 1] = d[in[i] & 15];
    }
    out[96] = 0;
}

/* small integers -> 32 little-endian bytes each (the input form of bls.FrFrom32) */
static void le32_from_u64(uint64_t v, uint8_t *out) {
    int i;
    memset(out, 0, 32);
    for (i = 0; i < 8; i++) out[i] = (uint8_t)(v >> (8 * i));
}

/* decimal string -> 32 little-endian bytes (for the reference's test secret, setup.go / kzg_single_proofs_test.go:37) */
static void le32_from_decimal(const char *s, uint8_t *out) {
    memset(out, 0, 32);
    for (; *s; s++) {
        unsigned carry = (unsigned)(*s - '0');
        int i;
        for (i = 0; i < 32; i++) {
            unsigned v = out[i] * 10u + carry;
            out[i] = (uint8_t)v;
            carry = v >> 8;
        }
    }
}

static const char *S_TEST = "1927409816240961209460912649124";
static const uint64_t TEST_POLY[16] = {1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13};
static const char *VEC_A = "b0193d91b11e9cb43cd452fbd0e64dba26307eef309fac038987a0ebe8dd0161502e2b3a449a68869d18d01b537406b5";
static const char *VEC_B = "85a3632d390e34197ce6a037b3b2301c497eaf73e82ff59f39bf85813c49d825e2281771ad206e23a8f4fbc50174f67d";
static const char *VEC_C0 = "b6dbe2759b822dee763823dc1e84bdc0b96f570bd196181774d45692281fd479e88ca67f017568626e0eba9a829c5995";
static const char *VEC_C18 = "9265afb01340af4e02c0bdb6de9a59f54840c5e8fe1a79cefa50dfe31ac9a27d819c20f4ef6ff235dfa41f345bb04bd5";
static const char *VEC_C31 = "b57ad7ed03d0e816980124c8a609c49317085352a9bd6229feb61ca2aff1ca149b148804e21c10bcf21d118ec3f6cfb5";

#define FR 32
#define G1 144

int main(void) {
    kzg_hip_fft *fs4 = NULL, *fs5 = NULL;
    kzg_hip_kzg *ks4 = NULL, *ks5 = NULL, *ks_bad = NULL;
    kzg_hip_fk20s *fk = NULL;
    uint8_t le[16 * 32], secret_le[32], secret_fr[FR], poly[32 * FR], c48[32 * 48];
    uint8_t *setup = malloc(33 * G1), *proofs = malloc(32 * G1), *scratch = malloc(64 * G1);
    uint8_t point[G1];
    char hx[97];
    int all_ok = 0, i;

    if (!setup || !proofs || !scratch) return 2;
    printf("library: %s, devices: %d\n", kzg_hip_version(), kzg_hip_device_count());
    if (kzg_hip_device_count() < 1) {
        /* no CPU fallback: the constructors must say so */
        expect_status(kzg_hip_fft_settings_new(0, 4, &fs4), KZG_HIP_ERR_NO_DEVICE, "NewFFTSettings without a gfx950 device");
        return failures ? 1 : 77;
    }

    /* ---- settings, inputs ---- */
    expect_status(kzg_hip_fft_settings_new(0, 4, &fs4), KZG_HIP_OK, "NewFFTSettings(4)");
    expect_status(kzg_hip_fft_settings_new(0, 5, &fs5), KZG_HIP_OK, "NewFFTSettings(5)");
    check(kzg_hip_fft_max_width(fs4) == 16 && kzg_hip_fft_max_width(fs5) == 32, "MaxWidth 16 / 32");
    for (i = 0; i < 16; i++) le32_from_u64(TEST_POLY[i], le + 32 * i);
    memset(poly, 0, sizeof poly);
    expect_status(kzg_hip_fr_from_le32(fs4, le, 16, poly, &all_ok), KZG_HIP_OK, "FrFrom32 x 16");
    check(all_ok == 1, "all 16 coefficients are canonical");
    le32_from_decimal(S_TEST, secret_le);
    expect_status(kzg_hip_fr_from_le32(fs4, secret_le, 1, secret_fr, &all_ok), KZG_HIP_OK, "FrFrom32(secret)");

    /* ---- vectors A, B: CommitToPoly / ComputeProofSingle on the 16-coefficient test polynomial, 17-point setup ---- */
    expect_status(kzg_hip_generate_testing_setup_g1(fs4, secret_fr, 17, setup), KZG_HIP_OK, "GenerateTestingSetup(17)");
    expect_status(kzg_hip_kzg_settings_new(fs4, setup, 17, &ks4), KZG_HIP_OK, "NewKZGSettings");
    expect_status(kzg_hip_commit_to_poly(ks4, poly, 16, point), KZG_HIP_OK, "CommitToPoly");
    expect_status(kzg_hip_g1_to_compressed(fs4, point, 1, c48), KZG_HIP_OK, "ToCompressedG1");
    hex48(c48, hx);
    check(strcmp(hx, VEC_A) == 0, "vector A: commitment bytes");
    expect_status(kzg_hip_compute_proof_single(ks4, poly, 16, 17, point), KZG_HIP_OK, "ComputeProofSingle(x = 17)");
    kzg_hip_g1_to_compressed(fs4, point, 1, c48);
    hex48(c48, hx);
    check(strcmp(hx, VEC_B) == 0, "vector B: proof bytes");

    /* ---- vector C: DAUsingFK20 at scale 5 (16 coefficients -> 32 proofs) ---- */
    expect_status(kzg_hip_generate_testing_setup_g1(fs5, secret_fr, 33, setup), KZG_HIP_OK, "GenerateTestingSetup(33)");
    expect_status(kzg_hip_kzg_settings_new(fs5, setup, 33, &ks5), KZG_HIP_OK, "NewKZGSettings (scale 5)");
    expect_status(kzg_hip_fk20_single_settings_new(ks5, 32, &fk), KZG_HIP_OK, "NewFK20SingleSettings(32)");
    expect_status(kzg_hip_da_using_fk20(fk, poly, 16, proofs), KZG_HIP_OK, "DAUsingFK20");
    expect_status(kzg_hip_g1_to_compressed(fs5, proofs, 32, c48), KZG_HIP_OK, "ToCompressedG1 x 32");
    hex48(c48, hx);
    check(strcmp(hx, VEC_C0) == 0, "vector C: proof 0");
    hex48(c48 + 18 * 48, hx);
    check(strcmp(hx, VEC_C18) == 0, "vector C: proof 18 (position 9 of fk20_single_test.go:30-41)");
    hex48(c48 + 31 * 48, hx);
    check(strcmp(hx, VEC_C31) == 0, "vector C: proof 31");

    /* ---- status codes 1..6 (the Go shim maps 1-2 to error, 3-6 to panic) ---- */
    memset(scratch, 0, 64 * G1);
    expect_status(kzg_hip_inplace_fft_fr(fs4, scratch, scratch + 32 * FR, 32, 0), KZG_HIP_ERR_TOO_WIDE, "InplaceFFT of 32 values on 16 roots");
    expect_status(kzg_hip_inplace_fft_fr(fs4, scratch, scratch + 32 * FR, 3, 0), KZG_HIP_ERR_NOT_POW2, "InplaceFFT of 3 values");
    expect_status(kzg_hip_commit_to_poly(ks4, scratch, 18, point), KZG_HIP_ERR_LEN_MISMATCH, "CommitToPoly with more coefficients than setup points");
    expect_status(kzg_hip_kzg_settings_new(fs5, setup, 16, &ks_bad), KZG_HIP_ERR_LEN_MISMATCH, "NewKZGSettings with a setup shorter than MaxWidth");
    memcpy(scratch, poly, 16 * FR);
    memset(scratch + 16 * FR, 0, 16 * FR);
    memcpy(scratch + 20 * FR, poly + FR, FR);                       /* a non-zero value in the upper half */
    expect_status(kzg_hip_fk20_single_da_optimized(fk, scratch, 32, proofs), KZG_HIP_ERR_UPPER_HALF, "FK20SingleDAOptimized with a dirty upper half");
    expect_status(kzg_hip_compute_proof_single(ks4, poly, 1, 17, point), KZG_HIP_ERR_BAD_ARG, "ComputeProofSingle of a constant");
    expect_status(kzg_hip_fft_fr(NULL, poly, 16, 0, scratch, NULL), KZG_HIP_ERR_BAD_ARG, "FFT on a NULL handle");
    memset(c48, 0xff, 48);
    expect_status(kzg_hip_g1_from_compressed(fs4, c48, 1, point), KZG_HIP_ERR_BAD_POINT, "FromCompressedG1 of 48 x 0xff");

    /* ---- package eth on a 16-element "blob" (eth/eth.go:175-182): no blobs -> the proof of the zero polynomial; an element >= r -> status 11 ---- */
    {
        kzg_hip_eth *es = NULL;
        unsigned char *lagrange = (unsigned char *)malloc(16 * G1), blob[16 * 32], proof[48], comm[48];
        expect_status(kzg_hip_fft_g1(fs4, setup, 16, 1, lagrange), KZG_HIP_OK, "FFTG1(setup[:16], inv) = the Lagrange setup");
        expect_status(kzg_hip_eth_settings_new(fs4, lagrange, 16, &es), KZG_HIP_OK, "eth settings (16 elements per blob)");
        expect_status(kzg_hip_eth_compute_aggregate_kzg_proof(es, NULL, 0, proof, NULL), KZG_HIP_OK, "ComputeAggregateKZGProof of no blobs");
        hex48(proof, hx);
        check(strncmp(hx, "c00000", 6) == 0 && hx[95] == '0', "the proof of the zero polynomial is the point at infinity");
        memset(blob, 0, sizeof blob);
        blob[0] = 5;
        expect_status(kzg_hip_eth_compute_aggregate_kzg_proof(es, blob, 1, proof, comm), KZG_HIP_OK, "ComputeAggregateKZGProof of one blob");
        memset(blob + 32 * 7, 0xff, 32);
        expect_status(kzg_hip_eth_compute_aggregate_kzg_proof(es, blob, 1, proof, comm), KZG_HIP_ERR_BAD_BLOB, "a field element >= r in a blob");
        kzg_hip_eth_settings_free(es);
        free(lagrange);
    }

    /* ---- several devices behind one handle (kzg_hip_multi_*): the list {0, 0} -- two entries on the one GPU a test box has -- must give the
     * bytes of the single-device calls: vector A through the batch form (two polynomials divided among the entries), vector C through ONE
     * DAUsingFK20 sharded inside the library (gather scheme, then the five-all-gather scheme) ---- */
    {
        kzg_hip_multi *m = NULL;
        kzg_hip_multi_fk20s *mfk = NULL;
        int devs[2] = {0, 0}, no_such[2] = {0, 4096};
        unsigned char two_polys[2 * 16 * FR], two_out[2 * G1];
        uint64_t before;
        expect_status(kzg_hip_multi_settings_new(no_such, 2, 5, setup, 33, &m), KZG_HIP_ERR_NO_DEVICE, "NewMultiKZGSettings with a device that is not there");
        expect_status(kzg_hip_multi_settings_new(devs, 2, 5, setup, 33, &m), KZG_HIP_OK, "NewMultiKZGSettings({0, 0}, scale 5)");
        check(kzg_hip_multi_device_count(m) == 2 && kzg_hip_multi_device(m, 1) == 0, "two entries, both on device 0");
        check(strcmp(kzg_hip_multi_transport(m), "peer-copy") == 0, "a repeated device exchanges by peer copies (RCCL needs distinct devices)");
        check(strncmp(kzg_hip_multi_transport_check(m), "ok: peer-copy, 2 entries", 24) == 0, "the constructor proved the exchange (pattern, all-gather, verify)");
        memcpy(two_polys, poly, 16 * FR);
        memcpy(two_polys + 16 * FR, poly, 16 * FR);
        expect_status(kzg_hip_multi_commit_to_poly_batch(m, two_polys, 16, 2, two_out), KZG_HIP_OK, "multi CommitToPoly x 2");
        kzg_hip_g1_to_compressed(fs5, two_out, 2, c48);
        hex48(c48, hx);
        check(strcmp(hx, VEC_A) == 0, "multi: vector A from entry 0");
        hex48(c48 + 48, hx);
        check(strcmp(hx, VEC_A) == 0, "multi: vector A from entry 1");
        expect_status(kzg_hip_multi_fk20_single_settings_new(m, 32, &mfk), KZG_HIP_OK, "multi NewFK20SingleSettings(32)");
        for (i = 0; i < 2; i++) {
            expect_status(kzg_hip_multi_set_fft_sharding(m, i), KZG_HIP_OK, i ? "sharded transforms" : "gather");
            before = kzg_hip_multi_exchanges(m);
            memset(proofs, 0, 32 * G1);
            expect_status(kzg_hip_multi_da_using_fk20(mfk, poly, 16, proofs), KZG_HIP_OK, "multi DAUsingFK20 of one polynomial");
            check(kzg_hip_multi_exchanges(m) - before == (i ? 5u : 1u), i ? "five all-gathers" : "one all-gather");
            kzg_hip_g1_to_compressed(fs5, proofs, 32, c48);
            hex48(c48, hx);
            check(strcmp(hx, VEC_C0) == 0, "multi: vector C proof 0");
            hex48(c48 + 18 * 48, hx);
            check(strcmp(hx, VEC_C18) == 0, "multi: vector C proof 18");
            hex48(c48 + 31 * 48, hx);
            check(strcmp(hx, VEC_C31) == 0, "multi: vector C proof 31");
        }
        expect_status(kzg_hip_multi_da_using_fk20(mfk, poly, 8, proofs), KZG_HIP_ERR_LEN_MISMATCH, "multi DAUsingFK20 with half the coefficients");
        kzg_hip_multi_fk20_single_settings_free(mfk);
        kzg_hip_multi_settings_free(m);
    }

    /* ---- frees, dependents first ---- */
    kzg_hip_fk20_single_settings_free(fk);
    kzg_hip_kzg_settings_free(ks5);
    kzg_hip_kzg_settings_free(ks4);
    kzg_hip_fft_settings_free(fs5);
    kzg_hip_fft_settings_free(fs4);
    free(setup);
    free(proofs);
    free(scratch);
    printf("%s: %d failure(s)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}
