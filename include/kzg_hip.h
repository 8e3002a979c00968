/*
 * kzg_hip.h -- C ABI of libkzg_hip.so: the MI355X (gfx950) implementation of go-kzg's commitment /
 * proof hot path.  Each entry point replaces one batch-level method of the reference's Go API
 * (file:line relative to protolambda/go-kzg); the cgo binding a maintainer would add is in
 * INTEGRATION.md and go-kzg_amd/goshim/.
 *
 * Data crossing the boundary uses the memory images of the reference's default (Kilic) backend, so Go
 * slices are passed zero-copy:
 *   Fr  : 32 B  = 4 x u64 little-endian limbs, Montgomery form (R = 2^256 mod r)      bls/bignum_kilic.go:21-23
 *   G1  : 144 B = 3 x 6 x u64 (X, Y, Z) Jacobian, Montgomery (R = 2^384 mod p), inf <=> Z == 0
 *                                                                                     bls/bls_kilic.go:30-35
 * Returned points are normalised: Z == R (affine) or Kilic's infinity image (0, R, 0), so that byte
 * comparison with any correct backend is meaningful.  Callers own all buffers; the library keeps no
 * caller pointer after a call returns (cgo rule).  All calls are blocking and thread-safe per handle.
 * Concurrent ONE-polynomial calls on a handle (kzg_hip_commit_to_poly, _compute_proof_single, _da_using_fk20, _da_using_fk20_multi,
 * _lincomb_points, _eth_blob_to_kzg_commitment_batch with batch == 1, _eth_compute_kzg_proof) are merged into batched launches (go-kzg_amd/csrc/coalesce.hpp; KZG_HIP_COALESCE=0
 * turns that off); a lone caller is not delayed.
 *
 * Functions ending in _dev take DEVICE pointers (HBM-resident inputs/outputs) and a hipStream_t passed
 * as void*; they enqueue work and return without synchronising the caller's stream.  (The FIRST commitment on a settings object builds
 * its fixed-base table: that call blocks its host thread until the build -- on the handle's own stream -- is done; the caller's stream is
 * not drained.)
 */
#ifndef KZG_HIP_H
#define KZG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* status codes.  The Go shim maps 1-2 to `error` (fft_fr.go:57-59,78-83; fft_g1.go:60-65) and 3-6 to panic
 * (bls/bls_kilic.go:133-135; kzg.go:22-27,44-52,74-91; fk20_single.go:60-62,140-154; fk20_multi.go:28-31,60-69). */
#define KZG_HIP_OK 0
#define KZG_HIP_ERR_TOO_WIDE 1      /* more values than roots of unity */
#define KZG_HIP_ERR_NOT_POW2 2      /* length is not a power of two */
#define KZG_HIP_ERR_LEN_MISMATCH 3  /* slice lengths do not match / setup too short */
#define KZG_HIP_ERR_UPPER_HALF 4    /* "bad input, second half should be zeroed" */
#define KZG_HIP_ERR_BAD_ARG 5       /* other misuse (n < 2, chunk length, NULL, ...) */
#define KZG_HIP_ERR_BAD_POINT 6     /* invalid compressed G1 */
#define KZG_HIP_ERR_NO_DEVICE 7     /* no gfx950 device visible: the library has NO CPU fallback */
#define KZG_HIP_ERR_HIP 8           /* HIP runtime error, see kzg_hip_last_error() */
#define KZG_HIP_ERR_UNSUPPORTED 9    /* a size class this library does not serve: kzg_hip_fft_settings_new with KZG_HIP_MAX_SCALE < max_scale <= 31 */
#define KZG_HIP_ERR_RECOVERY 10      /* "failed to reconstruct data correctly" (recover_from_samples.go:103-107) */
#define KZG_HIP_ERR_BAD_BLOB 11      /* "could not convert blobs to polynomials" (eth/eth.go:156-159,176-179): a field element >= r */

typedef struct kzg_hip_fft kzg_hip_fft;                 /* *kzg.FFTSettings         fft.go:34-42   */
typedef struct kzg_hip_kzg kzg_hip_kzg;                 /* *kzg.KZGSettings         kzg.go:11-19   */
typedef struct kzg_hip_fk20s kzg_hip_fk20s; /* *kzg.FK20SingleSettings  kzg.go:38-41   */
typedef struct kzg_hip_fk20m kzg_hip_fk20m; /* *kzg.FK20MultiSettings   kzg.go:66-71   */

/* ---- library / device ---- */
int kzg_hip_device_count(void);                 /* number of usable gfx950 devices (0 when none)            */
const char *kzg_hip_last_error(void);           /* thread-local text of the last KZG_HIP_ERR_HIP            */
const char *kzg_hip_version(void);

/* Pinning of caller memory (optional).  kzg_hip_commit_to_poly_batch and kzg_hip_eth_blob_to_kzg_commitment_batch (and their multi-device forms) check whether
 * their INPUT lies in pinned memory and then read the coefficients in place over PCIe instead of staging a copy of pageable memory: 67-78 k -> ~95 k commitments/s from host
 * buffers on one GPU.  The range stays pinned (and visible to every device) until kzg_hip_host_unregister; a Go caller pins the slice first (runtime.Pinner). */
int kzg_hip_host_register(void *host, uint64_t bytes);
int kzg_hip_host_unregister(void *host);

/* ---- FFTSettings: NewFFTSettings (fft.go:44-61) ----
 * max_scale <= KZG_HIP_MAX_SCALE (2^24 roots: 6 root tables of 0.5 GB + the G1 twiddles' digit rows, 8.9 GB); the reference's root table goes to
 * scale 31 (bls/globals.go:27-60), whose tables would not fit any device: KZG_HIP_MAX_SCALE < max_scale <= 31 returns KZG_HIP_ERR_UNSUPPORTED, larger
 * values KZG_HIP_ERR_BAD_ARG (the reference indexes past its table there).  Transforms are parity-tested up to 2^20 points
 * (tests/test_gpu_parity.py::test_fft_fr_above_65536). */
#define KZG_HIP_MAX_SCALE 24
int kzg_hip_fft_settings_new(int device, unsigned max_scale, kzg_hip_fft **out);
void kzg_hip_fft_settings_free(kzg_hip_fft *fs);
uint64_t kzg_hip_fft_max_width(const kzg_hip_fft *fs);
/* copies ExpandedRootsOfUnity (reversed == 0) or ReverseRootsOfUnity (reversed != 0): max_width + 1 Fr */
int kzg_hip_fft_roots(const kzg_hip_fft *fs, int reversed, void *out_fr);

/* FFTSettings.FFT (fft_fr.go:55-74): zero-pads n to the next power of two np; out must hold np Fr; *out_n = np */
int kzg_hip_fft_fr(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, int inv, void *out_fr, uint64_t *out_n);
/* FFTSettings.InplaceFFT (fft_fr.go:76-105): n must be a power of two, out != vals */
int kzg_hip_inplace_fft_fr(kzg_hip_fft *fs, const void *vals_fr, void *out_fr, uint64_t n, int inv);
/* `batch` independent transforms of n (power of two) contiguous values each */
int kzg_hip_fft_fr_batch(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, uint64_t batch, int inv, void *out_fr);
/* FFTSettings.FFTG1 (fft_g1.go:58-94) */
int kzg_hip_fft_g1(kzg_hip_fft *fs, const void *vals_g1, uint64_t n, int inv, void *out_g1);
/* FFTSettings.DASFFTExtension (das_extension.go:71-84): in place, n even-index values -> n odd-index values */
int kzg_hip_das_fft_extension(kzg_hip_fft *fs, void *vals_fr, uint64_t n);
int kzg_hip_das_fft_extension_batch(kzg_hip_fft *fs, void *vals_fr, uint64_t n, uint64_t batch);

/* device-resident batch forms of the three transforms the reference publishes benchmarks for (BENCH.md:31,43,55):
 * `batch` rows of n values each, inputs and outputs in HBM, Kilic images for G1 (converted and normalised inside).  d_out must not
 * overlap d_vals (InplaceFFT's rule, fft_fr.go:76: the name notwithstanding it takes a separate output slice). */
int kzg_hip_fft_fr_batch_dev(kzg_hip_fft *fs, const void *d_vals_fr, uint64_t n, uint64_t batch, int inv, void *d_out_fr, void *stream);
int kzg_hip_fft_g1_batch(kzg_hip_fft *fs, const void *vals_g1, uint64_t n, uint64_t batch, int inv, void *out_g1);   /* FFTG1 on `batch` rows of n points, host buffers */
int kzg_hip_fft_g1_batch_dev(kzg_hip_fft *fs, const void *d_vals_g1, uint64_t n, uint64_t batch, int inv, void *d_out_g1, void *stream);
int kzg_hip_das_fft_extension_batch_dev(kzg_hip_fft *fs, void *d_vals_fr, uint64_t n, uint64_t batch, void *stream);

/* bls.FrFrom32 / bls.FrTo32 over a slice (bls/bignum_kilic.go:33-55; range rule bls.ValidFr, bls/bignum_all.go:12-35):
 * n x 32 little-endian bytes <-> Montgomery images.  from: *all_ok = 0 if any value is >= r (those become 0). */
int kzg_hip_fr_from_le32(kzg_hip_fft *fs, const void *in_le32, uint64_t n, void *out_fr, int *all_ok);
int kzg_hip_fr_to_le32(kzg_hip_fft *fs, const void *in_fr, uint64_t n, void *out_le32);

/* ---- bls.LinCombG1 (bls/bls_kilic.go:132-150): Pippenger MSM; n == 0 -> infinity ---- */
int kzg_hip_lincomb_g1(kzg_hip_fft *fs, const void *points_g1, const void *scalars_fr, uint64_t n, void *out_g1);
/* Cached point set for repeated LinCombG1 on the SAME points (CommitToEvalPoly's IFFT of the setup, kzg_single_proofs.go:12-14;
 * the Lagrange setup of eth/helpers.go:99,159,199): uploaded and converted once, resident in HBM with its 2^64 multiples.
 * kzg_hip_lincomb_points[_batch]: out[b] = LinCombG1(points[:n], scalars[b]); n <= the set's size (KZG_HIP_ERR_LEN_MISMATCH
 * otherwise, the reference's length-mismatch panic); n == 0 -> infinity.  The _dev form takes device pointers and a stream. */
typedef struct kzg_hip_points kzg_hip_points;
int kzg_hip_points_new(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_points **out);
void kzg_hip_points_free(kzg_hip_points *pts);
/* A cached set of >= 64 points also gets a fixed-base table like a KZGSettings object (built by its first combination): budget in GB, default
 * KZG_HIP_POINTS_FB_BUDGET_GB or min(32 GB, free HBM - 24 GB) at creation; 0 keeps the set on the bucket pipeline.  Results are identical. */
int kzg_hip_points_set_table_budget_gb(kzg_hip_points *pts, double gb);
uint64_t kzg_hip_points_count(const kzg_hip_points *pts);
int kzg_hip_lincomb_points(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1);
int kzg_hip_lincomb_points_batch(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_lincomb_points_batch_dev(kzg_hip_points *pts, const void *d_scalars_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream);
/* bls.ToCompressedG1 over a slice (bls/bls_kilic.go:114-116): n points -> n x 48 B ZCash form */
int kzg_hip_g1_to_compressed(kzg_hip_fft *fs, const void *points_g1, uint64_t n, void *out48);
/* bls.FromCompressedG1 over a slice (bls/bls_kilic.go:118-121) */
int kzg_hip_g1_from_compressed(kzg_hip_fft *fs, const void *in48, uint64_t n, void *out_g1);
/* element-wise bls.MulG1 (bls/bls_kilic.go:41-45): out[i] = scalars[i] * points[i] */
int kzg_hip_g1_mul_vec(kzg_hip_fft *fs, const void *points_g1, const void *scalars_fr, uint64_t n, void *out_g1);
/* GenerateTestingSetup, G1 half (setup.go:9-26): out[i] = [secret^i] G1 */
int kzg_hip_generate_testing_setup_g1(kzg_hip_fft *fs, const void *secret_fr, uint64_t n, void *out_g1);

/* G1Point.MarshalText / UnmarshalText over a slice (bls/bls_all.go:20-39): n points <-> n x 96 lower-case hex characters (no 0x
 * prefix, no terminator).  Unmarshal returns KZG_HIP_ERR_BAD_POINT for a non-hex character or an invalid point. */
int kzg_hip_g1_marshal_text(kzg_hip_fft *fs, const void *points_g1, uint64_t n, char *out_hex96);
int kzg_hip_g1_unmarshal_text(kzg_hip_fft *fs, const char *hex96, uint64_t n, void *out_g1);
/* JSONTrustedSetup (eth/globals.go:33-49): decodes the "setup_G1" and "setup_G1_lagrange" arrays of a trusted-setup JSON
 * document (the format of eth/trusted_setup.json) into Kilic images, decompressing and subgroup-checking on the device.  Other
 * keys ("setup_G2", "roots_of_unity") are skipped: G2 stays with the CPU backend.  Counts are always written; the arrays only when
 * the out pointer is non-NULL (call once with NULL to size the buffers).  `capacity` = points each out array can hold. */
int kzg_hip_trusted_setup_from_json(kzg_hip_fft *fs, const char *json, uint64_t json_len, void *out_setup_g1, void *out_lagrange_g1,
                                    uint64_t capacity, uint64_t *n_setup_g1, uint64_t *n_lagrange_g1);

/* ---- KZGSettings: NewKZGSettings, prover side (kzg.go:21-36); the setup is uploaded once and stays in HBM ---- */
int kzg_hip_kzg_settings_new(kzg_hip_fft *fs, const void *secret_g1, uint64_t n_setup, kzg_hip_kzg **out);
void kzg_hip_kzg_settings_free(kzg_hip_kzg *ks);
/* HBM budget (GB, decimal) of the fixed-base commitment table of this settings object; call before the first commitment (a
 * later call frees the table, which is rebuilt lazily).  Default without this call: KZG_HIP_FB_BUDGET_GB, else 110 GB
 * clipped to free HBM - 24 GB (n = 4096: signed 16-bit windows, 8 of them walked by both GLV halves of a scalar: 103 GB, 16 additions per
 * coefficient); 60 selects 15-bit windows (58 GB, 18 additions), 17 selects 13-bit windows (16 GB, 20 additions). */
int kzg_hip_kzg_set_table_budget_gb(kzg_hip_kzg *ks, double gb);
/* KZGSettings.CommitToPoly (kzg_single_proofs.go:17-19) */
int kzg_hip_commit_to_poly(kzg_hip_kzg *ks, const void *coeffs_fr, uint64_t n, void *out_g1);
/* `batch` polynomials of n coefficients each -> `batch` commitments */
int kzg_hip_commit_to_poly_batch(kzg_hip_kzg *ks, const void *coeffs_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_commit_to_poly_batch_dev(kzg_hip_kzg *ks, const void *d_coeffs_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream);
/* KZGSettings.ComputeProofSingle (kzg_single_proofs.go:36-54; poly.go:14-40): x is a uint64 as in the reference */
int kzg_hip_compute_proof_single(kzg_hip_kzg *ks, const void *poly_fr, uint64_t n, uint64_t x, void *out_g1);
/* `batch` polynomials of n coefficients each, evaluation point xs[b] for polynomial b -> `batch` proofs; the _dev form takes
 * device pointers (polynomials, uint64 xs, outputs) and a stream */
int kzg_hip_compute_proof_single_batch(kzg_hip_kzg *ks, const void *poly_fr, uint64_t n, uint64_t batch, const uint64_t *xs, void *out_g1);
int kzg_hip_compute_proof_single_batch_dev(kzg_hip_kzg *ks, const void *d_poly_fr, uint64_t n, uint64_t batch, const void *d_x_u64, void *d_out_g1,
                                           void *stream);
/* KZGSettings.ToeplitzPart2 / ToeplitzPart3 (fk20_single.go:59-87); part3 writes n/2 points */
int kzg_hip_toeplitz_part2(kzg_hip_kzg *ks, const void *coeffs_fr, const void *x_ext_fft_g1, uint64_t n, void *out_g1);
int kzg_hip_toeplitz_part3(kzg_hip_kzg *ks, const void *h_ext_fft_g1, uint64_t n, void *out_g1);

/* ---- KZG multi proofs, prover side (kzg_multi_proofs.go) ----
 * KZGSettings.ComputeProofMulti (kzg_multi_proofs.go:13-43).  The reference never initialises xPowN (:20-24), so its divisor is
 * X^n rather than X^n - x^n and the quotient is poly[n:]; reproduced faithfully (the result is a valid proof while len <= 2n). */
int kzg_hip_compute_proof_multi(kzg_hip_kzg *ks, const void *poly_fr, uint64_t len, uint64_t x, uint64_t n, void *out_g1);
/* prover-side half of CheckProofMulti (kzg_multi_proofs.go:47-75): commitment to the interpolation polynomial on x * <w_n>
 * ([I(s)]_1 = LinCombG1(SecretG1, IFFT(ys)_i / x^i)) and x^n; the pairing stays with the verifier's CPU backend. */
int kzg_hip_check_proof_multi_interpolation(kzg_hip_kzg *ks, const void *ys_fr, uint64_t n, const void *x_fr, void *out_is1_g1, void *out_xpow_fr);

/* ---- FK20 single (kzg.go:43-64; fk20_single.go:122-196) ---- */
int kzg_hip_fk20_single_settings_new(kzg_hip_kzg *ks, uint64_t n2, kzg_hip_fk20s **out);
void kzg_hip_fk20_single_settings_free(kzg_hip_fk20s *fk);
int kzg_hip_fk20_single_x_ext_fft(const kzg_hip_fk20s *fk, void *out_g1 /* n2 points */);
int kzg_hip_fk20_single(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, void *out_g1 /* n */);
int kzg_hip_fk20_single_da_optimized(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n2, void *out_g1 /* n2 */);
/* `batch` polynomials of n coefficients -> batch x n proofs (FK20Single on each; BASELINE config 4b: n = 4096 at scale 13) */
int kzg_hip_fk20_single_batch(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_fk20_single_batch_dev(kzg_hip_fk20s *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream);
int kzg_hip_da_using_fk20(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, void *out_g1 /* 2n */);
/* `batch` polynomials of n coefficients -> batch x 2n proofs (DAUsingFK20 on each) */
int kzg_hip_da_using_fk20_batch(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_da_using_fk20_batch_dev(kzg_hip_fk20s *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream);

/* ---- FK20 multi (kzg.go:73-116; fk20_multi.go:25-133) ---- */
int kzg_hip_fk20_multi_settings_new(kzg_hip_kzg *ks, uint64_t n2, uint64_t chunk_len, kzg_hip_fk20m **out);
void kzg_hip_fk20_multi_settings_free(kzg_hip_fk20m *fk);
int kzg_hip_fk20_multi(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, void *out_g1 /* n / chunk_len */);
int kzg_hip_fk20_multi_da_optimized(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n2, void *out_g1 /* n2 / chunk_len */);
int kzg_hip_da_using_fk20_multi(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, void *out_g1 /* 2n / chunk_len */);
/* `batch` polynomials of n coefficients -> batch x 2n / chunk_len proofs (DAUsingFK20Multi on each), host buffers */
int kzg_hip_da_using_fk20_multi_batch(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_da_using_fk20_multi_batch_dev(kzg_hip_fk20m *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream);
/* Sharded form for one process per GPU: this rank computes hExtFFT for output positions
 * [j0, j0 + cnt) only and writes cnt points as OPAQUE 144-byte device-internal Jacobian images (not Kilic images: the
 * device keeps F_p in Montgomery radix 2^390); ranks all-gather the slices (RCCL, bytes) and call
 * kzg_hip_fk20_multi_finish_dev on the gathered 2k points, which returns normalised Kilic images. */
int kzg_hip_fk20_multi_hext_slice_dev(kzg_hip_fk20m *fk, const void *d_poly_fr, uint64_t n, uint64_t j0, uint64_t cnt, void *d_out_g1, void *stream);
int kzg_hip_fk20_multi_finish_dev(kzg_hip_fk20m *fk, const void *d_hext_g1, int bit_reverse, void *d_out_g1, void *stream);

/* ---- eth/ byte-level prover path: eth/globals.go:39-72, eth/eth.go:145-151, eth/helpers.go:98-103,179-203,264-273 ----
 * kzg_hip_eth_settings_new takes setup_G1_lagrange in NATURAL order (as eth/trusted_setup.json stores it) and applies the
 * bit-reversal permutation of eth/globals.go:48 itself; DomainFr is the bit-reversed scale-log2(n) domain (:61-66). */
typedef struct kzg_hip_eth kzg_hip_eth;
int kzg_hip_eth_settings_new(kzg_hip_fft *fs, const void *lagrange_g1, uint64_t n, kzg_hip_eth **out);
void kzg_hip_eth_settings_free(kzg_hip_eth *eth);
/* BlobToKZGCommitment over `batch` blobs of n x 32 little-endian bytes: out48[b] = commitment, ok[b] = 1, or ok[b] = 0 when a
 * field element is >= r (the reference returns (KZGCommitment{}, false): out48[b] is zeroed). */
int kzg_hip_eth_blob_to_kzg_commitment_batch(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out48, uint8_t *ok);
/* (batch == 1 is eth.BlobToKZGCommitment itself, eth/eth.go:145-151: concurrent one-blob calls on a handle are coalesced into
 * batched launches like kzg_hip_commit_to_poly's.) */
/* ComputeKZGProof (eth/helpers.go:179-203): polynomial in evaluation form (n Fr), z; writes the 48-byte proof and (optionally) y.
 * KZG_HIP_ERR_LEN_MISMATCH: "polynomial has invalid length"; KZG_HIP_ERR_BAD_ARG: "invalid z challenge" (z in the domain). */
int kzg_hip_eth_compute_kzg_proof(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *z_fr, void *out48, void *y_fr);
/* (one polynomial per call, like the reference: concurrent calls on a handle are coalesced into batched launches.)
 * The same over `batch` rows (polys_fr: batch x n Fr, zs_fr: batch Fr): out48[b] / ys_fr[b] (optional) and ok[b] = 1, or ok[b] = 0 with
 * out48[b] and ys_fr[b] zeroed where zs[b] lies in the domain (the reference's "invalid z challenge" for that row).  One launch chain for
 * the whole batch, no host round trip between the quotients and their commitment. */
int kzg_hip_eth_compute_kzg_proof_batch(kzg_hip_eth *eth, const void *polys_fr, uint64_t n, uint64_t batch, const void *zs_fr, void *out48, void *ys_fr, uint8_t *ok);
/* device-resident form: every pointer is device memory, work is enqueued on `stream`; d_bad_u32[b] != 0 marks an invalid z (that row's
 * proof bytes are then those of the point at infinity and must be ignored); d_ys_fr may be null */
int kzg_hip_eth_compute_kzg_proof_batch_dev(kzg_hip_eth *eth, const void *d_polys_fr, uint64_t n, uint64_t batch, const void *d_zs_fr, void *d_out48, void *d_ys_fr,
                                            void *d_bad_u32, void *stream);

/* eth.ComputeAggregateKZGProof (eth/eth.go:175-182 -> ComputeAggregateKZGProofFromPolynomials, eth/helpers.go:165-176): the blobs of one block
 * (batch x n x 32 little-endian bytes; batch == 0 is valid: the proof of the zero polynomial) -> the 48-byte aggregated proof.  On the device: the
 * polynomials, their commitments, the aggregated polynomial (bls.PolyLinComb) and its proof at the evaluation challenge; on the host, while the
 * device commits: the Fiat-Shamir transcript (hashPolysComms / hashToBLSField: one SHA-256 chain over all blobs).  out_commitments48 (optional,
 * batch x 48) receives the blobs' commitments, which the caller needs for the block anyway.
 * KZG_HIP_ERR_BAD_BLOB: "could not convert blobs to polynomials"; KZG_HIP_ERR_BAD_ARG: "invalid z challenge". */
int kzg_hip_eth_compute_aggregate_kzg_proof(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out_proof48, void *out_commitments48);
/* The prover-side pieces of eth.VerifyAggregateKZGProof (eth/eth.go:155-172): ComputeAggregatedPolyAndCommitment (eth/helpers.go:137-162) over the
 * blobs and the EXPECTED commitments (batch x 48), then y = EvaluatePolynomialInEvaluationForm(aggregatedPoly, z).  Writes the aggregated
 * commitment (one G1 Kilic image), z, y (optional) and the aggregated polynomial (n Fr, optional); the pairing check of
 * VerifyKZGProofFromPoints (eth/helpers.go:55-68) stays with the caller.  KZG_HIP_ERR_BAD_BLOB as above; KZG_HIP_ERR_BAD_POINT: a commitment does
 * not decode (:153-156).  z inside the domain (probability 2^-243): y = 0, what the reference's formula returns there (see below). */
int kzg_hip_eth_compute_aggregated_poly_and_commitment(kzg_hip_eth *eth, const void *blobs_le32, const void *commitments48, uint64_t batch, void *out_poly_fr,
                                                       void *out_commitment_g1, void *out_z_fr, void *out_y_fr);
/* bls.EvaluatePolyInEvaluationForm(y, poly, x, fs.ExpandedRootsOfUnity[:fs.MaxWidth], scale) (bls/globals.go:106-153, as called in
 * fft_fr_test.go:73-99): poly[i] = f(w^(i << scale)), n = MaxWidth >> scale (else KZG_HIP_ERR_LEN_MISMATCH: the reference's panic), barycentric
 * evaluation at x.  x one of the roots: y = 0 like the reference, whose last factor (x^n - 1) / n vanishes there whatever its batch inversion made of
 * the zero denominator (bls/globals.go:141-152) -- NOT f(x). */
int kzg_hip_evaluate_poly_in_evaluation_form(kzg_hip_fft *fs, const void *poly_fr, uint64_t n, const void *x_fr, uint32_t scale, void *out_y_fr);
/* eth.EvaluatePolynomialInEvaluationForm (eth/helpers.go:207-211): the same on DomainFr (bit-reversed order) */
int kzg_hip_eth_evaluate_polynomial_in_evaluation_form(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *x_fr, void *out_y_fr);
/* ---- erasure recovery ----
 * FFTSettings.ZeroPolyViaMultiplication (zero_poly.go:116-217): vanishing polynomial of the missing indices of a size-`length`
 * domain; writes `length` evaluations and `length` coefficients (zero-padded).  No missing index -> all zeros (:117-119). */
int kzg_hip_zero_poly_via_multiplication(kzg_hip_fft *fs, const uint64_t *missing_indices, uint64_t n_missing, uint64_t length,
                                         void *out_zero_eval_fr, void *out_zero_poly_fr);
/* FFTSettings.RecoverPolyFromSamples (recover_from_samples.go:42-109) with ZeroPolyViaMultiplication as the zero-poly function:
 * n samples, present[i] == 0 <=> samples[i] == nil; writes the n reconstructed values.  KZG_HIP_ERR_RECOVERY when a known
 * sample is not reproduced (the reference's error). */
int kzg_hip_recover_poly_from_samples(kzg_hip_fft *fs, const void *samples_fr, const uint8_t *present, uint64_t n, void *out_fr);

/* ---- several GPUs behind ONE handle (SURVEY.md 8b threading row: "multi-GPU handle owns one context per device"; 8e) ----
 * The reference is a single-process library (kzg.go:11-19): a drop-in that uses every GPU of a node does so inside this library.
 * kzg_hip_multi_settings_new builds an FFTSettings (NewFFTSettings(max_scale), fft.go:44) and a KZGSettings (NewKZGSettings, kzg.go:21-36)
 * with their tables on EVERY listed device.  A list may repeat a device (each entry gets its own settings and tables: how a 1-GPU box
 * exercises the multi-device code paths).
 *   _batch calls: the polynomials are divided into contiguous shares, one per device, results land in the caller's buffer in input order;
 *     no data-path collective (blobs are independent).
 *   kzg_hip_multi_da_using_fk20 / _da_using_fk20_multi: ONE polynomial over all devices (fk20_multi.go:58-109, kzg.go:73-116): the Toeplitz
 *     stage is sharded by output position, the hExtFFT slices are ALL-GATHERED, then either the first device runs the two G1 transforms
 *     ("gather") or both transforms are sharded by decimation with two more all-gathers each, the last one of the proof points ("sharded":
 *     the five all-gathers of SURVEY.md 8e; default from 4 devices on, kzg_hip_multi_set_fft_sharding / KZG_HIP_MULTI_FFT=gather|sharded).
 * Exchange transport (kzg_hip_multi_transport): "rccl" = ncclAllGather on ncclUint8 over single-process communicators (ncclCommInitAll;
 * RCCL over xGMI), used when every listed device is distinct; "peer-copy" = hipMemcpyPeerAsync between the devices' streams, used when the
 * list repeats a device or librccl cannot be bound; "host-staged" = device -> pinned host -> device, the last resort
 * (kzg_hip_multi_transport_note says why a transport was not used).  Every buffer another device reads or writes is hipMalloc memory of a
 * per-entry exchange arena (never the stream-ordered pool), peer access is enabled between all listed devices, and the constructor PROVES
 * the transport before returning: every entry writes a pattern, one all-gather, every entry verifies every byte
 * (kzg_hip_multi_transport_check); a transport that errs or delivers wrong bytes is replaced by the next one in the order above, and the
 * constructor fails with KZG_HIP_ERR_HIP only if none delivers.  Results are identical whatever the transport and identical to the
 * single-device calls (tests/test_multi_device.py). */
typedef struct kzg_hip_multi kzg_hip_multi;
typedef struct kzg_hip_multi_fk20s kzg_hip_multi_fk20s;
typedef struct kzg_hip_multi_fk20m kzg_hip_multi_fk20m;
int kzg_hip_multi_settings_new(const int *devices, uint32_t n_devices, unsigned max_scale, const void *secret_g1, uint64_t n_setup, kzg_hip_multi **out);
void kzg_hip_multi_settings_free(kzg_hip_multi *m);
uint32_t kzg_hip_multi_device_count(const kzg_hip_multi *m);
int kzg_hip_multi_device(const kzg_hip_multi *m, uint32_t i);          /* device ordinal of entry i (-1 out of range) */
kzg_hip_fft *kzg_hip_multi_fft(kzg_hip_multi *m, uint32_t i);          /* entry i's settings, BORROWED (owned by the multi handle): */
kzg_hip_kzg *kzg_hip_multi_kzg(kzg_hip_multi *m, uint32_t i);          /* any single-device call can be made on a chosen device     */
const char *kzg_hip_multi_transport(const kzg_hip_multi *m);           /* "rccl", "peer-copy" or "host-staged" */
const char *kzg_hip_multi_transport_note(const kzg_hip_multi *m);      /* why the transport is not "rccl" ("" otherwise) */
const char *kzg_hip_multi_transport_check(const kzg_hip_multi *m); /* outcome of the creation-time exchange test: "ok: <transport>, ..." */
uint64_t kzg_hip_multi_exchanges(const kzg_hip_multi *m);              /* all-gathers performed on this handle so far */
int kzg_hip_multi_set_fft_sharding(kzg_hip_multi *m, int mode);        /* 0 gather, 1 sharded transforms, -1 default policy */
int kzg_hip_multi_set_table_budget_gb(kzg_hip_multi *m, double gb);    /* kzg_hip_kzg_set_table_budget_gb on every entry */
/* CommitToPoly / ComputeProofSingle on `batch` polynomials (kzg_single_proofs.go:17-19,36-54), sharded by polynomial */
int kzg_hip_multi_commit_to_poly_batch(kzg_hip_multi *m, const void *coeffs_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_multi_compute_proof_single_batch(kzg_hip_multi *m, const void *poly_fr, uint64_t n, uint64_t batch, const uint64_t *xs, void *out_g1);
/* FFT / InplaceFFT (fft_fr.go:55-105) and DASFFTExtension (das_extension.go:71-84; in place) on `batch` rows of n values, rows divided among the devices */
int kzg_hip_multi_fft_fr_batch(kzg_hip_multi *m, const void *vals_fr, uint64_t n, uint64_t batch, int inv, void *out_fr);
int kzg_hip_multi_das_fft_extension_batch(kzg_hip_multi *m, void *vals_fr, uint64_t n, uint64_t batch);
int kzg_hip_multi_fft_g1_batch(kzg_hip_multi *m, const void *vals_g1, uint64_t n, uint64_t batch, int inv, void *out_g1);   /* FFTG1 (fft_g1.go:58-94) likewise */
/* package eth on every entry (eth/globals.go:39-72; lagrange_g1 in natural order as for kzg_hip_eth_settings_new): BlobToKZGCommitment (eth/eth.go:145-151)
 * and ComputeKZGProof (eth/helpers.go:179-203) on batches, rows divided among the devices; flags and error codes as in the single-device calls */
typedef struct kzg_hip_multi_eth kzg_hip_multi_eth;
int kzg_hip_multi_eth_settings_new(kzg_hip_multi *m, const void *lagrange_g1, uint64_t n, kzg_hip_multi_eth **out);
void kzg_hip_multi_eth_settings_free(kzg_hip_multi_eth *eth);
int kzg_hip_multi_eth_blob_to_kzg_commitment_batch(kzg_hip_multi_eth *eth, const void *blobs_le32, uint64_t batch, void *out48, uint8_t *ok);
int kzg_hip_multi_eth_compute_kzg_proof_batch(kzg_hip_multi_eth *eth, const void *polys_fr, uint64_t n, uint64_t batch, const void *zs_fr, void *out48, void *ys_fr, uint8_t *ok);
/* NewFK20SingleSettings on every entry (kzg.go:43-64); DAUsingFK20 on a batch (sharded by polynomial) and on ONE polynomial (sharded inside) */
int kzg_hip_multi_fk20_single_settings_new(kzg_hip_multi *m, uint64_t n2, kzg_hip_multi_fk20s **out);
void kzg_hip_multi_fk20_single_settings_free(kzg_hip_multi_fk20s *fk);
int kzg_hip_multi_da_using_fk20_batch(kzg_hip_multi_fk20s *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_multi_da_using_fk20(kzg_hip_multi_fk20s *fk, const void *poly_fr, uint64_t n, void *out_g1 /* 2n */);
/* NewFK20MultiSettings on every entry (kzg.go:73-116); DAUsingFK20Multi likewise */
int kzg_hip_multi_fk20_multi_settings_new(kzg_hip_multi *m, uint64_t n2, uint64_t chunk_len, kzg_hip_multi_fk20m **out);
void kzg_hip_multi_fk20_multi_settings_free(kzg_hip_multi_fk20m *fk);
int kzg_hip_multi_da_using_fk20_multi_batch(kzg_hip_multi_fk20m *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1);
int kzg_hip_multi_da_using_fk20_multi(kzg_hip_multi_fk20m *fk, const void *poly_fr, uint64_t n, void *out_g1 /* 2n / chunk_len */);

/* shape of the fixed-base table CommitToPoly walks (built lazily by the first commitment): signed window bits c, window count and
 * bytes of HBM; all zero before the first commitment or when the setup is too small for a table (classic bucket path).  Both GLV halves of a
 * scalar (k = k1 + k2 lambda, |k1|, |k2| < 2^127) walk the same rows, so `windows` = ceil(128 / c) and a coefficient costs 2 x windows mixed
 * additions (kzg_hip_kzg_table_additions): 4096 points at c = 16 are 8 windows = 103 GB, the default budget (110 GB; kzg_hip_kzg_set_table_budget_gb). */
int kzg_hip_kzg_table_info(kzg_hip_kzg *ks, uint32_t *window_bits, uint32_t *windows, uint64_t *table_bytes);
uint32_t kzg_hip_kzg_table_additions(kzg_hip_kzg *ks);
/* Projective outputs (off by default).  The reference's G1Point is a Jacobian triple and CommitToPoly / ComputeProofSingle return it with whatever Z the
 * additions left (kzg_single_proofs.go:17-19,36-54; bls/bls_kilic.go:30-35); this library normalises every result to Z = one, one F_p inversion per result:
 * ~0.11 ms of latency, a third of a lone CommitToPoly.  on != 0: CommitToPoly / ComputeProofSingle on this settings object (single, batch, _dev forms) return the
 * same group element as an un-normalised Jacobian image (Z != one; infinity stays (0, 1, 0)); callers compare with bls.EqualG1 / compress as they would the reference's. */
int kzg_hip_kzg_set_projective_outputs(kzg_hip_kzg *ks, int on);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
