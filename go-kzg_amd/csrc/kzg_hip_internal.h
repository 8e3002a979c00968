/*
 * kzg_hip_internal.h -- instrumentation exports of libkzg_hip.so that are NOT part of the drop-in boundary (include/kzg_hip.h): the hooks
 * bench.py, tools/ and the tests use to measure and to check the library from outside.  A Go / cgo caller never needs them; they are exported
 * (default visibility) only so that ctypes can reach them.  Implemented in capi_bench.hip (and capi_eth.hip for the SHA-256 hook).
 */
#ifndef KZG_HIP_INTERNAL_H
#define KZG_HIP_INTERNAL_H
#include "../../include/kzg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* ---- roofline legs of bench.py: HIP-event time of the dominant kernel since the last reset (ms) and launch count ---- */
void kzg_hip_prof_reset(kzg_hip_fft *fs, int enable);
int kzg_hip_prof_read(kzg_hip_fft *fs, const char *kernel, double *total_ms, uint64_t *launches);
/* live calibration for the roofline: lane-operations per second of v_mad_u64_u32 and v_add_u32 and lazy 13-limb F_p products per
 * second (8 resident waves per SIMD, independent chains) on the handle's device */
int kzg_hip_calibrate(kzg_hip_fft *fs, double *mad_per_s, double *add_per_s, double *fp_mul_per_s);
/* drop-in measurement: `threads` host threads each make `calls` blocking one-polynomial calls (op 0: kzg_hip_commit_to_poly,
 * op 1: kzg_hip_compute_proof_single) on host buffers taken round-robin from blobs_fr (nblobs x n Fr); out_g1 holds `threads`
 * points (each thread's last result); *seconds = wall time from the common start to the last return */
int kzg_hip_bench_drop_in(kzg_hip_kzg *ks, int op, const void *blobs_fr, uint64_t n, uint64_t nblobs, unsigned threads, unsigned calls, void *out_g1,
                          double *seconds);
/* the same for kzg_hip_eth_compute_kzg_proof: thread t evaluates at z = 5 + t; out48 holds `threads` proofs (each thread's last) */
int kzg_hip_bench_drop_in_eth_proof(kzg_hip_eth *eth, const void *polys_fr, uint64_t n, uint64_t npolys, unsigned threads, unsigned calls, void *out48,
                                    double *seconds);
/* and for kzg_hip_fft_fr on host buffers (thread t transforms row t % nrows of vals_fr, nrows x n Fr; out_fr: threads x n Fr): the
 * per-handle stream pool at work */
int kzg_hip_bench_threads_fft_fr(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, uint64_t nrows, unsigned threads, unsigned calls, void *out_fr, double *seconds);
/* bls.PolyLinComb over device-resident rows (bls/globals.go:155-178): out[i] = sum_c scalars[c] * vectors[c * stride + i]; bench.py forms the random
 * linear combination of a timed step's input polynomials with it when it checks ALL outputs of the step */
int kzg_hip_bench_poly_lincomb_dev(kzg_hip_fft *fs, const void *d_vectors_fr, uint64_t stride, const void *d_scalars_fr, uint64_t count, uint64_t n, void *d_out_fr,
                                   void *stream);
/* statistics of the request coalescer behind the one-polynomial calls of a KZG handle (op 0: CommitToPoly, 1: ComputeProofSingle), cumulative since its first call:
 * out[0..7] = requests, batches, ns executing, ns waiting for a device slot, ns gathering callers, ns waiting for row copies, largest concurrency estimate, batches
 * allowed in flight; all zero before the first call.  bench.py prints the 256-caller run's figures with it (drop_in.coalescer_256) */
int kzg_hip_coalesce_stats(kzg_hip_kzg *ks, int op, uint64_t out[8]);
/* test / measurement hook for the F_p inversion (device-internal Montgomery images, 48 bytes each, host buffers): element i is inverted by wavefront i cooperatively
 * (coop_inv.hpp) into out_coop and by one lane alone (inv<FpP>, field.hpp) into out_lane; either output may be null.  *ms_coop / *ms_lane (nullable): HIP-event time of
 * the respective launch */
int kzg_hip_test_fp_inv(kzg_hip_fft *fs, const void *in_fp, uint64_t n, void *out_coop, void *out_lane, double *ms_coop, double *ms_lane);
/* kzg_hip_lincomb_g1's promotion of repeated caller-supplied point sets (capi_core.hip): sets promoted so far on this handle, calls served by a promoted set */
int kzg_hip_lincomb_promotions(kzg_hip_fft *fs, uint64_t *promoted, uint64_t *served);
/* the same for F_r (Kilic's Montgomery images, 32 bytes each): wave-cooperative, one lane, and the workgroup batch inversion of the eth quotient kernel (one inversion per
 * 1024 values; zero inputs are treated as one there) */
int kzg_hip_test_fr_inv(kzg_hip_fft *fs, const void *in_fr, uint64_t n, void *out_coop, void *out_lane, void *out_block);
/* test hook: SHA-256 of a host buffer through the transcript's implementation (x86 SHA extensions or the portable loop; no device needed) */
void kzg_hip_test_sha256(const void *data, uint64_t len, void *out32);


#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
