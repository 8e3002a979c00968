// capi_recovery.hip -- erasure recovery (f3): ZeroPolyViaMultiplication, RecoverPolyFromSamples
#include "capi_common.hpp"

// ---------------------------------------------------------------------------------------------------------
// erasure recovery (row f3)
// ---------------------------------------------------------------------------------------------------------
// Small erasure sets: the vanishing polynomial evaluated directly on the domain (k_zero_eval_direct, length x n_missing products), then one inverse
// transform for the coefficients.  Large ones: the product tree of k_fr.hip (n log^2 n products: 65 536 points with half of them missing are 2^31
// products directly and ~2^24 through the tree), then one forward transform for the evaluations.  The polynomial is unique (monic, the given roots),
// so both give the reference's values bit for bit.  KZG_HIP_ZERO_POLY=direct|tree forces one (tests run both).
static int zero_poly_tree(kzg_hip_fft *fs, hipStream_t s, const uint64_t *d_missing, uint64_t n_missing, uint64_t length, fr *d_eval, fr *d_poly) {
    uint64_t leaves = 1;
    while (leaves * ZERO_TREE_LEAF < n_missing) leaves <<= 1;             // <= length / 16: the root has degree <= length
    const uint64_t dtot = leaves * ZERO_TREE_LEAF, pad = dtot - n_missing;
    dtmp<fr> d_a(s), d_b(s), d_f(s), d_g(s);
    CHK(d_a.alloc(dtot)); CHK(d_b.alloc(dtot)); CHK(d_f.alloc(2 * dtot)); CHK(d_g.alloc(dtot));
    launch_zero_leaves(s, fs->d_expanded, fs->W / length, d_missing, n_missing, leaves, d_a.p);
    fr *cur = d_a.p, *nxt = d_b.p;
    for (uint64_t d = ZERO_TREE_LEAF, nodes = leaves; nodes > 1; d <<= 1, nodes >>= 1) {
        fr_fft_rows(fs, s, cur, d, d, d_f.p, 2 * d, nodes, 0);            // every node's a, zero-extended to 2d values
        launch_zero_pair_products(s, d_f.p, 2 * d, nodes / 2, d_g.p);
        fr_fft_rows(fs, s, d_g.p, 2 * d, 2 * d, nxt, 2 * d, nodes / 2, 1);   // a b
        launch_zero_join(s, nxt, cur, d, nodes / 2);                      // + x^d (a + b)
        std::swap(cur, nxt);
    }
    launch_zero_unpad(s, cur, pad, n_missing, length, d_poly);
    fr_fft_rows(fs, s, d_poly, length, length, d_eval, length, 1, 0);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
static int zero_poly_dev(kzg_hip_fft *fs, hipStream_t s, const uint64_t *d_missing, uint64_t n_missing, uint64_t length, fr *d_eval, fr *d_poly) {
    static const int forced = [] { const char *e = getenv("KZG_HIP_ZERO_POLY"); return !e ? 0 : !strcmp(e, "direct") ? 1 : !strcmp(e, "tree") ? 2 : 0; }();
    // measured crossover (half of the domain missing): 8192 points, where both take 0.7 ms; 32 768 points: 4.9 ms direct, 1.2 ms through the tree
    if (forced == 2 || (forced == 0 && n_missing >= 1024 && n_missing * length >= (1ull << 26))) return zero_poly_tree(fs, s, d_missing, n_missing, length, d_eval, d_poly);
    launch_zero_eval_direct(s, fs->d_expanded, fs->W / length, d_missing, n_missing, length, d_eval);
    fr_fft_rows(fs, s, d_eval, length, length, d_poly, length, 1, 1);     // coefficients: degree n_missing < length
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_zero_poly_via_multiplication(kzg_hip_fft *fs, const uint64_t *missing_indices, uint64_t n_missing, uint64_t length,
                                         void *out_zero_eval_fr, void *out_zero_poly_fr) {
    if (!fs || !out_zero_eval_fr || !out_zero_poly_fr || (n_missing && !missing_indices)) return KZG_HIP_ERR_BAD_ARG;
    if (n_missing == 0) {                                    // zero_poly.go:117-119
        memset(out_zero_eval_fr, 0, length * sizeof(fr)); memset(out_zero_poly_fr, 0, length * sizeof(fr));
        return KZG_HIP_OK;
    }
    if (length > fs->W) return KZG_HIP_ERR_TOO_WIDE;         // "domain too small for requested length" :120-122
    if (!is_pow2(length)) return KZG_HIP_ERR_NOT_POW2;       // "length not a power of two" :123-125
    if (n_missing >= length) return KZG_HIP_ERR_BAD_ARG;     // "expected output smaller or equal to input length" :205-207
    for (uint64_t i = 0; i < n_missing; i++) if (missing_indices[i] >= length) return KZG_HIP_ERR_BAD_ARG;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<uint64_t> d_m(s); dtmp<fr> d_e(s), d_p(s);
    CHK(d_m.alloc(n_missing)); CHK(d_e.alloc(length)); CHK(d_p.alloc(length));
    HIPCHK(hipMemcpyAsync(d_m.p, missing_indices, n_missing * 8, hipMemcpyHostToDevice, s));
    CHK(zero_poly_dev(fs, s, d_m.p, n_missing, length, d_e.p, d_p.p));
    HIPCHK(hipMemcpyAsync(out_zero_eval_fr, d_e.p, length * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_zero_poly_fr, d_p.p, length * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_recover_poly_from_samples(kzg_hip_fft *fs, const void *samples_fr, const uint8_t *present, uint64_t n, void *out_fr) {
    if (!fs || !samples_fr || !present || !out_fr || n == 0) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    KZG_TRY
    std::vector<uint64_t> missing;
    for (uint64_t i = 0; i < n; i++) if (!present[i]) missing.push_back(i);   // recover_from_samples.go:44-49
    if (missing.size() >= n) return KZG_HIP_ERR_BAD_ARG;
    if (missing.empty()) { memcpy(out_fr, samples_fr, n * sizeof(fr)); return KZG_HIP_OK; }   // zero poly == 0: nothing to divide by; data complete
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<uint64_t> d_m(s); dtmp<uint8_t> d_pr(s); dtmp<uint32_t> d_flag(s);
    dtmp<fr> d_s(s), d_ze(s), d_zp(s), d_a(s), d_b(s), d_c(s), d_f(s);
    CHK(d_m.alloc(missing.size())); CHK(d_pr.alloc(n)); CHK(d_flag.alloc(1)); CHK(d_s.alloc(n)); CHK(d_ze.alloc(n)); CHK(d_zp.alloc(n));
    CHK(d_a.alloc(n)); CHK(d_b.alloc(n)); CHK(d_c.alloc(n)); CHK(d_f.alloc(2));
    fr five = fr_from_u64(5), f2[2] = {inv<FrP>(five), five};             // ShiftPoly uses 5^-1, UnshiftPoly 5 (:9-40)
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(d_m.p, missing.data(), missing.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_pr.p, present, n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_s.p, samples_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_f.p, f2, sizeof f2, hipMemcpyHostToDevice, s));
    CHK(zero_poly_dev(fs, s, d_m.p, missing.size(), n, d_ze.p, d_zp.p));
    launch_fr_pointwise(s, d_s.p, d_ze.p, d_pr.p, d_a.p, n, 0, nullptr);  // polyEvaluationsWithZero
    fr_fft_rows(fs, s, d_a.p, n, n, d_b.p, n, 1, 1);                      // polyWithZero
    launch_fr_scale_by_powers(s, d_b.p, d_f.p, n);                        // ShiftPoly(polyWithZero)
    launch_fr_scale_by_powers(s, d_zp.p, d_f.p, n);                       // ShiftPoly(zeroPoly)
    fr_fft_rows(fs, s, d_b.p, n, n, d_a.p, n, 1, 0);                      // evalShiftedPolyWithZero
    fr_fft_rows(fs, s, d_zp.p, n, n, d_c.p, n, 1, 0);                     // evalShiftedZeroPoly
    launch_fr_pointwise(s, d_a.p, d_c.p, d_pr.p, d_b.p, n, 1, nullptr);   // division
    fr_fft_rows(fs, s, d_b.p, n, n, d_a.p, n, 1, 1);                      // shiftedReconstructedPoly
    launch_fr_scale_by_powers(s, d_a.p, d_f.p + 1, n);                    // UnshiftPoly
    fr_fft_rows(fs, s, d_a.p, n, n, d_b.p, n, 1, 0);                      // reconstructedData
    launch_fr_pointwise(s, d_b.p, d_s.p, d_pr.p, nullptr, n, 2, d_flag.p);
    HIPCHK(hipGetLastError());
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_fr, d_b.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return flag ? KZG_HIP_ERR_RECOVERY : KZG_HIP_OK;
    KZG_CATCH
}
