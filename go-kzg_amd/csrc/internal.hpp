// internal.hpp -- launcher prototypes shared by the kernel translation units and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.hpp"
#include "g1.hpp"

namespace kzg {

// ---------------- k_fr.hip ----------------
// Batched radix-2 FFT over F_r (replaces _fft / InplaceFFT, fft_fr.go:30-105).
//   in  : batch x in_stride Fr, of which the first n_in of each row are used (rest zero-padded to n)
//   out : batch x n Fr, natural order.  roots = ExpandedRootsOfUnity or ReverseRootsOfUnity (W + 1 entries).
//   scale != nullptr: every output is multiplied by *scale (device pointer; n^-1 for the inverse).
//   tw4096 != nullptr and n == 4096: the radix-4 kernel on lazy limbs with that twiddle file (fr_fft4096.hpp; same direction as roots)
void launch_fr_fft(hipStream_t s, const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, uint64_t n, uint64_t batch,
                   const fr *roots, uint64_t W, const fr *scale, const uint32_t *tw4096 = nullptr, const fr *roots_l = nullptr);
// DASFFTExtension (das_extension.go:7-84), in place on batch rows of n values.
//   tw2048 != nullptr and n == 2048: the lazy-limb kernel with that twiddle file (fr_das2048.hpp, built from the same two tables)
void launch_das_ext(hipStream_t s, fr *vals, uint64_t n, uint64_t batch, const fr *expanded, const fr *reversed, uint64_t W,
                    const fr *inv_n, const uint32_t *tw2048 = nullptr);
// toeplitzCoeffsStepStrided (fk20_single.go:89-103): out[b][file][0..2k) from poly[b][0..n), optionally scaled.
void launch_toeplitz_coeffs(hipStream_t s, const fr *poly, uint64_t poly_stride, uint64_t n, uint64_t l, uint64_t batch, fr *out,
                            const fr *scale);
// quotients of `batch` polynomials (rows of poly_stride, n coefficients used) by (X - x[b]) (polyLongDiv with divisor [-x, 1],
// poly.go:14-40): row b of q (stride q_stride) gets n - 1 entries
void launch_quotient_linear(hipStream_t s, const fr *poly, uint64_t poly_stride, uint64_t n, uint64_t batch, const fr *x, fr *q, uint64_t q_stride);
void launch_fr_from_u64(hipStream_t s, const uint64_t *in, uint64_t in_stride, fr *out, uint64_t n);   // bls.AsFr over a slice
void launch_fr_zero_tails(hipStream_t s, fr *rows, uint64_t n_max, uint64_t batch, const uint64_t *lens, uint64_t lens_stride);
// returns (via *flag != 0) whether any of vals[0..n) is non-zero
void launch_fr_any_nonzero(hipStream_t s, const fr *vals, uint64_t n, uint32_t *flag);
void launch_fr_powers(hipStream_t s, const fr *base, uint64_t n, fr *out);   // out[i] = base^i

// eth/ byte-level path
void launch_fr_from_le32(hipStream_t s, const uint8_t *in, fr *out, uint64_t per_blob, uint64_t batch, uint32_t *bad);
void launch_fr_to_le32(hipStream_t s, const fr *in, uint8_t *out, uint64_t n);
void launch_fr_bitrev_gather(hipStream_t s, const fr *in, fr *out, uint64_t n);
// rows of (polynomial, z): q[row] = quotient, y_out[row] = p(z), flag[row] = 1 where z is in the domain (that row's quotient is zero)
constexpr uint64_t ZERO_TREE_LEAF = 16;   // roots per leaf of the vanishing-polynomial product tree (k_zero_leaves)
void launch_zero_leaves(hipStream_t s, const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t leaves, fr *a);
void launch_zero_pair_products(hipStream_t s, const fr *f, uint64_t m, uint64_t pairs, fr *out);
void launch_zero_join(hipStream_t s, fr *c, const fr *a, uint64_t d, uint64_t pairs);
void launch_zero_unpad(hipStream_t s, const fr *root, uint64_t pad, uint64_t n_missing, uint64_t length, fr *poly);
void launch_fr_mul_table_rows(hipStream_t s, fr *data, const fr *table, uint64_t stride, uint64_t n, uint64_t batch);
void launch_poly_lincomb(hipStream_t s, const fr *vectors, uint64_t stride, const fr *scalars, uint64_t count, uint64_t n, fr *out);   // bls.PolyLinComb
void launch_eth_quotient(hipStream_t s, const fr *poly, uint64_t poly_stride, const fr *domain, uint64_t n, uint64_t batch, const fr *z, uint64_t z_stride,
                         const fr *inv_n, fr *q, fr *y_out, uint32_t *flag, uint64_t dom_stride = 1, fr *scratch = nullptr);   // scratch: eth_quotient_scratch_elems(n, batch) elements (the row-split form of small batches) or null
uint64_t eth_quotient_scratch_elems(uint64_t n, uint64_t batch);

void launch_fr_scale_by_inv_powers(hipStream_t s, fr *c, const fr *x, uint64_t n, fr *xpow_n);   // c_i /= x^i; *xpow_n = x^n

// erasure recovery
void launch_zero_eval_direct(hipStream_t s, const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t length, fr *zero_eval);
void launch_fr_scale_by_powers(hipStream_t s, fr *poly, const fr *base, uint64_t n);
void launch_fr_pointwise(hipStream_t s, const fr *a, const fr *b, const uint8_t *present, fr *out, uint64_t n, int mode, uint32_t *flag);

// ---------------- k_g1.hip ----------------
// out[i] = scalars[i * s_stride] * pts[(i % pts_mod)]   (element-wise bls.MulG1; scalars in Montgomery form)
void launch_g1_mul_vec(hipStream_t s, const g1j *pts, uint64_t pts_mod, const fr *scalars, uint64_t s_stride, uint64_t n, g1j *out);
// acc[i] = sum over f < nfiles of scalars[b][f][j] * files[f][j]  -- the FK20-multi Toeplitz stage (fk20_multi.go:79-91)
hipError_t launch_g1_file_msm(hipStream_t s, const g1j *files, const fr *scalars, uint64_t nfiles, uint64_t k2, uint64_t j0, uint64_t cnt,
                              uint64_t batch, g1j *out);
// out[b][rev(i)] = i < n_valid ? in[b][i] : inf     (bit-reversal + "h[:n] || inf" padding, fk20_single.go:163-166)
void launch_g1_bitrev_copy(hipStream_t s, const g1j *in, uint64_t in_stride, uint64_t n_valid, g1j *out, uint64_t n, uint64_t batch);
// one radix-2 DIT stage on bit-reversed data (replaces the loop of _fftG1, fft_g1.go:44-55); `roots` holds the twiddles as
// GLV pairs (k mod lambda, k div lambda) in standard form, see g1_mul_glv
void launch_g1_fft_stage(hipStream_t s, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W);   // wnaf: the twiddles' precomputed digit strings (KZG_WNAF_ROW bytes each)
// decimation-in-frequency stage on the same pairs / twiddles: (x, y) -> (x + y, (x - y) w); and the odd-position clear between the
// FK20 transforms (h[:n] || inf in bit-reversed order)
void launch_g1_fft_stage_dif(hipStream_t s, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W);
void launch_g1_clear_odd(hipStream_t s, g1j *data, uint64_t n_total);
void launch_g1_take_even(hipStream_t s, const g1j *in, g1j *out, uint64_t total);   // out[t] = in[2 t]
bool g1_quad_enabled();   // the stage launchers put four lanes on a butterfly for launches of at most 16 384 butterflies (g1_quad.hpp) unless KZG_HIP_G1_QUAD=0
// latency mode: Stockham passes of radix 16 evaluated directly (k_g1.hip); result in data, tmp = batch x n scratch, scale optional
void launch_fb_direct_pass1(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, const fr *roots, uint64_t W, uint64_t batch,
                            uint32_t logR, g1j *out, bool glv = false);   // FK20 Toeplitz stage + first direct pass of the inverse transform (a lone polynomial)
// several lanes per multiplication (g1_coop_kernels.hpp; translation units of their own)
void launch_g1_stage_coop_dif(hipStream_t s, int lanes, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W, uint64_t total);
void launch_g1_stage_coop_dit(hipStream_t s, int lanes, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W, uint64_t total);
void launch_g1_direct_coop(hipStream_t s, int lanes, uint32_t wgs, size_t pad_lds, const g1j *src, uint64_t src_stride, uint64_t src_valid, g1j *dst, uint32_t logn, uint32_t logR,
                           uint64_t Ns, const fr *roots, uint64_t W, const fr *sc, uint64_t total, uint32_t logT, uint32_t logU);
void launch_g1_fft_direct(hipStream_t s, const g1j *in, uint64_t in_stride, uint64_t n_valid, g1j *data, g1j *tmp, uint64_t n, uint64_t batch, const fr *roots,
                          uint64_t W, const fr *scale, uint32_t max_logr = 4, int lanes = 1, uint32_t bits_done = 0, uint64_t n_out = 0);   // n_out: only the first n_out outputs are wanted (0 = all)
// to_kilic: also leave the device-internal Montgomery domain (R' = 2^390) for Kilic's (2^384): every API output path ends here
void launch_g1_normalize(hipStream_t s, const g1j *in, g1j *out, uint64_t n, bool to_kilic = false);
void launch_fr_inv_test(hipStream_t s, const fr *in, uint64_t n, fr *out_coop, fr *out_lane, fr *out_block);   // test hook: F_r inversion three ways (k_fr.hip)
void launch_fp_inv_both(hipStream_t s, const fp *in, fp *out_coop, fp *out_lane, uint64_t n, int mode);   // test / measurement hook: wave-cooperative (bit 0) and one-lane (bit 1) inversion of n elements, one wavefront each
void launch_g1_from_kilic(hipStream_t s, g1j *data, uint64_t n);   // in place: caller-supplied points enter the internal domain
void launch_g1_to_affine(hipStream_t s, const g1j *in, g1a *out, uint64_t n);
void launch_g1_compress(hipStream_t s, const g1j *in, uint8_t *out48, uint64_t n);
void launch_g1_decompress(hipStream_t s, const uint8_t *in48, g1j *out, uint64_t n, uint32_t *bad_flag);
void launch_g1_fixed_base_powers(hipStream_t s, const fr *powers, uint64_t n, g1j *out);   // out[i] = powers[i] * G

// ---------------- k_msm.hip ----------------
// Lanes of ONE wavefront on every SIMD of the calling thread's current device: CUs x 4 SIMDs x 64 lanes (65 536 on the 256 CUs of an MI355X
// in SPX mode; fewer on a partitioned part).  Every "does this launch still fit one wavefront per SIMD" heuristic of the launchers is written
// against this value, never against a literal.  Cached per device, thread-safe.
uint64_t device_simd_lanes();
inline uint64_t device_simds() { return device_simd_lanes() / 64; }
struct msm_plan {
    uint32_t c;        // window bits (fixed-base walk: signed digits in [-2^(c-1), 2^(c-1)]; bucket MSM: always 8)
    uint32_t nwin;     // windows of the fixed-base walk
    uint32_t nb;       // 2^(c-1)
    uint32_t ngroups;  // bucket MSM: 16 window groups (points only) or 8 (table also holds 2^64 P_i at [table_n + i])
    uint32_t fixed;    // 1: plan of a fixed-base table (k_fb_*), 0: bucket MSM
    uint32_t glv;      // fixed-base walk: 1 = the table holds ceil(128 / c) windows and both GLV halves of a scalar walk them (k_fb_accumulate_glv)
    uint64_t table_n;  // points per row of the table
};
// the bucket pipeline packs (point index << 2 | half | sign) into 32 bits and counts entries (32 per scalar) in 32 bits: both must fit
inline bool msm_index_range_ok(const msm_plan &p, uint64_t n) { return n < (1ull << 27) && 2 * p.table_n < (1ull << 30); }
size_t msm_workspace_bytes(const msm_plan &p, uint64_t n, uint64_t batch);
// batch MSMs over the same affine points, scalars in rows of sc_stride: out[b] = sum_i scalars[b][i] * P_i, NORMALISED (Z = one),
// as Kilic images when to_kilic
void launch_msm(hipStream_t s, const msm_plan &p, const g1a *table, const fr *scalars, uint64_t sc_stride, uint64_t n, uint64_t batch, void *workspace, g1j *out,
                bool to_kilic);
// fixed-base window table: out[w * n + i] = 2^(c w) * pts[i], affine
void launch_msm_window_table(hipStream_t s, const g1a *pts, uint64_t n, uint32_t c, uint32_t nwin, g1j *tmp, g1a *out);

// fixed-base table MSM over the device-resident setup (see k_msm.hip)
size_t fb_partials_bytes(uint64_t n, uint64_t batch);
hipError_t launch_fb_build(hipStream_t s, const g1a *pts, uint64_t n, uint32_t c, uint32_t nwin, g1a *table);
void launch_fb_msm(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t sc_stride, uint64_t n,
                   uint64_t batch, void *partials, g1j *out, bool to_kilic, bool glv = false, bool projective = false);

// out[(b, f, jj)] = scalars[b][f * row + j0 + jj] * P[f * row + j0 + jj] over a fixed-base table of table_n = nfiles * row points
// (glv, here and below: the table holds ceil(128 / c) windows and both GLV halves of every scalar walk them)
void launch_fb_mul_vec(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t row, uint64_t j0,
                       uint64_t cnt, uint64_t batch, g1j *out, bool glv = false);
// the same stage fused with the first two decimation-in-frequency stages of the inverse G1 transform (single-file tables, N >= 4):
// out[b][.] = two DIF stages applied to (scalars[b][j] * P_j)_j; roots = ReverseRootsOfUnity (Montgomery) of width W
void launch_fb_mul_vec_dif2(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, const fr *roots, uint64_t W,
                            uint64_t batch, g1j *out, bool glv = false);
// all files of an output position summed in one lane: out[b][jj] = sum_f scalars[b][f * row + j0 + jj] * P[f * row + j0 + jj]
void launch_fb_mul_vec_files(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t row, uint64_t j0,
                             uint64_t cnt, uint64_t batch, g1j *out, bool glv = false);
void launch_g1_sum_files(hipStream_t s, const g1j *tmp, uint64_t nfiles, uint64_t cnt, uint64_t batch, g1j *out);

// profiling hook (HIP events around the dominant kernel), see capi.hip
void prof_begin(hipStream_t s, const char *name);
void prof_end(hipStream_t s, const char *name);

}  // namespace kzg
