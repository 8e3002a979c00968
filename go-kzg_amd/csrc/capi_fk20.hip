// capi_fk20.hip -- FK20 single and multi (a9-a12): settings, pipelines, sharded slice / finish
#include "capi_common.hpp"

// ---------------------------------------------------------------------------------------------------------
// FK20 (single == multi with chunk length 1)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_fk20_x(const g1j *secret, uint64_t n, uint64_t l, uint64_t k, g1j *x /* l x k */) {
    // kzg.go:53-58 (single) / :101-111 (multi): x_off[i] = SecretG1[n - l - 1 - off - i l] for i < k - 1, x_off[k-1] = inf
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= l * k) return;
    uint64_t off = t / k, i = t % k;
    x[t] = (i + 1 < k) ? secret[n - l - 1 - off - i * l] : g1_inf();
}

static int fk20_core_new(kzg_hip_kzg *ks, uint64_t n2, uint64_t l, fk20_core *c) {
    kzg_hip_fft *fs = ks->fs;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    uint64_t n = n2 / 2, k = n / l, k2 = 2 * k;
    c->ks = ks; c->n2 = n2; c->l = l; c->k = k;
    dtmp<g1j> d_x(s), d_f(s);
    CHK(d_x.alloc(l * k)); CHK(d_f.alloc(l * k2));
    HIPCHK(hipMalloc((void **)&c->d_files, l * k2 * sizeof(g1j)));
    hipLaunchKernelGGL(k_fk20_x, dim3((uint32_t)((l * k + 255) / 256)), dim3(256), 0, s, ks->d_secret, n, l, k, d_x.p);
    CHK(g1_fft_rows(fs, s, d_x.p, k, k, d_f.p, k2, l, 0));   // toeplitzPart1: FFTG1(x || inf^k), fk20_single.go:40-56
    launch_g1_normalize(s, d_f.p, c->d_files, l * k2);
    HIPCHK(hipGetLastError());
    {   // fixed-base table over the file points, sized by KZG_HIP_FK20_FB_BUDGET_GB (default: min(48 GB, free HBM - 12 GB)).  Both GLV halves of a coefficient
        // walk one table of ceil(128 / c) windows (round 5; KZG_HIP_FB_GLV=0: the plain layout): scale 12, l = 1 (4096 file points): c = 13, 2 x 10 additions, 16 GB
        // (plain: c = 13, 20, 32 GB);  scale 13 (8192 points): c = 13, 2 x 10, 32 GB (plain: c = 12, 22, 35 GB);  scale 16, l = 16 (65 536 points): c = 10, 2 x 13, 42 GB
        // (plain: c = 9, 29, 47 GB).
        // An allocation failure falls back to the next smaller window and finally to the table-free double-and-add path.
        double budget_gb = table_budget_gb("KZG_HIP_FK20_FB_BUDGET_GB", 48.0, 12.0);
        uint64_t npts = l * k2;
        const bool glv = fb_glv_enabled();
        dtmp<g1a> d_fa(s);
        if (npts >= 64) { CHK(d_fa.alloc(npts)); launch_g1_to_affine(s, c->d_files, d_fa.p, npts); }
        for (uint32_t cc = glv ? 16 : 14; cc >= 4 && npts >= 64; cc--) {
            const uint32_t nwin = glv ? fb_windows_glv(cc) : fb_windows(cc);
            // (GLV: a window size whose additions per coefficient a smaller table also reaches is skipped)
            if (glv && cc > 4 && fb_windows_glv(cc - 1) == nwin) continue;
            double bytes = (double)nwin * (double)npts * (double)(1u << (cc - 1)) * sizeof(g1a);
            if (bytes > budget_gb * 1e9) continue;
            g1a *tab = nullptr;
            if (hipMalloc((void **)&tab, (size_t)nwin * npts * (1u << (cc - 1)) * sizeof(g1a)) != hipSuccess) { (void)hipGetLastError(); continue; }
            hipError_t e = launch_fb_build(s, d_fa.p, npts, cc, nwin, tab);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipGetLastError(); hipFree(tab); continue; }
            c->d_files_fb = tab; c->fb_c = cc; c->fb_nwin = nwin; c->fb_glv = glv;
            break;
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}

// steps 1-3: Toeplitz coefficients (pre-scaled by 1/2k, which folds the inverse FFT's scale into the scalars),
// FFT_Fr, and hExtFFT[j] = sum_f C_f[j] * X_f[j] for j in [j0, j0 + cnt)
int fk20_hext(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, uint64_t j0, uint64_t cnt, g1j *d_hext) {
    kzg_hip_fft *fs = c->ks->fs;
    uint64_t l = c->l, k2 = 2 * c->k;
    dtmp<fr> d_tc(s), d_cf(s);
    CHK(d_tc.alloc(batch * l * k2)); CHK(d_cf.alloc(batch * l * k2));
    launch_toeplitz_coeffs(s, d_poly, poly_stride, n, l, batch, d_tc.p, fs->d_inv_pow2 + ilog2(k2));
    fr_fft_rows(fs, s, d_tc.p, k2, k2, d_cf.p, k2, batch * l, 0);
    if (c->d_files_fb) {
        if (l == 1) launch_fb_mul_vec(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_hext, c->fb_glv);
        else if (batch * cnt >= device_simd_lanes()) launch_fb_mul_vec_files(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_hext, c->fb_glv);   // enough output positions to fill the GPU: one lane sums all files
        else {   // few positions (one polynomial, a shard): a lane per (file, position), then the sum over the files
            dtmp<g1j> d_tmp(s);
            CHK(d_tmp.alloc(batch * l * cnt));
            launch_fb_mul_vec(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_tmp.p, c->fb_glv);
            launch_g1_sum_files(s, d_tmp.p, l, cnt, batch, d_hext);
        }
    } else if (l == 1 && j0 == 0 && cnt == k2) launch_g1_mul_vec(s, c->d_files, k2, d_cf.p, 1, batch * k2, d_hext);
    else HIPCHK(launch_g1_file_msm(s, c->d_files, d_cf.p, l, k2, j0, cnt, batch, d_hext));
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
// steps 4-7: h = IFFT_G1(hExtFFT)[:k] (scale already folded), out = FFT_G1(h || inf^k) (da) or FFT_G1(h) (plain),
// optional reverse-bit-order, normalise
// (second half: d_h = the inverse transform of hExtFFT, batch x 2k points of which the first k are h; d_h is overwritten)
static int fk20_finish_from_h(fk20_core *c, hipStream_t s, g1j *d_h, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    kzg_hip_fft *fs = c->ks->fs;
    uint64_t k = c->k, k2 = 2 * k, on = da ? k2 : k;
    dtmp<g1j> d_b(s);
    CHK(d_b.alloc(batch * k2));
    CHK(g1_fft_rows(fs, s, d_h, k2, k, d_b.p, on, batch, 0));              // fk20_single.go:163-167 / :129
    if (bit_reverse) { launch_g1_bitrev_copy(s, d_b.p, on, on, d_h, on, batch); launch_g1_normalize(s, d_h, d_out, batch * on, true); }
    else launch_g1_normalize(s, d_b.p, d_out, batch * on, true);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int fk20_finish(fk20_core *c, hipStream_t s, const g1j *d_hext, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    kzg_hip_fft *fs = c->ks->fs;
    const uint64_t k2 = 2 * c->k;
    dtmp<g1j> d_a(s);
    CHK(d_a.alloc(batch * k2));
    CHK(g1_fft_rows(fs, s, d_hext, k2, k2, d_a.p, k2, batch, 1, nullptr, c->k));   // ToeplitzPart3, fk20_single.go:80-87 (only h[:k] is read below)
    return fk20_finish_from_h(c, s, d_a.p, batch, da, bit_reverse, d_out);
}
// A lone polynomial of a single-file settings object with its table resident, on the direct passes: the Toeplitz stage and the FIRST radix-16 pass of
// the inverse transform in one kernel (k_fb_direct_pass1: every term of that pass is a fixed-base product, nwin additions instead of a variable-base
// multiplication), the remaining passes continue from there.  KZG_HIP_FK20_PASS1=0 turns it off (tests compare both).
static bool fk20_pass1_fused_ok(const fk20_core *c, uint64_t batch) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_FK20_PASS1"); return e && e[0] == '0'; }();
    const uint64_t k2 = 2 * c->k;
    return !off && c->l == 1 && c->d_files_fb && k2 >= 32 && g1_fft_direct_mode(k2, batch) && g1_fft_direct_logr(k2, batch) == 4;
}
static int fk20_run_pass1_fused(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    kzg_hip_fft *fs = c->ks->fs;
    const uint64_t k2 = 2 * c->k;
    dtmp<fr> d_tc(s), d_cf(s); dtmp<g1j> d_p1(s), d_a(s), d_tmp(s);
    CHK(d_tc.alloc(batch * k2)); CHK(d_cf.alloc(batch * k2)); CHK(d_p1.alloc(batch * k2)); CHK(d_a.alloc(batch * k2)); CHK(d_tmp.alloc(batch * k2));
    launch_toeplitz_coeffs(s, d_poly, poly_stride, n, 1, batch, d_tc.p, fs->d_inv_pow2 + ilog2(k2));   // 1 / 2k folded into the scalars
    fr_fft_rows(fs, s, d_tc.p, k2, k2, d_cf.p, k2, batch, 0);
    launch_fb_direct_pass1(s, c->d_files_fb, k2, c->fb_c, c->fb_nwin, d_cf.p, fs->d_reversed, fs->W, batch, 4, d_p1.p, c->fb_glv);
    launch_g1_fft_direct(s, d_p1.p, k2, k2, d_a.p, d_tmp.p, k2, batch, fs->d_reversed, fs->W, nullptr, 4, g1_fft_direct_lanes(k2, batch), 4, c->k);   // only h[:k] is read below
    HIPCHK(hipGetLastError());
    return fk20_finish_from_h(c, s, d_a.p, batch, da, bit_reverse, d_out);
}
// A single-file settings object with its table resident: the Toeplitz stage absorbs the first two stages of the inverse
// transform (k_fb_mul_vec_dif2), the remaining ones run decimation-in-frequency and leave h bit-reversed, which is the layout the
// forward (decimation-in-time) transform reads: 10 instead of 12 multiplying stages for the inverse transform and no reordering
// passes.  Same group elements as the plain pipeline; outputs are normalised, so the bytes are identical (tests compare both).
// The plain form (FK20Single, fk20_single.go:122-137: proofs = FFT_G1(h) on k points) continues from the EVEN positions of that layout, which
// are h[:k] in k-point bit-reversed order.
static bool fk20_fused_ok(const fk20_core *c, uint64_t batch, int da) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_FK20_FUSE"); return e && e[0] == '0'; }();
    const uint64_t k2 = 2 * c->k;
    return !off && c->l == 1 && c->d_files_fb && k2 >= 8 && !g1_fft_direct_mode(k2, batch) && (da || !g1_fft_direct_mode(c->k, batch));
}
static int fk20_run_fused(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    kzg_hip_fft *fs = c->ks->fs;
    const uint64_t k2 = 2 * c->k;
    dtmp<fr> d_tc(s), d_cf(s); dtmp<g1j> d_a(s), d_b(s);
    CHK(d_tc.alloc(batch * k2)); CHK(d_cf.alloc(batch * k2)); CHK(d_a.alloc(batch * k2)); CHK(d_b.alloc(batch * k2));
    launch_toeplitz_coeffs(s, d_poly, poly_stride, n, 1, batch, d_tc.p, fs->d_inv_pow2 + ilog2(k2));   // 1 / 2k folded into the scalars
    fr_fft_rows(fs, s, d_tc.p, k2, k2, d_cf.p, k2, batch, 0);
    launch_fb_mul_vec_dif2(s, c->d_files_fb, k2, c->fb_c, c->fb_nwin, d_cf.p, fs->d_reversed, fs->W, batch, d_a.p, c->fb_glv);
    for (uint64_t m = k2 / 8; m >= 1; m >>= 1) launch_g1_fft_stage_dif(s, d_a.p, k2, batch, m, fs->d_glv_reversed, fs->d_wnaf_reversed, fs->W);
    if (!da) {                                                  // FK20Single: transform of k points on h[:k]
        const uint64_t k = c->k;
        launch_g1_take_even(s, d_a.p, d_b.p, batch * k);
        for (uint64_t m = 1; m < k; m <<= 1) launch_g1_fft_stage(s, d_b.p, k, batch, m, fs->d_glv_expanded, fs->d_wnaf_expanded, fs->W);
        if (bit_reverse) { launch_g1_bitrev_copy(s, d_b.p, k, k, d_a.p, k, batch); launch_g1_normalize(s, d_a.p, d_out, batch * k, true); }
        else launch_g1_normalize(s, d_b.p, d_out, batch * k, true);
        HIPCHK(hipGetLastError());
        return KZG_HIP_OK;
    }
    launch_g1_clear_odd(s, d_a.p, batch * k2);                  // h[:k] || inf^k, in bit-reversed order
    for (uint64_t m = 1; m < k2; m <<= 1) launch_g1_fft_stage(s, d_a.p, k2, batch, m, fs->d_glv_expanded, fs->d_wnaf_expanded, fs->W);
    if (bit_reverse) { launch_g1_bitrev_copy(s, d_a.p, k2, k2, d_b.p, k2, batch); launch_g1_normalize(s, d_b.p, d_out, batch * k2, true); }
    else launch_g1_normalize(s, d_a.p, d_out, batch * k2, true);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
// Ragged batches run padded with copies of their last polynomial: the stage kernels take the irregular width-5 NAF schedule only where a wavefront
// holds one twiddle ((n / 2 / m) * batch a multiple of 64, or >= 256), so 17 or 31 polynomials took 45.5 / 45.9 ms against 36.5 for 32, and 63 took
// 79.9 against 64.7 for 64.  Up to 32 polynomials a stage is one wavefront per SIMD whatever the count; beyond, rows are only added where they cost under 3 %.
static uint64_t fk20_padded_batch(uint64_t batch) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_FK20_PAD"); return e && e[0] == '0'; }();
    if (off) return batch;
    if (batch > 16 && batch < 32) return 32;
    if (batch > 32 && (batch & 7)) {                          // beyond one wavefront per SIMD padding is work: only where it is < 3 % (63 -> 64: 79.9 -> 66.5 ms,
        const uint64_t p = (batch + 7) & ~7ull;                 // 127 -> 128: 140 -> 121 ms; 65 -> 72 and 100 -> 104 measured slower)
        if ((p - batch) * 32 <= batch) return p;
    }
    return batch;
}
int fk20_run_dev(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    if (const uint64_t padded = fk20_padded_batch(batch); padded != batch) {
        const uint64_t on = da ? 2 * c->k : c->k;
        dtmp<fr> d_p2(s); dtmp<g1j> d_o2(s);
        CHK(d_p2.alloc(padded * n)); CHK(d_o2.alloc(padded * on));
        HIPCHK(hipMemcpy2DAsync(d_p2.p, n * sizeof(fr), d_poly, poly_stride * sizeof(fr), n * sizeof(fr), batch, hipMemcpyDeviceToDevice, s));
        for (uint64_t b = batch; b < padded; b++)
            HIPCHK(hipMemcpyAsync(d_p2.p + b * n, d_poly + (batch - 1) * poly_stride, n * sizeof(fr), hipMemcpyDeviceToDevice, s));
        if (fk20_fused_ok(c, padded, da)) CHK(fk20_run_fused(c, s, d_p2.p, n, n, padded, da, bit_reverse, d_o2.p));
        else {
            dtmp<g1j> d_hext(s);
            CHK(d_hext.alloc(padded * 2 * c->k));
            CHK(fk20_hext(c, s, d_p2.p, n, n, padded, 0, 2 * c->k, d_hext.p));
            CHK(fk20_finish(c, s, d_hext.p, padded, da, bit_reverse, d_o2.p));
        }
        HIPCHK(hipMemcpyAsync(d_out, d_o2.p, batch * on * sizeof(g1j), hipMemcpyDeviceToDevice, s));
        return KZG_HIP_OK;
    }
    if (fk20_fused_ok(c, batch, da)) return fk20_run_fused(c, s, d_poly, poly_stride, n, batch, da, bit_reverse, d_out);
    if (fk20_pass1_fused_ok(c, batch)) return fk20_run_pass1_fused(c, s, d_poly, poly_stride, n, batch, da, bit_reverse, d_out);
    uint64_t k2 = 2 * c->k;
    dtmp<g1j> d_hext(s);
    CHK(d_hext.alloc(batch * k2));
    CHK(fk20_hext(c, s, d_poly, poly_stride, n, batch, 0, k2, d_hext.p));
    return fk20_finish(c, s, d_hext.p, batch, da, bit_reverse, d_out);
}
// host-buffer front end: poly rows of `row_len` values of which the first n are the coefficients
static int fk20_run_host(fk20_core *c, const void *poly_fr, uint64_t row_len, uint64_t n, uint64_t batch, int check_upper, int da, int bit_reverse, void *out_g1) {
    kzg_hip_fft *fs = c->ks->fs;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    uint64_t on = da ? 2 * c->k : c->k;
    dtmp<fr> d_poly(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_flag(s);
    CHK(d_poly.alloc(batch * row_len)); CHK(d_out.alloc(batch * on)); CHK(d_flag.alloc(1));
    HIPCHK(hipMemcpyAsync(d_poly.p, poly_fr, batch * row_len * sizeof(fr), hipMemcpyHostToDevice, s));
    if (check_upper) {   // "bad input, second half should be zeroed", fk20_single.go:150-154 / fk20_multi.go:65-69
        HIPCHK(hipMemsetAsync(d_flag.p, 0, 4, s));
        launch_fr_any_nonzero(s, d_poly.p + n, row_len - n, d_flag.p);
        uint32_t flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (flag) return KZG_HIP_ERR_UPPER_HALF;
    }
    CHK(fk20_run_dev(c, s, d_poly.p, row_len, n, batch, da, bit_reverse, d_out.p));
    return d2h_staged(c->ks->fs, s, out_g1, d_out.p, batch * on * sizeof(g1j));
}

int kzg_hip_fk20_single_settings_new(kzg_hip_kzg *ks, uint64_t n2, kzg_hip_fk20s **out) {
    if (!ks || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (n2 > ks->fs->W) return KZG_HIP_ERR_TOO_WIDE;     // kzg.go:44-46
    if (!is_pow2(n2)) return KZG_HIP_ERR_NOT_POW2;       // kzg.go:47-49
    if (n2 < 2) return KZG_HIP_ERR_BAD_ARG;              // kzg.go:50-52
    kzg_hip_fk20s *fk = new kzg_hip_fk20s;
    int st = fk20_core_new(ks, n2, 1, &fk->c);
    if (st) { kzg_hip_fk20_single_settings_free(fk); return st; }
    *out = fk;
    return KZG_HIP_OK;
}
void kzg_hip_fk20_single_settings_free(kzg_hip_fk20s *fk) {
    if (!fk) return;
    if (fk->c.ks) { hipSetDevice(fk->c.ks->fs->device); hipDeviceSynchronize(); }
    hipFree(fk->c.d_files); hipFree(fk->c.d_files_fb);
    (void)hipGetLastError();
    delete fk;
}
int kzg_hip_fk20_single_x_ext_fft(const kzg_hip_fk20s *fk, void *out_g1) {
    if (!fk || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    kzg_hip_fft *fs = fk->c.ks->fs;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    dtmp<g1j> d_tmp(s);
    CHK(d_tmp.alloc(fk->c.n2));
    launch_g1_normalize(s, fk->c.d_files, d_tmp.p, fk->c.n2, true);   // device-internal -> Kilic images
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_tmp.p, fk->c.n2 * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_fk20_single(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;   // ToeplitzPart2 length panic, fk20_single.go:60-62
    return fk20_run_host(&fk->c, poly_fr, n, n, 1, 0, 0, 0, out_g1);
}
// `batch` polynomials of n coefficients -> batch x n proofs (FK20Single on each): host buffers / device pointers + stream
int kzg_hip_fk20_single_batch(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;   // ToeplitzPart2 length panic, fk20_single.go:60-62
    if (!batch) return KZG_HIP_OK;
    return fk20_run_host(&fk->c, poly_fr, n, n, batch, 0, 0, 0, out_g1);
}
int kzg_hip_fk20_single_batch_dev(kzg_hip_fk20s *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream) {
    if (!fk || !d_poly_fr || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    dev_guard g(fk->c.ks->fs);
    return fk20_run_dev(&fk->c, (hipStream_t)stream, (const fr *)d_poly_fr, n, n, batch, 0, 0, (g1j *)d_out_g1);
}
int kzg_hip_fk20_single_da_optimized(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n2, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n2 > fk->c.ks->fs->W) return KZG_HIP_ERR_TOO_WIDE;    // fk20_single.go:140-144
    if (!is_pow2(n2)) return KZG_HIP_ERR_NOT_POW2;            // fk20_single.go:146-148
    if (n2 != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    return fk20_run_host(&fk->c, poly_fr, n2, n2 / 2, 1, 1, 1, 0, out_g1);
}
int kzg_hip_da_using_fk20_batch(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;   // fk20_single.go:178-180
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;               // fk20_single.go:181-183
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    return fk20_run_host(&fk->c, poly_fr, n, n, batch, 0, 1, 1, out_g1);
}
// one DAUsingFK20 / DAUsingFK20Multi call through the handle's coalescer: concurrent callers share one batched run
static int fk20_da_coalesced(fk20_core *c, const void *poly_fr, uint64_t n, void *out_g1) {
    KZG_TRY
    const uint64_t on = 2 * c->k;
    coalescer *co = get_coalescer(c->ks->fs, c->co_da, n * sizeof(fr), on * sizeof(g1j), 48);   // two half batches side by side from 48 callers on (r04: 930/s from 64 threads; one batch of 64: 760-840/s)
    auto exec = [c, co, n, on](coalesce_buf &b, uint64_t batch) -> int {
        hipSetDevice(c->ks->fs->device);
        hipStream_t s = b.stream;
        drain_on_exit drain(s);
        dtmp<fr> d_poly(s); dtmp<g1j> d_out(s);
        CHK(d_poly.alloc(batch * n)); CHK(d_out.alloc(batch * on));
        HIPCHK(hipMemcpyAsync(d_poly.p, b.h_in, batch * co->in_row_bytes(), hipMemcpyHostToDevice, s));
        CHK(fk20_run_dev(c, s, d_poly.p, n, n, batch, 1, 1, d_out.p));
        HIPCHK(hipMemcpyAsync(b.h_out, d_out.p, batch * on * sizeof(g1j), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return KZG_HIP_OK;
    };
    return co->submit(poly_fr, n * sizeof(fr), n, 0, out_g1, on * sizeof(g1j), exec, KZG_HIP_ERR_HIP);
    KZG_CATCH
}
int kzg_hip_da_using_fk20(kzg_hip_fk20s *fk, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;   // fk20_single.go:178-180
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;               // fk20_single.go:181-183
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!coalescing_enabled()) return fk20_run_host(&fk->c, poly_fr, n, n, 1, 0, 1, 1, out_g1);
    return fk20_da_coalesced(&fk->c, poly_fr, n, out_g1);
}
int kzg_hip_da_using_fk20_batch_dev(kzg_hip_fk20s *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream) {
    if (!fk || !d_poly_fr || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    dev_guard g(fk->c.ks->fs);
    return fk20_run_dev(&fk->c, (hipStream_t)stream, (const fr *)d_poly_fr, n, n, batch, 1, 1, (g1j *)d_out_g1);
}

int kzg_hip_fk20_multi_settings_new(kzg_hip_kzg *ks, uint64_t n2, uint64_t chunk_len, kzg_hip_fk20m **out) {
    if (!ks || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (n2 > ks->fs->W) return KZG_HIP_ERR_TOO_WIDE;          // kzg.go:74-76
    if (!is_pow2(n2)) return KZG_HIP_ERR_NOT_POW2;            // kzg.go:77-79
    if (n2 < 2) return KZG_HIP_ERR_BAD_ARG;                   // kzg.go:80-82
    if (chunk_len > n2 / 2) return KZG_HIP_ERR_BAD_ARG;       // kzg.go:83-85
    if (!is_pow2(chunk_len)) return KZG_HIP_ERR_NOT_POW2;     // kzg.go:86-88
    if (chunk_len < 1) return KZG_HIP_ERR_BAD_ARG;            // kzg.go:89-91
    kzg_hip_fk20m *fk = new kzg_hip_fk20m;
    int st = fk20_core_new(ks, n2, chunk_len, &fk->c);
    if (st) { kzg_hip_fk20_multi_settings_free(fk); return st; }
    *out = fk;
    return KZG_HIP_OK;
}
void kzg_hip_fk20_multi_settings_free(kzg_hip_fk20m *fk) {
    if (!fk) return;
    if (fk->c.ks) { hipSetDevice(fk->c.ks->fs->device); hipDeviceSynchronize(); }
    hipFree(fk->c.d_files); hipFree(fk->c.d_files_fb);
    (void)hipGetLastError();
    delete fk;
}
int kzg_hip_fk20_multi(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (fk->c.ks->fs->W < 2 * n) return KZG_HIP_ERR_TOO_WIDE;     // fk20_multi.go:28-31
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    return fk20_run_host(&fk->c, poly_fr, n, n, 1, 0, 0, 0, out_g1);
}
int kzg_hip_fk20_multi_da_optimized(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n2, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (fk->c.ks->fs->W < n2) return KZG_HIP_ERR_TOO_WIDE;        // fk20_multi.go:60-63
    if (n2 != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    return fk20_run_host(&fk->c, poly_fr, n2, n2 / 2, 1, 1, 1, 0, out_g1);
}
int kzg_hip_da_using_fk20_multi(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;     // fk20_multi.go:115-117
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;                 // fk20_multi.go:118-120
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!coalescing_enabled()) return fk20_run_host(&fk->c, poly_fr, n, n, 1, 0, 1, 1, out_g1);
    return fk20_da_coalesced(&fk->c, poly_fr, n, out_g1);
}
int kzg_hip_da_using_fk20_multi_batch(kzg_hip_fk20m *fk, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!fk || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;     // fk20_multi.go:115-117
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;                 // fk20_multi.go:118-120
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    return fk20_run_host(&fk->c, poly_fr, n, n, batch, 0, 1, 1, out_g1);
}
int kzg_hip_da_using_fk20_multi_batch_dev(kzg_hip_fk20m *fk, const void *d_poly_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream) {
    if (!fk || !d_poly_fr || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > fk->c.ks->fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    dev_guard g(fk->c.ks->fs);
    return fk20_run_dev(&fk->c, (hipStream_t)stream, (const fr *)d_poly_fr, n, n, batch, 1, 1, (g1j *)d_out_g1);
}
int kzg_hip_fk20_multi_hext_slice_dev(kzg_hip_fk20m *fk, const void *d_poly_fr, uint64_t n, uint64_t j0, uint64_t cnt, void *d_out_g1, void *stream) {
    if (!fk || !d_poly_fr || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (2 * n != fk->c.n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (j0 + cnt > 2 * fk->c.k) return KZG_HIP_ERR_BAD_ARG;
    if (!cnt) return KZG_HIP_OK;
    dev_guard g(fk->c.ks->fs);
    return fk20_hext(&fk->c, (hipStream_t)stream, (const fr *)d_poly_fr, n, n, 1, j0, cnt, (g1j *)d_out_g1);
}
int kzg_hip_fk20_multi_finish_dev(kzg_hip_fk20m *fk, const void *d_hext_g1, int bit_reverse, void *d_out_g1, void *stream) {
    if (!fk || !d_hext_g1 || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(fk->c.ks->fs);
    return fk20_finish(&fk->c, (hipStream_t)stream, (const g1j *)d_hext_g1, 1, 1, bit_reverse, (g1j *)d_out_g1);
}
