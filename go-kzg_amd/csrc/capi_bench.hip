// capi_bench.hip -- instrumentation for bench.py and the tests: HIP-event records, live VALU calibration, native drop-in drivers (csrc/kzg_hip_internal.h)
#include "capi_common.hpp"

// ---------------------------------------------------------------------------------------------------------
// profiling hook: HIP events around named kernels on the stream they are launched on (bench.py roofline leg)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct prof_rec { std::string name; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<prof_rec> g_prof;
}
namespace kzg {
void prof_begin(hipStream_t s, const char *name) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_rec r; r.name = name;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s);
    g_prof.push_back(r);
}
void prof_end(hipStream_t s, const char *name) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof.size(); i-- > 0;)
        if (g_prof[i].name == name) { hipEventRecord(g_prof[i].e1, s); break; }
}
}

// ---------------------------------------------------------------------------------------------------------
// instrumentation
// ---------------------------------------------------------------------------------------------------------
// bench.py's drop_in leg: `threads` host threads (std::thread, no interpreter lock in the way) each make `calls` blocking
// ONE-polynomial calls to the reference-shaped entry point on host buffers, exactly what a goroutine per blob would do through
// cgo.  op 0: kzg_hip_commit_to_poly, 1: kzg_hip_compute_proof_single (x = 17 + thread).  blobs: nblobs x n Fr; out: threads x G1.
int kzg_hip_bench_drop_in(kzg_hip_kzg *ks, int op, const void *blobs_fr, uint64_t n, uint64_t nblobs, unsigned threads, unsigned calls, void *out_g1,
                          double *seconds) {
    if (!ks || !blobs_fr || !out_g1 || !seconds || !threads || !calls || !nblobs) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<std::thread> ts;
    std::vector<int> status(threads, 0);
    std::mutex mu; std::condition_variable cv; unsigned arrived = 0; bool go = false;
    for (unsigned t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            { std::unique_lock<std::mutex> lk(mu); arrived++; cv.notify_all(); cv.wait(lk, [&] { return go; }); }
            for (unsigned c = 0; c < calls; c++) {
                const uint8_t *in = (const uint8_t *)blobs_fr + ((uint64_t)(t + c) % nblobs) * n * sizeof(fr);
                int st = op == 0 ? kzg_hip_commit_to_poly(ks, in, n, (uint8_t *)out_g1 + (size_t)t * sizeof(g1j))
                                 : kzg_hip_compute_proof_single(ks, in, n, 17 + t, (uint8_t *)out_g1 + (size_t)t * sizeof(g1j));
                if (st) { status[t] = st; break; }
            }
        });
    std::chrono::steady_clock::time_point t0;
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived == threads; }); go = true; t0 = std::chrono::steady_clock::now(); cv.notify_all(); }
    for (auto &th : ts) th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int st : status) if (st) return st;
    return KZG_HIP_OK;
    KZG_CATCH
}
// the same for eth.ComputeKZGProof (eth/helpers.go:179-203): polys = npolys x n Fr (evaluation form), z = 5 + thread (outside the domain);
// out: threads x 48 bytes
int kzg_hip_bench_drop_in_eth_proof(kzg_hip_eth *eth, const void *polys_fr, uint64_t n, uint64_t npolys, unsigned threads, unsigned calls, void *out48, double *seconds) {
    if (!eth || !polys_fr || !out48 || !seconds || !threads || !calls || !npolys) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<std::thread> ts;
    std::vector<int> status(threads, 0);
    std::mutex mu; std::condition_variable cv; unsigned arrived = 0; bool go = false;
    for (unsigned t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            const fr z = fr_from_u64(5 + t);
            { std::unique_lock<std::mutex> lk(mu); arrived++; cv.notify_all(); cv.wait(lk, [&] { return go; }); }
            for (unsigned c = 0; c < calls; c++) {
                const uint8_t *in = (const uint8_t *)polys_fr + ((uint64_t)(t + c) % npolys) * n * sizeof(fr);
                int st = kzg_hip_eth_compute_kzg_proof(eth, in, n, &z, (uint8_t *)out48 + (size_t)t * 48, nullptr);
                if (st) { status[t] = st; break; }
            }
        });
    std::chrono::steady_clock::time_point t0;
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived == threads; }); go = true; t0 = std::chrono::steady_clock::now(); cv.notify_all(); }
    for (auto &th : ts) th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int st : status) if (st) return st;
    return KZG_HIP_OK;
    KZG_CATCH
}
// bls.PolyLinComb over device-resident rows (bls/globals.go:155-178): out[i] = sum_c scalars[c] * vectors[c * stride + i].  bench.py's check of
// EVERY output of a timed step needs the random linear combination of the step's input polynomials; this is the kernel the eth aggregation uses.
int kzg_hip_bench_poly_lincomb_dev(kzg_hip_fft *fs, const void *d_vectors_fr, uint64_t stride, const void *d_scalars_fr, uint64_t count, uint64_t n, void *d_out_fr, void *stream) {
    if (!fs || !d_vectors_fr || !d_scalars_fr || !d_out_fr || !n) return KZG_HIP_ERR_BAD_ARG;
    dev_select sel(fs);
    launch_poly_lincomb((hipStream_t)stream, (const fr *)d_vectors_fr, stride, (const fr *)d_scalars_fr, count, n, (fr *)d_out_fr);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
// the same for the host-buffer (I)FFT over F_r (fft_fr.go:55-74): `threads` host threads x `calls` blocking kzg_hip_fft_fr calls of n values each
// (thread t transforms vals[t % nrows]); out: threads x n Fr (each thread's last result)
int kzg_hip_bench_threads_fft_fr(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, uint64_t nrows, unsigned threads, unsigned calls, void *out_fr, double *seconds) {
    if (!fs || !vals_fr || !out_fr || !seconds || !threads || !calls || !nrows) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<std::thread> ts;
    std::vector<int> status(threads, 0);
    std::mutex mu; std::condition_variable cv; unsigned arrived = 0; bool go = false;
    for (unsigned t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            { std::unique_lock<std::mutex> lk(mu); arrived++; cv.notify_all(); cv.wait(lk, [&] { return go; }); }
            const uint8_t *in = (const uint8_t *)vals_fr + ((uint64_t)t % nrows) * n * sizeof(fr);
            for (unsigned c = 0; c < calls; c++) {
                uint64_t on = 0;
                int st = kzg_hip_fft_fr(fs, in, n, 0, (uint8_t *)out_fr + (size_t)t * n * sizeof(fr), &on);
                if (st) { status[t] = st; break; }
            }
        });
    std::chrono::steady_clock::time_point t0;
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived == threads; }); go = true; t0 = std::chrono::steady_clock::now(); cv.notify_all(); }
    for (auto &th : ts) th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int st : status) if (st) return st;
    return KZG_HIP_OK;
    KZG_CATCH
}
// ---- live calibration of the instruction rates that bound the integer kernels (bench.py: roofline.mac).  No figure for the
// v_mad_u64_u32 rate is in the local guides, so it is measured on the GPU the bench runs on: 8 independent chains per lane, every
// SIMD holding 8 waves, ~4 ms per kernel.  Same loops as tools/microbench.hip.
#define CAL_ITERS 2048
__global__ __launch_bounds__(256) void k_cal_mad(uint32_t *out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[8]; uint32_t a = seed + t, b = seed * 3 + t;
    for (int c = 0; c < 8; c++) x[c] = seed + c + t;
    for (int i = 0; i < CAL_ITERS; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc");
    }
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc ^= (uint32_t)x[c] ^ (uint32_t)(x[c] >> 32);
    out[t] = acc;
}
__global__ __launch_bounds__(256) void k_cal_add(uint32_t *out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[8]; uint32_t a = seed + t;
    for (int c = 0; c < 8; c++) x[c] = seed + c + t;
    for (int i = 0; i < CAL_ITERS; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
    }
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc ^= x[c];
    out[t] = acc;
}
__global__ __launch_bounds__(256) void k_cal_fp_mul(fp *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fq x = unpackq(io[t]), y = unpackq(io[t ^ 1]);
    for (int i = 0; i < iters; i++) { x = mulq_inl(x, y); y = mulq_inl(y, x); }
    io[t] = packq(addq(x, y));
}
// lane-operations per second of v_mad_u64_u32 and v_add_u32, and lazy 13-limb F_p products per second (mont_core30), on `fs`'s device
int kzg_hip_calibrate(kzg_hip_fft *fs, double *mad_per_s, double *add_per_s, double *fp_mul_per_s) {
    if (!fs || !mad_per_s || !add_per_s || !fp_mul_per_s) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, fs->device);
    const int blocks = cus * 8, threads = 256;
    dtmp<uint32_t> d(s); dtmp<fp> dfp(s);
    CHK(d.alloc((size_t)blocks * threads)); CHK(dfp.alloc((size_t)blocks * threads));
    HIPCHK(hipMemsetAsync(dfp.p, 0x11, (size_t)blocks * threads * sizeof(fp), s));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int which) -> double {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {                  // first repetition warms up, the best of the rest counts
            hipEventRecord(e0, s);
            if (which == 0) hipLaunchKernelGGL(k_cal_mad, dim3(blocks), dim3(threads), 0, s, d.p, 12345u);
            else if (which == 1) hipLaunchKernelGGL(k_cal_add, dim3(blocks), dim3(threads), 0, s, d.p, 12345u);
            else hipLaunchKernelGGL(k_cal_fp_mul, dim3(blocks), dim3(threads), 0, s, dfp.p, 64);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        return (double)best * 1e-3;
    };
    const double lanes = (double)blocks * threads;
    *mad_per_s = lanes * CAL_ITERS * 8 / timed(0);
    *add_per_s = lanes * CAL_ITERS * 8 / timed(1);
    *fp_mul_per_s = lanes * 64 * 2 / timed(2);
    hipEventDestroy(e0); hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_test_fp_inv(kzg_hip_fft *fs, const void *in_fp, uint64_t n, void *out_coop, void *out_lane, double *ms_coop, double *ms_lane) {
    if (!fs || !in_fp || !n || n > (1u << 20)) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    dtmp<fp> d_in(s), d_a(s), d_b(s);
    CHK(d_in.alloc(n)); CHK(d_a.alloc(n)); CHK(d_b.alloc(n));
    HIPCHK(hipMemcpyAsync(d_in.p, in_fp, n * sizeof(fp), hipMemcpyHostToDevice, s));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        void *out = which ? out_lane : out_coop; double *ms = which ? ms_lane : ms_coop;
        if (!out) continue;
        hipEventRecord(e0, s);
        launch_fp_inv_both(s, d_in.p, d_a.p, d_b.p, n, which ? 2 : 1);
        hipEventRecord(e1, s);
        HIPCHK(hipMemcpyAsync(out, which ? d_b.p : d_a.p, n * sizeof(fp), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float t = 0; hipEventElapsedTime(&t, e0, e1);
        if (ms) *ms = t;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_test_fr_inv(kzg_hip_fft *fs, const void *in_fr, uint64_t n, void *out_coop, void *out_lane, void *out_block) {
    if (!fs || !in_fr || !n || n > (1u << 20) || !out_coop || !out_lane || !out_block) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    dtmp<fr> d_in(s), d_a(s), d_b(s), d_c(s);
    CHK(d_in.alloc(n)); CHK(d_a.alloc(n)); CHK(d_b.alloc(n)); CHK(d_c.alloc(n));
    HIPCHK(hipMemcpyAsync(d_in.p, in_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    launch_fr_inv_test(s, d_in.p, n, d_a.p, d_b.p, d_c.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_coop, d_a.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_lane, d_b.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_block, d_c.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_coalesce_stats(kzg_hip_kzg *ks, int op, uint64_t out[8]) {
    if (!ks || !out || op < 0 || op > 1) return KZG_HIP_ERR_BAD_ARG;
    for (int i = 0; i < 8; i++) out[i] = 0;
    std::lock_guard<std::mutex> lk(ks->fs->mu);                  // get_coalescer creates the object under the settings' mutex
    coalescer *co = op == 0 ? ks->co_commit.get() : ks->co_proof.get();
    if (co) co->stats(out);
    return KZG_HIP_OK;
}
void kzg_hip_prof_reset(kzg_hip_fft *fs, int enable) {
    (void)fs;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = enable != 0;
}
int kzg_hip_prof_read(kzg_hip_fft *fs, const char *kernel, double *total_ms, uint64_t *launches) {
    if (!fs || !kernel) return KZG_HIP_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0; uint64_t cnt = 0;
    for (auto &r : g_prof) {
        if (r.name != kernel) continue;
        HIPCHK(hipEventSynchronize(r.e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        tot += ms; cnt++;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    return KZG_HIP_OK;
}
