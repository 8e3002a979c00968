// capi_common.hpp -- shared by the capi_*.hip translation units (the C ABI of include/kzg_hip.h): status / error macros, the handle
// structs behind the opaque pointers, stream leases and stream-ordered temporaries, and the internal pipelines one unit offers another.
// There is deliberately NO CPU fallback anywhere behind this header: without a gfx950 device every constructor returns KZG_HIP_ERR_NO_DEVICE.
#pragma once
#include "../../include/kzg_hip.h"
#include "kzg_hip_internal.h"
#include "internal.hpp"
#include "fr_fft4096.hpp"
#include "fr_das2048.hpp"
#include "coalesce.hpp"
#include "sha256.hpp"

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <thread>
#include <chrono>
#include <condition_variable>
#include <vector>
#include <cstdio>
#include <cstring>
#include <cstdlib>

using namespace kzg;

extern thread_local std::string g_last_error;   // defined in capi_core.hip

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            char buf_[512];                                                                                  \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_last_error = buf_;                                                                             \
            (void)hipGetLastError(); /* consumed: a later hipGetLastError() after a launch must not see it */ \
            return KZG_HIP_ERR_HIP;                                                                          \
        }                                                                                                    \
    } while (0)
#define CHK(expr) do { int s_ = (expr); if (s_ != KZG_HIP_OK) return s_; } while (0)
// no C++ exception may cross the extern "C" boundary (cgo / ctypes callers): host allocations sized by the caller are the
// only throwing operations in this file
#define KZG_TRY try {
#define KZG_CATCH                                                                                             \
    } catch (const std::bad_alloc &) { g_last_error = "host allocation failed"; return KZG_HIP_ERR_HIP; }    \
    catch (const std::exception &e_) { g_last_error = e_.what(); return KZG_HIP_ERR_HIP; }

inline bool is_pow2(uint64_t v) { return (v & (v - 1)) == 0; }   // bls.IsPowerOfTwo (bls/globals.go:72-74): true for 0
inline uint64_t next_pow2(uint64_t v) { if (v == 0) return 1; uint64_t p = 1; while (p < v) p <<= 1; return p; }   // fft.go:11-16
inline uint32_t ilog2(uint64_t v) { uint32_t r = 0; while ((1ull << r) < v) r++; return r; }

// ---------------------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------------------
struct kzg_hip_fft {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned max_scale = 0;
    uint64_t W = 0;
    std::vector<fr> h_expanded, h_reversed;
    fr *d_expanded = nullptr, *d_reversed = nullptr;
    fr *d_expanded_l = nullptr, *d_reversed_l = nullptr;   // the same roots as images 2^261 (the constant operand of fr_lazy.hpp's product): k_fr_fft_upper; W > 4096 only
    fr *d_inv_pow2 = nullptr;   // (2^k)^-1, k = 0..63 (Montgomery)
    uint32_t *d_tw_das2048 = nullptr;              // twiddle file of the lazy-limb DASFFTExtension(2048) (fr_das2048.hpp); null below scale 12
    uint32_t *d_tw4096[2] = {nullptr, nullptr};   // twiddle files of the radix-4 passes, forward / inverse (fr_fft4096.hpp; narrow settings objects: the part their transforms use); null below scale 2
    fr *d_glv_expanded = nullptr, *d_glv_reversed = nullptr;   // twiddles as GLV pairs for the G1 FFT (g1_mul_glv)
    int8_t *d_wnaf_expanded = nullptr, *d_wnaf_reversed = nullptr;   // ... and their width-5 NAF digit strings (KZG_WNAF_ROW bytes per twiddle)
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;   // pinned staging for large results of calls that hold `mu` (d2h_staged)
    std::mutex mu;
    struct pool_slot { hipStream_t s = nullptr; uint8_t *h_pin = nullptr; size_t pin_cap = 0; };   // a stream + its pinned staging area (stream_lease)
    std::mutex pool_mu; std::condition_variable pool_cv; std::vector<pool_slot> pool_idle; int pool_total = 0;
    struct lincomb_promo *promo = nullptr;   // kzg_hip_lincomb_g1's memory of recent caller-supplied point sets (capi_core.hip); created on first use
};
struct kzg_hip_kzg {
    kzg_hip_fft *fs = nullptr;
    uint64_t n_setup = 0;
    g1j *d_secret = nullptr;     // SecretG1, normalised Jacobian images
    g1a *d_secret_a = nullptr;   // affine table for the MSM
    g1a *d_fixed = nullptr;      // fixed-base window table (lazily built)
    msm_plan fixed_plan{};
    double budget_gb = -1.0;             // fixed-base table budget; < 0: default policy (ensure_fixed_table)
    std::atomic<bool> projective{false}; // kzg_hip_kzg_set_projective_outputs: table-walk results leave as Jacobian images with Z != 1 (no inversion per result)
    hipStream_t copy_stream = nullptr;   // uploads of the host-buffer batch entry point, overlapped with the walk of the previous chunk
    hipEvent_t copy_done[2] = {nullptr, nullptr};
    std::unique_ptr<coalescer> co_commit, co_proof;   // merge concurrent one-polynomial calls into batched launches (coalesce.hpp)
    std::shared_mutex tab_mu;      // table lifetime: coalesced batches walk d_fixed outside the handle mutex (shared), kzg_hip_kzg_set_table_budget_gb frees it (unique)
};
struct fk20_core {
    kzg_hip_kzg *ks = nullptr;
    uint64_t n2 = 0, l = 1, k = 0;   // n2 = 2n, chunk length l, k = n / l
    g1j *d_files = nullptr;          // l x 2k points: xExtFFT (single) / xExtFFTFiles (multi)
    g1a *d_files_fb = nullptr;       // fixed-base table over the l x 2k file points (k_fb_mul_vec); null -> double-and-add path
    uint32_t fb_c = 0, fb_nwin = 0; bool fb_glv = false;   // window bits, windows, layout (glv: ceil(128 / c) windows walked by both halves of every scalar)
    std::unique_ptr<coalescer> co_da;   // concurrent DAUsingFK20 / DAUsingFK20Multi calls
};
struct kzg_hip_fk20s { fk20_core c; };
struct kzg_hip_fk20m { fk20_core c; };

struct dev_guard {
    kzg_hip_fft *fs; std::unique_lock<std::mutex> lk;
    explicit dev_guard(kzg_hip_fft *f) : fs(f), lk(f->mu) { hipSetDevice(f->device); }
};

// selects the handle's device for the calling thread (goroutine-backed OS threads start on device 0); no lock: for entry points that only
// read the settings' immutable tables and order their work on a caller-supplied stream
struct dev_select { explicit dev_select(kzg_hip_fft *f) { hipSetDevice(f->device); } };
// A stream of the handle's pool for one host-buffer call (FFT, FFTG1, DASFFTExtension, uncached LinCombG1, conversions, recovery): these only
// read immutable settings tables and allocate their temporaries stream-ordered, so calls from different threads need no common lock and no
// common stream.  Up to POOL_MAX streams per handle, created on demand; further callers wait for one to come back.  If no stream can be
// created at all the call falls back to the handle's stream under its mutex.
struct stream_lease {
    static constexpr int POOL_MAX = 16;
    static constexpr size_t PIN_MAX = 8u << 20;                  // calls that move at most this much go through the slot's pinned staging area
    kzg_hip_fft *fs; hipStream_t s = nullptr; kzg_hip_fft::pool_slot slot; std::unique_lock<std::mutex> fallback;
    explicit stream_lease(kzg_hip_fft *f) : fs(f) {
        hipSetDevice(f->device);
        std::unique_lock<std::mutex> lk(f->pool_mu);
        for (;;) {
            if (!f->pool_idle.empty()) { slot = f->pool_idle.back(); f->pool_idle.pop_back(); s = slot.s; return; }
            if (f->pool_total < POOL_MAX) {
                if (hipStreamCreateWithFlags(&slot.s, hipStreamNonBlocking) == hipSuccess) { stream_cache_own(slot.s); f->pool_total++; s = slot.s; return; }
                (void)hipGetLastError(); slot.s = nullptr;
                if (f->pool_total == 0) { lk.unlock(); fallback = std::unique_lock<std::mutex>(f->mu); s = f->stream; return; }
            }
            f->pool_cv.wait(lk);
        }
    }
    // `bytes` of pinned host memory owned by this call, visible to the device at *dev (zero-copy: a kernel that touches every byte exactly
    // once reads its input and writes its output there, no staged hipMemcpy of pageable memory, no device buffer); null if unavailable
    uint8_t *pinned(size_t bytes, void **dev) {
        if (fallback.owns_lock() || bytes > PIN_MAX) return nullptr;
        if (slot.pin_cap < bytes) {
            if (slot.h_pin) { hipHostFree(slot.h_pin); slot.h_pin = nullptr; slot.pin_cap = 0; }
            size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
            if (hipHostMalloc((void **)&slot.h_pin, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); slot.h_pin = nullptr; return nullptr; }
            slot.pin_cap = cap;
        }
        if (hipHostGetDevicePointer(dev, slot.h_pin, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return slot.h_pin;
    }
    ~stream_lease() {
        if (fallback.owns_lock()) { (void)hipStreamSynchronize(s); return; }
        (void)hipStreamSynchronize(slot.s);   // an early error return may leave kernels in flight that touch the slot's pinned area or stream-ordered temporaries: the next lessee must not see them (free when idle)
        std::lock_guard<std::mutex> lk(fs->pool_mu);
        fs->pool_idle.push_back(slot);
        fs->pool_cv.notify_one();
    }
};
// Device -> pageable host memory for a call that holds fs->mu: through the handle's pinned staging area (grown on demand, at most 16 MiB) and a
// host memcpy.  hipMemcpyAsync into pageable memory changes mechanism above ~4 MiB (the runtime pins the destination on the fly): the 4.7 MB
// of proofs of an 8-polynomial DAUsingFK20 batch took 5 ms longer than the 4.1 MB of a 7-polynomial one.  Synchronises the stream.
inline int d2h_staged(kzg_hip_fft *fs, hipStream_t s, void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes > (16u << 20) || bytes < (1u << 20)) {   // larger results: the runtime's own pinning is cheaper than a second pass over the bytes
        HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return KZG_HIP_OK;
    }
    if (fs->h_stage_cap < bytes) {
        if (fs->h_stage) { hipHostFree(fs->h_stage); fs->h_stage = nullptr; fs->h_stage_cap = 0; }
        size_t cap = 8u << 20;
        while (cap < bytes) cap <<= 1;
        if (hipHostMalloc((void **)&fs->h_stage, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); fs->h_stage = nullptr;
            HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            return KZG_HIP_OK;
        }
        fs->h_stage_cap = cap;
    }
    HIPCHK(hipMemcpyAsync(fs->h_stage, dev_src, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(host_dst, fs->h_stage, bytes);
    return KZG_HIP_OK;
}
// coalesced executors enqueue kernels that read and write a batch's PINNED rows in place: whatever way the executor returns (an error
// status after some kernels were already enqueued included), the stream has drained before the rows are handed back to their callers
struct drain_on_exit {
    hipStream_t s;
    explicit drain_on_exit(hipStream_t st) : s(st) {}
    ~drain_on_exit() { (void)hipStreamSynchronize(s); }
};
// Stream-ordered temporaries.  hipMallocAsync / hipFreeAsync cost the host 2 + 8 us per pair (rocprofv3 --hip-runtime-trace, round 6), and a one-polynomial call makes
// five to seven of them -- the frees AFTER its last synchronisation, i.e. on the caller's critical path: 50 of a lone eth.ComputeKZGProof's 450 us were spent freeing.
// Small blocks (<= 8 MiB) of streams the LIBRARY owns are therefore kept in a per-stream cache of power-of-two size classes and handed to the next temporary of that
// stream: reuse on the same stream is ordered behind every earlier use exactly as the runtime's own pool orders it.  Caller-supplied streams (the _dev entry points) are
// never cached: the library cannot know when they die.  stream_cache_own() at creation, stream_cache_disown() before hipStreamDestroy (frees the stream's blocks).
void stream_cache_own(hipStream_t s);                               // capi_core.hip
void stream_cache_disown(hipStream_t s);
void *stream_cache_take(hipStream_t s, size_t class_bytes);         // nullptr: nothing cached (or not an owned stream)
bool stream_cache_give(hipStream_t s, void *p, size_t class_bytes); // false: not kept (the caller frees)
inline size_t stream_cache_class(size_t bytes) {                    // 0: too large to cache
    if (bytes > (8u << 20)) return 0;
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}
template <class T> struct dtmp {
    T *p = nullptr; hipStream_t s; size_t cls = 0;
    dtmp(hipStream_t st) : s(st) {}
    int alloc(size_t count) {
        if (!count) count = 1;
        size_t bytes = count * sizeof(T);
        cls = stream_cache_class(bytes);
        if (cls) {
            if ((p = (T *)stream_cache_take(s, cls))) return KZG_HIP_OK;
            bytes = cls;                                            // a block of the whole class, so that it can serve the class later
        }
        HIPCHK(hipMallocAsync((void **)&p, bytes, s));
        return KZG_HIP_OK;
    }
    ~dtmp() { if (p && !(cls && stream_cache_give(s, p, cls))) hipFreeAsync(p, s); }
};

// (ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues, default 4: kernels of streams that share a queue run one after
// the other.  Measured with 16 host threads of FFT_Fr(4096) on host buffers: x3.9 of one thread with 4 queues, x5.4 with 8, x5.6 with 16 --
// but with 8 queues a LONE coalesced CommitToPoly takes 1.5 ms instead of 0.41 (the batch stream and the handle stream land on different
// queues), so the library leaves the runtime's default alone; a caller that runs many host-buffer transforms side by side can set it.)

struct kzg_hip_points {
    kzg_hip_fft *fs = nullptr;
    uint64_t n = 0;
    g1a *d_tab = nullptr;          // [P_0 .. P_{n-1} | 2^64 P_0 .. 2^64 P_{n-1}], affine, (0, 0) = inf: the bucket pipeline's rows
    kzg_hip_kzg *ks = nullptr;     // the same points as a settings object: its fixed-base table (built lazily within the set's budget) turns a
                                   // linear combination on the cached set into the table walk of CommitToPoly; null below 64 points
    std::unique_ptr<coalescer> co; // concurrent one-MSM calls (bls.LinCombG1 from many goroutines) merge into batched launches
};
struct kzg_hip_eth {
    kzg_hip_fft *fs = nullptr;
    kzg_hip_kzg *ks = nullptr;     // "SecretG1" = bit-reversed Lagrange setup (kzgSetupLagrange, eth/globals.go:48)
    uint64_t n = 0;
    fr *d_domain = nullptr;        // DomainFr: w^bitrev(i) (eth/globals.go:61-66)
    std::unique_ptr<coalescer> co_blob;   // concurrent one-blob BlobToKZGCommitment calls (eth/eth.go:145-151) merge into batched launches
    std::unique_ptr<coalescer> co_proof;  // concurrent ComputeKZGProof calls (eth/helpers.go:179-203)
};

// ---------------------------------------------------------------------------------------------------------
// pipelines shared between the units (defined in the unit named on the right; C++ linkage, hidden visibility)
// ---------------------------------------------------------------------------------------------------------
void fr_fft_rows(kzg_hip_fft *fs, hipStream_t s, const fr *d_in, uint64_t in_stride, uint64_t n_in, fr *d_out, uint64_t n, uint64_t batch, int inv);   // capi_core.hip
uint32_t g1_fft_direct_logr(uint64_t n, uint64_t batch);   // capi_core.hip
bool g1_fft_direct_mode(uint64_t n, uint64_t batch);   // capi_core.hip
int g1_fft_direct_lanes(uint64_t n, uint64_t batch);   // capi_core.hip
int g1_fft_rows(kzg_hip_fft *fs, hipStream_t s, const g1j *d_in, uint64_t in_stride, uint64_t n_valid, g1j *d_data, uint64_t n, uint64_t batch, int inv,
                           const fr *scale = nullptr, uint64_t n_out = 0);   // capi_core.hip
int das_ext_rows(kzg_hip_fft *fs, hipStream_t s, fr *d, uint64_t n, uint64_t batch);   // capi_core.hip
msm_plan classic_plan(uint64_t n, bool folded = false);   // capi_core.hip
void set_inf_image(void *out_g1);   // capi_core.hip
int kzg_settings_build(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_kzg **out);   // capi_kzg.hip
int ensure_fixed_table(kzg_hip_kzg *ks, hipStream_t);   // capi_kzg.hip
const void *host_mapped_pointer(const void *host, size_t bytes);
int h2d_copy(void *dst, const void *src, size_t bytes, hipStream_t s);   // capi_kzg.hip: cut at the boundaries of registered ranges   // capi_kzg.hip
int commit_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride = 0, bool to_kilic = true);   // capi_kzg.hip
double table_budget_gb(const char *env, double cap_gb, double headroom_gb);   // capi_kzg.hip
int lincomb_points_rows(kzg_hip_points *pts, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride = 0, bool holds_mu = false);   // capi_core.hip
int lincomb_points_coalesced(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1);   // capi_kzg.hip
uint32_t fb_windows(uint32_t c);   // capi_kzg.hip
uint32_t fb_windows_glv(uint32_t c);   // capi_kzg.hip
bool fb_glv_enabled();   // capi_kzg.hip
bool coalescing_enabled();   // capi_kzg.hip
coalescer *get_coalescer(kzg_hip_fft *fs, std::unique_ptr<coalescer> &slot, size_t in_row, size_t out_row, int callers_per_batch = 0);   // capi_kzg.hip
int coalesce_upload_rows(coalesce_buf &b, uint64_t batch, size_t in_row_bytes, uint64_t n_max, fr *d_rows, uint64_t *d_meta);   // capi_kzg.hip
int proof_single_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_poly, uint64_t n, uint64_t batch, const uint64_t *d_x_u64, uint64_t x_stride, g1j *d_out);   // capi_kzg.hip
int fk20_hext(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, uint64_t j0, uint64_t cnt, g1j *d_hext);   // capi_fk20.hip
int fk20_finish(fk20_core *c, hipStream_t s, const g1j *d_hext, uint64_t batch, int da, int bit_reverse, g1j *d_out);   // capi_fk20.hip
int fk20_run_dev(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int da, int bit_reverse, g1j *d_out);   // capi_fk20.hip
