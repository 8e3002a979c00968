// capi.hip -- the C ABI of include/kzg_hip.h: argument checks with the reference's error behaviour, device-resident
// settings handles, and the pipelines that chain the kernels of k_fr.hip / k_g1.hip / k_msm.hip.
// There is deliberately NO CPU fallback: without a gfx950 device every constructor returns KZG_HIP_ERR_NO_DEVICE.
#include "../../include/kzg_hip.h"
#include "internal.hpp"
#include "fr_fft4096.hpp"
#include "fr_das2048.hpp"
#include "coalesce.hpp"
#include "sha256.hpp"

#include <algorithm>
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

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            char buf_[512];                                                                                  \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_last_error = buf_;                                                                             \
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

static bool is_pow2(uint64_t v) { return (v & (v - 1)) == 0; }   // bls.IsPowerOfTwo (bls/globals.go:72-74): true for 0
static uint64_t next_pow2(uint64_t v) { if (v == 0) return 1; uint64_t p = 1; while (p < v) p <<= 1; return p; }   // fft.go:11-16
static uint32_t ilog2(uint64_t v) { uint32_t r = 0; while ((1ull << r) < v) r++; return r; }

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
};
struct kzg_hip_kzg {
    kzg_hip_fft *fs = nullptr;
    uint64_t n_setup = 0;
    g1j *d_secret = nullptr;     // SecretG1, normalised Jacobian images
    g1a *d_secret_a = nullptr;   // affine table for the MSM
    g1a *d_fixed = nullptr;      // fixed-base window table (lazily built)
    msm_plan fixed_plan{};
    double budget_gb = -1.0;             // fixed-base table budget; < 0: default policy (ensure_fixed_table)
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
    uint32_t fb_c = 0, fb_nwin = 0;
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
                if (hipStreamCreateWithFlags(&slot.s, hipStreamNonBlocking) == hipSuccess) { f->pool_total++; s = slot.s; return; }
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
        if (fallback.owns_lock()) return;
        std::lock_guard<std::mutex> lk(fs->pool_mu);
        fs->pool_idle.push_back(slot);
        fs->pool_cv.notify_one();
    }
};
// Device -> pageable host memory for a call that holds fs->mu: through the handle's pinned staging area (grown on demand, at most 16 MiB) and a
// host memcpy.  hipMemcpyAsync into pageable memory changes mechanism above ~4 MiB (the runtime pins the destination on the fly): the 4.7 MB
// of proofs of an 8-polynomial DAUsingFK20 batch took 5 ms longer than the 4.1 MB of a 7-polynomial one.  Synchronises the stream.
static int d2h_staged(kzg_hip_fft *fs, hipStream_t s, void *host_dst, const void *dev_src, size_t bytes) {
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
// stream-ordered temporary
template <class T> struct dtmp {
    T *p = nullptr; hipStream_t s;
    dtmp(hipStream_t st) : s(st) {}
    int alloc(size_t count) {
        if (!count) count = 1;
        HIPCHK(hipMallocAsync((void **)&p, count * sizeof(T), s));
        return KZG_HIP_OK;
    }
    ~dtmp() { if (p) hipFreeAsync(p, s); }
};

// (ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues, default 4: kernels of streams that share a queue run one after
// the other.  Measured with 16 host threads of FFT_Fr(4096) on host buffers: x3.9 of one thread with 4 queues, x5.4 with 8, x5.6 with 16 --
// but with 8 queues a LONE coalesced CommitToPoly takes 1.5 ms instead of 0.41 (the batch stream and the handle stream land on different
// queues), so the library leaves the runtime's default alone; a caller that runs many host-buffer transforms side by side can set it.)

extern "C" {

int kzg_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}
const char *kzg_hip_last_error(void) { return g_last_error.c_str(); }
const char *kzg_hip_version(void) { return "kzg_hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------------------
// FFTSettings
// ---------------------------------------------------------------------------------------------------------
static fr scale2_root_of_unity(unsigned k) {   // 7^((r-1)/2^k), bls/globals.go:24-60
    uint32_t e[8]; uint32_t br = 0;
    for (int i = 0; i < 8; i++) e[i] = subb(FrP::mod(i), i == 0 ? 1u : 0u, br);
    for (unsigned s = 0; s < k; s++)
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? e[i + 1] << 31 : 0);
    fr seven = fr_from_u64(7), acc = one<FrP>();
    for (int i = 255; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i / 32] >> (i % 32)) & 1u) acc = mul(acc, seven);
    }
    return acc;
}

// G1-FFT twiddles leave Montgomery form (Kilic FromRed) and are split k = k2 lambda + k1 once, on the host
static int upload_g1_twiddles(kzg_hip_fft *fs) {
    size_t bytes = (fs->W + 1) * sizeof(fr);
    std::vector<fr> ge(fs->W + 1), gr(fs->W + 1);
    for (uint64_t i = 0; i <= fs->W; i++) ge[i] = glv_decompose(from_mont<FrP>(fs->h_expanded[i]));
    for (uint64_t i = 0; i <= fs->W; i++) gr[i] = ge[fs->W - i];
    HIPCHK(hipMalloc((void **)&fs->d_glv_expanded, bytes));
    HIPCHK(hipMalloc((void **)&fs->d_glv_reversed, bytes));
    HIPCHK(hipMemcpy(fs->d_glv_expanded, ge.data(), bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(fs->d_glv_reversed, gr.data(), bytes, hipMemcpyHostToDevice));
    // the recoding the stage kernels would otherwise repeat per butterfly (130 steps per half): once per twiddle, here
    std::vector<int8_t> we((fs->W + 1) * KZG_WNAF_ROW), wr((fs->W + 1) * KZG_WNAF_ROW);
    for (uint64_t i = 0; i <= fs->W; i++) glv_wnaf5_row(ge[i], &we[i * KZG_WNAF_ROW]);
    for (uint64_t i = 0; i <= fs->W; i++) memcpy(&wr[i * KZG_WNAF_ROW], &we[(fs->W - i) * KZG_WNAF_ROW], KZG_WNAF_ROW);
    HIPCHK(hipMalloc((void **)&fs->d_wnaf_expanded, we.size()));
    HIPCHK(hipMalloc((void **)&fs->d_wnaf_reversed, wr.size()));
    HIPCHK(hipMemcpy(fs->d_wnaf_expanded, we.data(), we.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(fs->d_wnaf_reversed, wr.data(), wr.size(), hipMemcpyHostToDevice));
    return KZG_HIP_OK;
}
static bool device_is_gfx950(int device) {
    hipDeviceProp_t pr;
    return hipGetDeviceProperties(&pr, device) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0;
}
int kzg_hip_fft_settings_new(int device, unsigned max_scale, kzg_hip_fft **out) {
    if (!out || max_scale > 31) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0 || device >= ndev) return KZG_HIP_ERR_NO_DEVICE;
    if (!device_is_gfx950(device)) return KZG_HIP_ERR_NO_DEVICE;   // kernels are built for gfx950 only; there is no fallback
    HIPCHK(hipSetDevice(device));
    {   // every pipeline allocates its temporaries stream-ordered (hipMallocAsync): keep freed blocks in the device's pool instead of
        // returning them to the driver at each synchronisation (release threshold 0 is the default and costs ~0.1 ms per call)
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
            uint64_t keep = 8ull << 30;                          // up to 8 GiB of idle temporaries stay cached
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    KZG_TRY
    std::unique_ptr<kzg_hip_fft, void (*)(kzg_hip_fft *)> own(new kzg_hip_fft, kzg_hip_fft_settings_free);   // frees on every error path
    kzg_hip_fft *fs = own.get();
    fs->device = device; fs->max_scale = max_scale; fs->W = 1ull << max_scale;
    HIPCHK(hipStreamCreateWithFlags(&fs->stream, hipStreamNonBlocking));
    // expandRootOfUnity (fft.go:21-32): W + 1 powers, first and last are 1; reversed copy (fft.go:49-54)
    fr w = scale2_root_of_unity(max_scale);
    fs->h_expanded.resize(fs->W + 1); fs->h_reversed.resize(fs->W + 1);
    fs->h_expanded[0] = one<FrP>();
    for (uint64_t i = 1; i <= fs->W; i++) fs->h_expanded[i] = mul(fs->h_expanded[i - 1], w);
    for (uint64_t i = 0; i <= fs->W; i++) fs->h_reversed[i] = fs->h_expanded[fs->W - i];
    size_t bytes = (fs->W + 1) * sizeof(fr);
    HIPCHK(hipMalloc((void **)&fs->d_expanded, bytes));
    HIPCHK(hipMalloc((void **)&fs->d_reversed, bytes));
    HIPCHK(hipMemcpy(fs->d_expanded, fs->h_expanded.data(), bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(fs->d_reversed, fs->h_reversed.data(), bytes, hipMemcpyHostToDevice));
    CHK(upload_g1_twiddles(fs));
    fr invs[64]; fr half = inv<FrP>(fr_from_u64(2));
    invs[0] = one<FrP>();
    for (int i = 1; i < 64; i++) invs[i] = mul(invs[i - 1], half);
    HIPCHK(hipMalloc((void **)&fs->d_inv_pow2, sizeof invs));
    HIPCHK(hipMemcpy(fs->d_inv_pow2, invs, sizeof invs, hipMemcpyHostToDevice));
    if (fs->W >= 4) {   // the twiddle file of the radix-4 passes (narrow settings objects get the part their transforms use)
        std::vector<uint32_t> tw(fr4::TW_WORDS);
        for (int dir = 0; dir < 2; dir++) {
            fr4::build_twiddles(dir ? fs->h_reversed.data() : fs->h_expanded.data(), fs->W, tw.data());
            HIPCHK(hipMalloc((void **)&fs->d_tw4096[dir], tw.size() * 4));
            HIPCHK(hipMemcpy(fs->d_tw4096[dir], tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
        }
    }
    if (fs->W >= fr4::N) {
        if (fs->W > fr4::N) {   // transforms above 4096 points: the roots once more, pre-scaled for the lazy-limb product
            std::vector<fr> le(fs->W + 1), lr(fs->W + 1);
            const fr k32 = fr_from_u64(32);
            for (uint64_t i = 0; i <= fs->W; i++) le[i] = mul(fs->h_expanded[i], k32);
            for (uint64_t i = 0; i <= fs->W; i++) lr[i] = le[fs->W - i];
            HIPCHK(hipMalloc((void **)&fs->d_expanded_l, bytes));
            HIPCHK(hipMalloc((void **)&fs->d_reversed_l, bytes));
            HIPCHK(hipMemcpy(fs->d_expanded_l, le.data(), bytes, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(fs->d_reversed_l, lr.data(), bytes, hipMemcpyHostToDevice));
        }
        std::vector<uint32_t> td(das2k::TW_WORDS);
        das2k::build_twiddles(fs->h_expanded.data(), fs->h_reversed.data(), fs->W, td.data());
        HIPCHK(hipMalloc((void **)&fs->d_tw_das2048, td.size() * 4));
        HIPCHK(hipMemcpy(fs->d_tw_das2048, td.data(), td.size() * 4, hipMemcpyHostToDevice));
    }
    *out = own.release();
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_fft_settings_free(kzg_hip_fft *fs) {
    if (!fs) return;
    hipSetDevice(fs->device);
    if (fs->stream) hipStreamSynchronize(fs->stream);
    hipFree(fs->d_expanded); hipFree(fs->d_reversed); hipFree(fs->d_expanded_l); hipFree(fs->d_reversed_l); hipFree(fs->d_inv_pow2); hipFree(fs->d_tw4096[0]); hipFree(fs->d_tw4096[1]); hipFree(fs->d_tw_das2048); hipFree(fs->d_glv_expanded); hipFree(fs->d_glv_reversed); hipFree(fs->d_wnaf_expanded); hipFree(fs->d_wnaf_reversed);
    if (fs->stream) hipStreamDestroy(fs->stream);
    if (fs->h_stage) hipHostFree(fs->h_stage);
    for (auto &ps : fs->pool_idle) { hipStreamSynchronize(ps.s); hipStreamDestroy(ps.s); if (ps.h_pin) hipHostFree(ps.h_pin); }
    (void)hipGetLastError();
    delete fs;
}
uint64_t kzg_hip_fft_max_width(const kzg_hip_fft *fs) { return fs ? fs->W : 0; }
int kzg_hip_fft_roots(const kzg_hip_fft *fs, int reversed, void *out_fr) {
    if (!fs || !out_fr) return KZG_HIP_ERR_BAD_ARG;
    memcpy(out_fr, reversed ? fs->h_reversed.data() : fs->h_expanded.data(), (fs->W + 1) * sizeof(fr));
    return KZG_HIP_OK;
}

// device-side (I)FFT over F_r on resident rows
static void fr_fft_rows(kzg_hip_fft *fs, hipStream_t s, const fr *d_in, uint64_t in_stride, uint64_t n_in, fr *d_out, uint64_t n, uint64_t batch, int inv) {
    launch_fr_fft(s, d_in, in_stride, n_in, d_out, n, batch, inv ? fs->d_reversed : fs->d_expanded, fs->W, inv ? fs->d_inv_pow2 + ilog2(n) : nullptr,
                  fs->d_tw4096[inv ? 1 : 0], inv ? fs->d_reversed_l : fs->d_expanded_l);
}

static int fft_fr_impl(kzg_hip_fft *fs, const void *vals, uint64_t n_in, uint64_t n, uint64_t batch, int inv, void *out) {
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    const size_t in_bytes = n_in * batch * sizeof(fr), out_bytes = n * batch * sizeof(fr);
    void *dp = nullptr;
    uint8_t *hp = n <= 4096 ? lease.pinned(in_bytes + out_bytes, &dp) : nullptr;
    if (hp) {   // LDS-resident transforms read every input and write every output exactly once: straight from / to pinned host memory
        if (in_bytes) memcpy(hp, vals, in_bytes);
        fr_fft_rows(fs, s, (const fr *)dp, n_in, n_in, (fr *)((uint8_t *)dp + in_bytes), n, batch, inv);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        memcpy(out, hp + in_bytes, out_bytes);
        return KZG_HIP_OK;
    }
    dtmp<fr> d_in(s), d_out(s);
    CHK(d_in.alloc(n_in * batch)); CHK(d_out.alloc(n * batch));
    if (n_in) HIPCHK(hipMemcpyAsync(d_in.p, vals, n_in * batch * sizeof(fr), hipMemcpyHostToDevice, s));
    fr_fft_rows(fs, s, d_in.p, n_in, n_in, d_out.p, n, batch, inv);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, n * batch * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_fft_fr(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, int inv, void *out_fr, uint64_t *out_n) {
    if (!fs || !out_fr || (!vals_fr && n)) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;          // fft_fr.go:57-59
    uint64_t np = next_pow2(n);                          // fft_fr.go:60
    if (out_n) *out_n = np;
    return fft_fr_impl(fs, vals_fr, n, np, 1, inv, out_fr);
}
int kzg_hip_inplace_fft_fr(kzg_hip_fft *fs, const void *vals_fr, void *out_fr, uint64_t n, int inv) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;          // fft_fr.go:78-80
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;        // fft_fr.go:81-83
    if (n == 0) return KZG_HIP_OK;
    if (!vals_fr || !out_fr) return KZG_HIP_ERR_BAD_ARG;
    return fft_fr_impl(fs, vals_fr, n, n, 1, inv, out_fr);
}
int kzg_hip_fft_fr_batch(kzg_hip_fft *fs, const void *vals_fr, uint64_t n, uint64_t batch, int inv, void *out_fr) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n == 0 || batch == 0) return KZG_HIP_OK;
    if (!vals_fr || !out_fr) return KZG_HIP_ERR_BAD_ARG;
    return fft_fr_impl(fs, vals_fr, n, n, batch, inv, out_fr);
}

// G1 FFT on resident rows: in (row stride in_stride, first n_valid entries used, the rest = inf) -> data (batch x n).
// scale: nullptr, or a device Fr every output is multiplied by (the n^-1 of the inverse transform, fft_g1.go:72-85); FK20 callers
// fold their scale into the Toeplitz coefficients instead.  Few butterflies (a lone transform) take the direct radix-16 passes,
// whose latency is log16(n) scalar multiplications; batches take the radix-2 network, which does 7.5 times less work.
// Radix of the direct passes for `batch` transforms of n points, as log2: 16 while 16 n batch lanes fit the resident wavefronts twice
// over (n batch <= 8192: one or two 4096-point transforms, 3 passes), 8 up to n batch = 16384 (3-4 transforms: 4 passes of 131 072 lanes:
// 12 ms per transform against 19 ms for the 12 launches of the radix-2 network, which are one scalar-multiplication latency each);
// 0 = the radix-2 network (larger batches fill the chip per stage).  KZG_HIP_G1_FFT = "direct" / "radix2" forces a path (A/B runs).
static uint32_t g1_fft_direct_logr(uint64_t n, uint64_t batch) {
    static const int forced = [] { const char *e = getenv("KZG_HIP_G1_FFT"); return !e ? 0 : (e[0] == 'd' ? 1 : 2); }();   // (initialised once, thread-safe)
    if (forced) return forced == 1 ? 4u : 0u;
    if (n < 2) return 0;
    // with four lanes per butterfly (g1_quad.hpp) the radix-2 network beats the direct passes from two transforms on (DAUsingFK20 on 2 / 4 polynomials:
    // 20.8 / 21.0 ms against 23.3 / 30.5 ms); a lone transform stays direct (15.1 ms against 20.7 ms)
    // (the passes themselves run on quads or pairs where that leaves no SIMD with two wavefronts, i.e. up to 2048 points: g1_fft_direct_lanes; 4096 points on
    // pairs would be four radix-8 passes of 1.8 ms, the same 7.1 ms as three radix-16 passes of 2.4 ms on single lanes: measured, not used)
    if (g1_quad_enabled()) return n * batch <= 4096 ? 4 : 0;
    if (n * batch <= 8192) return 4;
    if (n * batch <= 16384) return 3;
    return 0;
}
static bool g1_fft_direct_mode(uint64_t n, uint64_t batch) { return g1_fft_direct_logr(n, batch) != 0; }
// lanes per (output, term) of a direct pass: as many as keep the pass at one wavefront per SIMD (65 536 lanes)
static int g1_fft_direct_lanes(uint64_t n, uint64_t batch) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_G1_DIRECT_COOP"); return e && e[0] == '0'; }();
    if (!g1_quad_enabled() || off) return 1;
    const uint64_t items = (n * batch) << g1_fft_direct_logr(n, batch);
    return items * 4 <= 65536 ? 4 : items * 2 <= 65536 ? 2 : 1;
}
// (n_out: the caller only reads the first n_out outputs -- the direct passes then skip the rest of their last pass; 0 = all)
static int g1_fft_rows(kzg_hip_fft *fs, hipStream_t s, const g1j *d_in, uint64_t in_stride, uint64_t n_valid, g1j *d_data, uint64_t n, uint64_t batch, int inv,
                       const fr *scale = nullptr, uint64_t n_out = 0) {
    if (g1_fft_direct_mode(n, batch)) {
        dtmp<g1j> d_tmp(s);
        CHK(d_tmp.alloc(n * batch));
        launch_g1_fft_direct(s, d_in, in_stride, n_valid, d_data, d_tmp.p, n, batch, inv ? fs->d_reversed : fs->d_expanded, fs->W, scale, g1_fft_direct_logr(n, batch),
                             g1_fft_direct_lanes(n, batch), 0, n_out);
        return KZG_HIP_OK;
    }
    launch_g1_bitrev_copy(s, d_in, in_stride, n_valid, d_data, n, batch);
    const fr *roots = inv ? fs->d_glv_reversed : fs->d_glv_expanded;
    const int8_t *wnaf = inv ? fs->d_wnaf_reversed : fs->d_wnaf_expanded;
    for (uint64_t m = 1; m < n; m <<= 1) launch_g1_fft_stage(s, d_data, n, batch, m, roots, wnaf, fs->W);
    if (scale) launch_g1_mul_vec(s, d_data, n * batch, scale, 0, n * batch, d_data);
    return KZG_HIP_OK;
}

int kzg_hip_fft_g1(kzg_hip_fft *fs, const void *vals_g1, uint64_t n, int inv, void *out_g1) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;          // fft_g1.go:60-62
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;        // fft_g1.go:63-65
    if (n == 0 || !vals_g1 || !out_g1) return KZG_HIP_ERR_BAD_ARG;   // n == 0: the reference divides by zero (fft_g1.go:76)
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<g1j> d_in(s), d_data(s);
    CHK(d_in.alloc(n)); CHK(d_data.alloc(n));
    HIPCHK(hipMemcpyAsync(d_in.p, vals_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_in.p, n);
    CHK(g1_fft_rows(fs, s, d_in.p, n, n, d_data.p, n, 1, inv, inv ? fs->d_inv_pow2 + ilog2(n) : nullptr));   // fft_g1.go:72-85: inverse: every output times n^-1
    launch_g1_normalize(s, d_data.p, d_in.p, n, true);
    std::swap(d_in.p, d_data.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_data.p, n * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}

// DASFFTExtension over resident rows (in place).  In a settings object of exactly twice the row length -- the only width at which the reference's
// recursion (das_extension.go:7-84, which always walks the FULL-width tables) computes the extension -- rows of 4096 values and more go
// through the lazy-limb transforms: coefficients (inverse transform), x -> w_2n x (one product per coefficient), values again.  The odd-index
// evaluations are unique, so this is the reference's result bit for bit.  Other widths and sizes: the recursion itself, stage by stage.
static int das_ext_rows(kzg_hip_fft *fs, hipStream_t s, fr *d, uint64_t n, uint64_t batch) {
    static const bool radix2_forced = [] { const char *e = getenv("KZG_HIP_FR_FFT"); return e && !strcmp(e, "radix2"); }();
    // (... and launches of 2^20 values in rows of at most 64: the short transforms share workgroups, k_fr_fft_small)
    const bool long_rows = n >= fr4::N && n <= 16 * (uint64_t)fr4::N, short_rows = n >= 4 && n <= 64 && n * batch >= (256ull * fr4::N);   // (measured: 8 values 3.8 -> 0.5 ns, 64 values 8.7 -> 5.7 ns per row; no gain from 128 on)
    if (2 * n == fs->W && (long_rows || short_rows) && fs->d_tw4096[0] && !radix2_forced) {
        dtmp<fr> d_c(s);
        CHK(d_c.alloc(n * batch));
        fr_fft_rows(fs, s, d, n, n, d_c.p, n, batch, 1);
        launch_fr_mul_table_rows(s, d_c.p, fs->d_expanded, 1, n, batch);
        fr_fft_rows(fs, s, d_c.p, n, n, d, n, batch, 0);
        return KZG_HIP_OK;
    }
    launch_das_ext(s, d, n, batch, fs->d_expanded, fs->d_reversed, fs->W, fs->d_inv_pow2 + ilog2(n), fs->d_tw_das2048);
    return KZG_HIP_OK;
}

int kzg_hip_das_fft_extension_batch(kzg_hip_fft *fs, void *vals_fr, uint64_t n, uint64_t batch) {
    if (!fs || !vals_fr) return KZG_HIP_ERR_BAD_ARG;
    if (n * 2 > fs->W) return KZG_HIP_ERR_TOO_WIDE;      // panic das_extension.go:72-74
    if (n < 2 || !is_pow2(n)) return KZG_HIP_ERR_BAD_ARG; // "bad usage" das_extension.go:22-24
    if (!batch) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    void *dp = nullptr;
    uint8_t *hp = n <= 4096 ? lease.pinned(n * batch * sizeof(fr), &dp) : nullptr;
    if (hp) {   // the LDS-resident kernel reads and writes each value once: in place in pinned host memory
        memcpy(hp, vals_fr, n * batch * sizeof(fr));
        CHK(das_ext_rows(fs, s, (fr *)dp, n, batch));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        memcpy(vals_fr, hp, n * batch * sizeof(fr));
        return KZG_HIP_OK;
    }
    dtmp<fr> d(s);
    CHK(d.alloc(n * batch));
    HIPCHK(hipMemcpyAsync(d.p, vals_fr, n * batch * sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(das_ext_rows(fs, s, d.p, n, batch));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(vals_fr, d.p, n * batch * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_fft_fr_batch_dev(kzg_hip_fft *fs, const void *d_vals_fr, uint64_t n, uint64_t batch, int inv, void *d_out_fr, void *stream) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n == 0 || batch == 0) return KZG_HIP_OK;
    if (!d_vals_fr || !d_out_fr) return KZG_HIP_ERR_BAD_ARG;
    dev_select sel(fs);       // the caller's stream orders the work; settings tables are read-only
    fr_fft_rows(fs, (hipStream_t)stream, (const fr *)d_vals_fr, n, n, (fr *)d_out_fr, n, batch, inv);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_fft_g1_batch_dev(kzg_hip_fft *fs, const void *d_vals_g1, uint64_t n, uint64_t batch, int inv, void *d_out_g1, void *stream) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n == 0 || !d_vals_g1 || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    dev_select sel(fs);       // the caller's stream orders the work; settings tables are read-only
    hipStream_t s = (hipStream_t)stream;
    dtmp<g1j> d_in(s), d_data(s);
    CHK(d_in.alloc(n * batch)); CHK(d_data.alloc(n * batch));
    HIPCHK(hipMemcpyAsync(d_in.p, d_vals_g1, n * batch * sizeof(g1j), hipMemcpyDeviceToDevice, s));
    launch_g1_from_kilic(s, d_in.p, n * batch);
    CHK(g1_fft_rows(fs, s, d_in.p, n, n, d_data.p, n, batch, inv, inv ? fs->d_inv_pow2 + ilog2(n) : nullptr));
    launch_g1_normalize(s, d_data.p, (g1j *)d_out_g1, n * batch, true);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_das_fft_extension_batch_dev(kzg_hip_fft *fs, void *d_vals_fr, uint64_t n, uint64_t batch, void *stream) {
    if (!fs || !d_vals_fr) return KZG_HIP_ERR_BAD_ARG;
    if (n * 2 > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    if (n < 2 || !is_pow2(n)) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    dev_select sel(fs);       // the caller's stream orders the work; settings tables are read-only
    CHK(das_ext_rows(fs, (hipStream_t)stream, (fr *)d_vals_fr, n, batch));
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_das_fft_extension(kzg_hip_fft *fs, void *vals_fr, uint64_t n) { return kzg_hip_das_fft_extension_batch(fs, vals_fr, n, 1); }

// ---------------------------------------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------------------------------------
// bucket-MSM plan: signed 8-bit windows over the GLV halves (k_msm.hip); `folded`: the table also holds the 2^64 multiples
static msm_plan classic_plan(uint64_t n, bool folded = false) {
    msm_plan p{};
    p.c = 8; p.nwin = 16; p.nb = 128; p.ngroups = folded ? 8 : 16; p.fixed = 0; p.table_n = n;
    return p;
}
static void set_inf_image(void *out_g1) { g1j z = g1_to_kilic(g1_inf()); memcpy(out_g1, &z, sizeof z); }   // Kilic Zero(): (0, R, 0)

// ---- cached point sets for bls.LinCombG1 (bls/bls_kilic.go:132-150): callers such as CommitToEvalPoly (kzg_single_proofs.go:12-14,
// the IFFT of the setup) and eth/helpers.go:99,159,199 (the Lagrange setup) multiply the SAME points by fresh scalars every call.
// The handle keeps them in HBM as affine device-internal images together with 2^64 P_i, which folds the 16 windows of each GLV
// half onto 8 bucket groups: 56 instead of 120 doublings on the critical path of a lone MSM.
static int kzg_settings_build(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_kzg **out);
static int ensure_fixed_table(kzg_hip_kzg *ks, hipStream_t);
static int commit_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride = 0);
static double table_budget_gb(const char *env, double cap_gb, double headroom_gb);
struct kzg_hip_points {
    kzg_hip_fft *fs = nullptr;
    uint64_t n = 0;
    g1a *d_tab = nullptr;          // [P_0 .. P_{n-1} | 2^64 P_0 .. 2^64 P_{n-1}], affine, (0, 0) = inf: the bucket pipeline's rows
    kzg_hip_kzg *ks = nullptr;     // the same points as a settings object: its fixed-base table (built lazily within the set's budget) turns a
                                   // linear combination on the cached set into the table walk of CommitToPoly; null below 64 points
    std::unique_ptr<coalescer> co; // concurrent one-MSM calls (bls.LinCombG1 from many goroutines) merge into batched launches
};
__global__ __launch_bounds__(128, 2) void k_points_shift64(const g1a *pts, uint64_t n, g1j *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    g1a p = pts[t];
    if (is_inf(p)) { out[t] = g1_inf(); return; }
    g1jq q = g1jq_unpack(to_jac(p));
#pragma nounroll
    for (int i = 0; i < 64; i++) q = g1jq_dbl(q);
    out[t] = g1jq_pack(q);
}
void kzg_hip_points_free(kzg_hip_points *pts) {
    if (!pts) return;
    hipSetDevice(pts->fs->device);
    hipDeviceSynchronize();
    pts->co.reset();
    if (pts->ks) kzg_hip_kzg_settings_free(pts->ks);
    hipFree(pts->d_tab);
    (void)hipGetLastError();
    delete pts;
}
int kzg_hip_points_new(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_points **out) {
    if (!fs || !out || (n && !points_g1)) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    KZG_TRY
    std::unique_ptr<kzg_hip_points, void (*)(kzg_hip_points *)> own(new kzg_hip_points, kzg_hip_points_free);
    own->fs = fs; own->n = n;
    if (n) {
        dev_guard g(fs);
        hipStream_t s = fs->stream;
        dtmp<g1j> d_raw(s), d_hi(s);
        CHK(d_raw.alloc(n)); CHK(d_hi.alloc(n));
        HIPCHK(hipMalloc((void **)&own->d_tab, 2 * n * sizeof(g1a)));
        HIPCHK(hipMemcpyAsync(d_raw.p, points_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
        launch_g1_from_kilic(s, d_raw.p, n);
        launch_g1_to_affine(s, d_raw.p, own->d_tab, n);
        hipLaunchKernelGGL(k_points_shift64, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s, own->d_tab, n, d_hi.p);
        launch_g1_to_affine(s, d_hi.p, own->d_tab + n, n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    if (n >= 64) {   // (takes the handle mutex itself)
        CHK(kzg_settings_build(fs, points_g1, n, &own->ks));
        // budget of the set's fixed-base table: KZG_HIP_POINTS_FB_BUDGET_GB, else min(32 GB, free HBM - 24 GB) at creation (4096 points: 13-bit windows,
        // 20 of them, 32 GB); 0 keeps the set on the bucket pipeline.  kzg_hip_points_set_table_budget_gb changes it per set.
        own->ks->budget_gb = table_budget_gb("KZG_HIP_POINTS_FB_BUDGET_GB", 32.0, 24.0);
    }
    *out = own.release();
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_points_set_table_budget_gb(kzg_hip_points *pts, double gb) {
    if (!pts) return KZG_HIP_ERR_BAD_ARG;
    if (!pts->ks) return KZG_HIP_OK;
    return kzg_hip_kzg_set_table_budget_gb(pts->ks, gb);
}
uint64_t kzg_hip_points_count(const kzg_hip_points *pts) { return pts ? pts->n : 0; }
// batch MSMs against points[:n]: scalars in rows of n; out = batch normalised Kilic images (device)
static int lincomb_points_rows(kzg_hip_points *pts, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride = 0) {
    if (pts->ks) {   // the cached set's fixed-base table, when its budget allows one: n x windows mixed additions per combination, no sort, no buckets
        { dev_guard g(pts->fs); CHK(ensure_fixed_table(pts->ks, s)); }
        std::shared_lock<std::shared_mutex> tl(pts->ks->tab_mu);
        if (pts->ks->d_fixed) return commit_rows(pts->ks, s, d_sc, n, batch, d_out, sc_stride ? sc_stride : n);
    }
    msm_plan p = classic_plan(pts->n, true);
    if (!msm_index_range_ok(p, n)) return KZG_HIP_ERR_TOO_WIDE;      // the packed bucket entries would wrap
    dtmp<uint8_t> d_ws(s);
    CHK(d_ws.alloc(msm_workspace_bytes(p, n, batch)));
    launch_msm(s, p, pts->d_tab, d_sc, sc_stride ? sc_stride : n, n, batch, d_ws.p, d_out, true);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_lincomb_points_batch_dev(kzg_hip_points *pts, const void *d_scalars_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream) {
    if (!pts || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > pts->n) return KZG_HIP_ERR_LEN_MISMATCH;          // bls.LinCombG1 length mismatch panic, bls_kilic.go:133-135
    if (!batch) return KZG_HIP_OK;
    if (n == 0 || !d_scalars_fr) return KZG_HIP_ERR_BAD_ARG;
    hipSetDevice(pts->fs->device);
    return lincomb_points_rows(pts, (hipStream_t)stream, (const fr *)d_scalars_fr, n, batch, (g1j *)d_out_g1);
}
int kzg_hip_lincomb_points_batch(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!pts || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > pts->n) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    if (n == 0) { for (uint64_t b = 0; b < batch; b++) set_inf_image((uint8_t *)out_g1 + b * sizeof(g1j)); return KZG_HIP_OK; }   // bls/bls_test.go:69-78
    if (!scalars_fr) return KZG_HIP_ERR_BAD_ARG;
    stream_lease lease(pts->fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<fr> d_sc(s); dtmp<g1j> d_out(s);
    CHK(d_sc.alloc(n * batch)); CHK(d_out.alloc(batch));
    HIPCHK(hipMemcpyAsync(d_sc.p, scalars_fr, n * batch * sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(lincomb_points_rows(pts, s, d_sc.p, n, batch, d_out.p));
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
static int lincomb_points_coalesced(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1);   // below, with the other coalesced entries
int kzg_hip_lincomb_points(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (pts && scalars_fr && out_g1 && n && n <= pts->n) return lincomb_points_coalesced(pts, scalars_fr, n, out_g1);
    return kzg_hip_lincomb_points_batch(pts, scalars_fr, n, 1, out_g1);
}

int kzg_hip_lincomb_g1(kzg_hip_fft *fs, const void *points_g1, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (!fs || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n == 0) { set_inf_image(out_g1); return KZG_HIP_OK; }   // bls/bls_test.go:69-78
    if (!points_g1 || !scalars_fr) return KZG_HIP_ERR_BAD_ARG;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    msm_plan p = classic_plan(n);                               // one-shot points: no 2^64 rows (computing them costs the 64 doublings they save)
    if (!msm_index_range_ok(p, n)) return KZG_HIP_ERR_TOO_WIDE;
    dtmp<g1j> d_pts(s), d_out(s); dtmp<g1a> d_tab(s); dtmp<fr> d_sc(s); dtmp<uint8_t> d_ws(s);
    CHK(d_pts.alloc(n)); CHK(d_out.alloc(1)); CHK(d_tab.alloc(n)); CHK(d_sc.alloc(n)); CHK(d_ws.alloc(msm_workspace_bytes(p, n, 1)));
    HIPCHK(hipMemcpyAsync(d_pts.p, points_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_sc.p, scalars_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_pts.p, n);
    launch_g1_to_affine(s, d_pts.p, d_tab.p, n);
    launch_msm(s, p, d_tab.p, d_sc.p, n, n, 1, d_ws.p, d_out.p, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}

int kzg_hip_fr_from_le32(kzg_hip_fft *fs, const void *in_le32, uint64_t n, void *out_fr, int *all_ok) {
    if (!fs || (n && (!in_le32 || !out_fr))) return KZG_HIP_ERR_BAD_ARG;
    if (all_ok) *all_ok = 1;
    if (!n) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<uint8_t> d_in(s); dtmp<fr> d_out(s); dtmp<uint32_t> d_bad(s);
    CHK(d_in.alloc(32 * n)); CHK(d_out.alloc(n)); CHK(d_bad.alloc(1));
    HIPCHK(hipMemsetAsync(d_bad.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(d_in.p, in_le32, 32 * n, hipMemcpyHostToDevice, s));
    launch_fr_from_le32(s, d_in.p, d_out.p, n, 1, d_bad.p);
    HIPCHK(hipGetLastError());
    uint32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_fr, d_out.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (all_ok) *all_ok = bad ? 0 : 1;
    return KZG_HIP_OK;
}
int kzg_hip_fr_to_le32(kzg_hip_fft *fs, const void *in_fr, uint64_t n, void *out_le32) {
    if (!fs || (n && (!in_fr || !out_le32))) return KZG_HIP_ERR_BAD_ARG;
    if (!n) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<uint8_t> d_out(s); dtmp<fr> d_in(s);
    CHK(d_in.alloc(n)); CHK(d_out.alloc(32 * n));
    HIPCHK(hipMemcpyAsync(d_in.p, in_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    launch_fr_to_le32(s, d_in.p, d_out.p, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_le32, d_out.p, 32 * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_g1_to_compressed(kzg_hip_fft *fs, const void *points_g1, uint64_t n, void *out48) {
    if (!fs || (n && (!points_g1 || !out48))) return KZG_HIP_ERR_BAD_ARG;
    if (!n) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<g1j> d_pts(s); dtmp<uint8_t> d_out(s);
    CHK(d_pts.alloc(n)); CHK(d_out.alloc(48 * n));
    HIPCHK(hipMemcpyAsync(d_pts.p, points_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_pts.p, n);
    launch_g1_compress(s, d_pts.p, d_out.p, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out48, d_out.p, 48 * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_g1_from_compressed(kzg_hip_fft *fs, const void *in48, uint64_t n, void *out_g1) {
    if (!fs || (n && (!in48 || !out_g1))) return KZG_HIP_ERR_BAD_ARG;
    if (!n) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<g1j> d_pts(s); dtmp<uint8_t> d_in(s); dtmp<uint32_t> d_flag(s);
    CHK(d_pts.alloc(n)); CHK(d_in.alloc(48 * n)); CHK(d_flag.alloc(1));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(d_in.p, in48, 48 * n, hipMemcpyHostToDevice, s));
    launch_g1_decompress(s, d_in.p, d_pts.p, n, d_flag.p);
    HIPCHK(hipGetLastError());
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_g1, d_pts.p, n * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return flag ? KZG_HIP_ERR_BAD_POINT : KZG_HIP_OK;
}
int kzg_hip_g1_mul_vec(kzg_hip_fft *fs, const void *points_g1, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (!fs || (n && (!points_g1 || !scalars_fr || !out_g1))) return KZG_HIP_ERR_BAD_ARG;
    if (!n) return KZG_HIP_OK;
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<g1j> d_pts(s), d_out(s); dtmp<fr> d_sc(s);
    CHK(d_pts.alloc(n)); CHK(d_out.alloc(n)); CHK(d_sc.alloc(n));
    HIPCHK(hipMemcpyAsync(d_pts.p, points_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_sc.p, scalars_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_pts.p, n);
    launch_g1_mul_vec(s, d_pts.p, n, d_sc.p, 1, n, d_out.p);
    launch_g1_normalize(s, d_out.p, d_pts.p, n, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_pts.p, n * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_generate_testing_setup_g1(kzg_hip_fft *fs, const void *secret_fr, uint64_t n, void *out_g1) {
    if (!fs || !secret_fr || (n && !out_g1)) return KZG_HIP_ERR_BAD_ARG;
    if (!n) return KZG_HIP_OK;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    dtmp<g1j> d_a(s), d_b(s); dtmp<fr> d_pw(s), d_s(s);
    CHK(d_a.alloc(n)); CHK(d_b.alloc(n)); CHK(d_pw.alloc(n)); CHK(d_s.alloc(1));
    HIPCHK(hipMemcpyAsync(d_s.p, secret_fr, sizeof(fr), hipMemcpyHostToDevice, s));
    launch_fr_powers(s, d_s.p, n, d_pw.p);
    launch_g1_fixed_base_powers(s, d_pw.p, n, d_a.p);
    launch_g1_normalize(s, d_a.p, d_b.p, n, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_b.p, n * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// KZGSettings
// ---------------------------------------------------------------------------------------------------------
// uploads n Kilic images, converts to the device-internal domain, normalises and keeps Jacobian + affine copies resident.
// Every error path frees what was built (the handle is owned by a unique_ptr until the last step).
static int kzg_settings_build(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_kzg **out) {
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    std::unique_ptr<kzg_hip_kzg, void (*)(kzg_hip_kzg *)> own(new kzg_hip_kzg, kzg_hip_kzg_settings_free);
    kzg_hip_kzg *ks = own.get();
    ks->fs = fs; ks->n_setup = n;
    dtmp<g1j> d_raw(s);
    CHK(d_raw.alloc(n));
    HIPCHK(hipMalloc((void **)&ks->d_secret, n * sizeof(g1j)));
    HIPCHK(hipMalloc((void **)&ks->d_secret_a, n * sizeof(g1a)));
    HIPCHK(hipMemcpyAsync(d_raw.p, points_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_raw.p, n);
    launch_g1_normalize(s, d_raw.p, ks->d_secret, n);
    launch_g1_to_affine(s, ks->d_secret, ks->d_secret_a, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    *out = own.release();
    return KZG_HIP_OK;
}
int kzg_hip_kzg_settings_new(kzg_hip_fft *fs, const void *secret_g1, uint64_t n_setup, kzg_hip_kzg **out) {
    if (!fs || !out || !secret_g1) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (n_setup < fs->W) return KZG_HIP_ERR_LEN_MISMATCH;   // kzg.go:25-27
    KZG_TRY
    return kzg_settings_build(fs, secret_g1, n_setup, out);
    KZG_CATCH
}
void kzg_hip_kzg_settings_free(kzg_hip_kzg *ks) {
    if (!ks) return;
    hipSetDevice(ks->fs->device);
    hipDeviceSynchronize();   // _dev callers may still have work in flight that reads the tables: drain the device first
    hipFree(ks->d_secret); hipFree(ks->d_secret_a); hipFree(ks->d_fixed);
    if (ks->copy_stream) hipStreamDestroy(ks->copy_stream);
    for (int i = 0; i < 2; i++) if (ks->copy_done[i]) hipEventDestroy(ks->copy_done[i]);
    (void)hipGetLastError();
    delete ks;
}

// number of signed c-bit windows of a canonical scalar (< r < 2^255): ceil(255 / c), plus one only if the top window's
// digit (top bits of r - 1, plus the incoming carry) can exceed 2^(c-1) and carry out (c = 15 carries: 18 windows, c = 16 does not: 16)
// table budget in GB: the environment override, else min(cap, free HBM - headroom)
static double table_budget_gb(const char *env, double cap_gb, double headroom_gb) {
    if (const char *e = getenv(env)) return atof(e);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0.0;
    double g = (double)free_b / 1e9 - headroom_gb;
    return g > cap_gb ? cap_gb : (g > 0.0 ? g : 0.0);
}
static uint32_t fb_windows(uint32_t c) {
    uint32_t nw = (255 + c - 1) / c, sh = c * (nw - 1);
    uint64_t top = (0x73eda753299d7d48ull >> (sh - 192)) + 1;
    return top > (1ull << (c - 1)) ? nw + 1 : nw;
}

// Lazily builds the fixed-base table T[(w n + i) D + d - 1] = d 2^(c w) SecretG1[i] (k_msm.hip).  The window size is the
// largest whose table fits the HBM budget.  Budget, in this order: kzg_hip_kzg_set_table_budget_gb (per handle, the opt-in for
// the 206 GB c = 16 table), the KZG_HIP_FB_BUDGET_GB environment variable, else the DEFAULT of 64 GB (n = 4096: c = 14,
// 19 windows, 61 GB) clipped to free HBM - 24 GB so that several settings objects (monomial + eth Lagrange + FK20) co-reside.
// Measured, n = 4096, 512 blobs per launch: c = 11 (10 GB) ~39k, c = 13 (32 GB) ~55k, c = 14 (61 GB) ~77k, c = 16 (206 GB,
// 16 windows) ~88k commitments/s (bench.py table_sweep).  If the allocation fails (another process on the GPU, fragmentation)
// the next smaller window is tried, and finally the bucket path, which needs no table: a commitment never fails for lack of HBM.
// The build runs on the HANDLE's stream and only that stream is waited for (by the host thread that found no table): a caller's stream
// passed to a _dev entry point is never synchronised here -- its work already enqueued keeps running under the build.  Callers hold fs->mu.
static int ensure_fixed_table(kzg_hip_kzg *ks, hipStream_t) {
    if (ks->d_fixed || ks->fixed_plan.c == 0xffffffffu) return KZG_HIP_OK;
    hipStream_t s = ks->fs->stream;
    double budget_gb = ks->budget_gb >= 0.0 ? ks->budget_gb : table_budget_gb("KZG_HIP_FB_BUDGET_GB", 64.0, 24.0);
    if (ks->n_setup < 64) { ks->fixed_plan.c = 0xffffffffu; return KZG_HIP_OK; }   // classic path only
    for (uint32_t c = 16; c >= 5; c--) {
        if (c == 15) continue;                               // measured slower than c = 14 (18 windows, 116 GB)
        double bytes = (double)fb_windows(c) * (double)ks->n_setup * (double)(1u << (c - 1)) * sizeof(g1a);
        if (bytes > budget_gb * 1e9) continue;
        msm_plan p{};
        p.c = c; p.nwin = fb_windows(c); p.nb = 1u << (c - 1); p.ngroups = 1; p.fixed = 1; p.table_n = ks->n_setup;
        size_t entries = (size_t)p.nwin * ks->n_setup * p.nb;
        g1a *tab = nullptr;
        if (hipMalloc((void **)&tab, entries * sizeof(g1a)) != hipSuccess) { (void)hipGetLastError(); continue; }   // retry smaller
        hipError_t e = launch_fb_build(s, ks->d_secret_a, ks->n_setup, p.c, p.nwin, tab);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { (void)hipGetLastError(); hipFree(tab); continue; }   // temporaries of the build did not fit
        ks->d_fixed = tab; ks->fixed_plan = p;
        return KZG_HIP_OK;
    }
    ks->fixed_plan.c = 0xffffffffu;                          // no table fits: bucket path
    return KZG_HIP_OK;
}

// MSM of `batch` resident scalar rows against SecretG1[:n]; out = batch normalised points (device).  The partial-sum / bucket
// workspace is allocated per call, stream-ordered on the launch stream (hipMallocAsync pool: no device synchronisation after the
// first use), so concurrent callers on different streams never share scratch memory.
static int commit_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride) {
    CHK(ensure_fixed_table(ks, s));
    bool fixed = ks->d_fixed != nullptr;
    if (!sc_stride) sc_stride = n;
    msm_plan p = fixed ? ks->fixed_plan : classic_plan(ks->n_setup);
    if (!fixed && !msm_index_range_ok(p, n)) return KZG_HIP_ERR_TOO_WIDE;
    size_t ws_main = fixed ? fb_partials_bytes(n, batch) : msm_workspace_bytes(p, n, batch);
    dtmp<uint8_t> d_ws(s);
    CHK(d_ws.alloc(ws_main));
    if (fixed) launch_fb_msm(s, ks->d_fixed, p.table_n, p.c, p.nwin, d_sc, sc_stride, n, batch, d_ws.p, d_out, true);   // sums, normalises, converts
    else launch_msm(s, p, ks->d_secret_a, d_sc, sc_stride, n, batch, d_ws.p, d_out, true);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_kzg_set_table_budget_gb(kzg_hip_kzg *ks, double gb) {
    if (!ks || !(gb >= 0.0)) return KZG_HIP_ERR_BAD_ARG;
    std::unique_lock<std::shared_mutex> tl(ks->tab_mu);        // waits for coalesced batches that are walking the current table
    dev_guard g(ks->fs);
    if (ks->d_fixed) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(ks->d_fixed)); ks->d_fixed = nullptr; }
    ks->fixed_plan = msm_plan{};
    ks->budget_gb = gb;
    return KZG_HIP_OK;
}

int kzg_hip_commit_to_poly_batch_dev(kzg_hip_kzg *ks, const void *d_coeffs_fr, uint64_t n, uint64_t batch, void *d_out_g1, void *stream) {
    if (!ks || !d_out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;   // slice bounds of SecretG1[:len(coeffs)], kzg_single_proofs.go:18
    if (!batch) return KZG_HIP_OK;
    if (n == 0 || !d_coeffs_fr) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(ks->fs);
    return commit_rows(ks, (hipStream_t)stream, (const fr *)d_coeffs_fr, n, batch, (g1j *)d_out_g1);
}
int kzg_hip_commit_to_poly_batch(kzg_hip_kzg *ks, const void *coeffs_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!ks || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    if (n == 0) { for (uint64_t b = 0; b < batch; b++) set_inf_image((uint8_t *)out_g1 + b * sizeof(g1j)); return KZG_HIP_OK; }
    if (!coeffs_fr) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(ks->fs);
    hipStream_t s = ks->fs->stream;
    dtmp<fr> d_sc(s); dtmp<g1j> d_out(s);
    CHK(d_sc.alloc(n * batch)); CHK(d_out.alloc(batch));
    // Large batches are uploaded in chunks on a second stream: the copy of chunk i + 1 (from pageable host memory it occupies the
    // calling thread) runs while the GPU walks chunk i.  Chunks keep >= 256 blobs so that a walk still fills one round of waves.
    uint64_t chunk = batch >= 1024 ? 512 : (batch >= 512 ? 256 : batch);
    if (chunk < batch && n * sizeof(fr) >= (64u << 10)) {
        if (!ks->copy_stream) {
            HIPCHK(hipStreamCreateWithFlags(&ks->copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ks->copy_done[i], hipEventDisableTiming));
        }
        CHK(ensure_fixed_table(ks, s));
        HIPCHK(hipStreamSynchronize(s));                     // d_sc was allocated in order on s: make it visible to the copy stream
        int slot = 0;
        for (uint64_t b0 = 0; b0 < batch; b0 += chunk, slot ^= 1) {
            uint64_t cnt = batch - b0 < chunk ? batch - b0 : chunk;
            HIPCHK(hipMemcpyAsync(d_sc.p + b0 * n, (const fr *)coeffs_fr + b0 * n, n * cnt * sizeof(fr), hipMemcpyHostToDevice, ks->copy_stream));
            HIPCHK(hipEventRecord(ks->copy_done[slot], ks->copy_stream));
            HIPCHK(hipStreamWaitEvent(s, ks->copy_done[slot], 0));
            CHK(commit_rows(ks, s, d_sc.p + b0 * n, n, cnt, d_out.p + b0));
        }
    } else {
        HIPCHK(hipMemcpyAsync(d_sc.p, coeffs_fr, n * batch * sizeof(fr), hipMemcpyHostToDevice, s));
        CHK(commit_rows(ks, s, d_sc.p, n, batch, d_out.p));
    }
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
// ---- one-polynomial calls: concurrent callers on a handle are merged into batched launches (coalesce.hpp) ----
static bool coalescing_enabled() {
    static const bool on = [] { const char *e = getenv("KZG_HIP_COALESCE"); return !(e && e[0] == '0'); }();
    return on;
}
// rows per staging buffer: as many as fit 32 MiB of pinned memory per direction, within [4, 256]
static uint64_t coalesce_rows(size_t in_row, size_t out_row) {
    size_t row = in_row > out_row ? in_row : out_row;
    uint64_t r = (64u << 20) / (row ? row : 1);          // 64 MiB of pinned rows per staging buffer: 113 rows of 4096 proofs (a 56-row cap split 64 callers 56 + 8)
    return r < 4 ? 4 : (r > 256 ? 256 : r);
}
static coalescer *get_coalescer(kzg_hip_fft *fs, std::unique_ptr<coalescer> &slot, size_t in_row, size_t out_row) {
    std::lock_guard<std::mutex> lk(fs->mu);
    if (!slot) slot.reset(new coalescer(fs->device, in_row, out_row, coalesce_rows(in_row, out_row)));
    return slot.get();
}
// uploads the batch's rows (pinned, row stride in_row_bytes) as dense n_max-wide rows and zero-fills the tails
static int coalesce_upload_rows(coalesce_buf &b, uint64_t batch, size_t in_row_bytes, uint64_t n_max, fr *d_rows, uint64_t *d_meta) {
    hipStream_t s = b.stream;
    HIPCHK(hipMemcpyAsync(d_meta, b.h_meta, batch * sizeof(coalesce_row), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpy2DAsync(d_rows, n_max * sizeof(fr), b.h_in, in_row_bytes, n_max * sizeof(fr), batch, hipMemcpyHostToDevice, s));
    launch_fr_zero_tails(s, d_rows, n_max, batch, d_meta, 2);
    return KZG_HIP_OK;
}
int kzg_hip_commit_to_poly(kzg_hip_kzg *ks, const void *coeffs_fr, uint64_t n, void *out_g1) {
    if (!coalescing_enabled() || !ks || !out_g1 || !coeffs_fr || n == 0 || n > ks->n_setup)
        return kzg_hip_commit_to_poly_batch(ks, coeffs_fr, n, 1, out_g1);         // argument errors and n == 0 take the plain path
    KZG_TRY
    coalescer *co = get_coalescer(ks->fs, ks->co_commit, ks->n_setup * sizeof(fr), sizeof(g1j));
    auto exec = [ks, co](coalesce_buf &b, uint64_t batch) -> int {
        hipSetDevice(ks->fs->device);
        hipStream_t s = b.stream;
        drain_on_exit drain(s);
        std::shared_lock<std::shared_mutex> tl(ks->tab_mu);                         // the table stays until this batch has drained
        { dev_guard g(ks->fs); CHK(ensure_fixed_table(ks, s)); }                    // the lazy table build is the only shared mutation
        uint64_t n_max = 0;
        for (uint64_t i = 0; i < batch; i++) n_max = b.h_meta[i].n > n_max ? b.h_meta[i].n : n_max;
        dtmp<fr> d_rows(s); dtmp<uint64_t> d_meta(s);       // only ragged batches are compacted on the device (allocated below)
        // the 144-byte results are written by the last kernel straight into the pinned output rows (no copy kernel queued behind
        // the other batches' walks: it was measured at 80 us per batch under load)
        void *dp_out = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dp_out, b.h_out, 0));
        g1j *d_out = (g1j *)dp_out;
        static const bool trace = getenv("KZG_HIP_COALESCE_TRACE") != nullptr;      // phase times on stderr (adds two synchronisations)
        const auto t0 = std::chrono::steady_clock::now();
        // Uniform rows (the normal case: every caller commits a full blob) are read IN PLACE from the pinned staging buffer: each
        // scalar is loaded exactly once by the table walk, so the 128 KiB per blob stream over PCIe under the walk's own latency
        // hiding instead of costing a separate 0.5 ms copy.  Ragged batches are compacted and zero-filled on the device.
        bool uniform = ks->d_fixed != nullptr;
        for (uint64_t i = 0; i < batch && uniform; i++) uniform = b.h_meta[i].n == n_max;
        const fr *d_src = d_rows.p; uint64_t stride = n_max;
        if (uniform) {
            void *dp = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp, b.h_in, 0));
            d_src = (const fr *)dp; stride = co->in_row_bytes() / sizeof(fr);
        } else {
            CHK(d_rows.alloc(batch * n_max)); CHK(d_meta.alloc(2 * batch));
            d_src = d_rows.p;
            CHK(coalesce_upload_rows(b, batch, co->in_row_bytes(), n_max, d_rows.p, d_meta.p));
        }
        if (trace) hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        CHK(commit_rows(ks, s, d_src, n_max, batch, d_out, stride));
        if (trace) hipStreamSynchronize(s);
        const auto t2 = std::chrono::steady_clock::now();
        HIPCHK(hipStreamSynchronize(s));
        if (trace) {
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
            fprintf(stderr, "[commit batch %llu] upload %.0f us, kernels %.0f us, final synchronisation %.0f us\n", (unsigned long long)batch, us(t0, t1), us(t1, t2),
                    us(t2, std::chrono::steady_clock::now()));
        }
        return KZG_HIP_OK;
    };
    return co->submit(coeffs_fr, n * sizeof(fr), n, 0, out_g1, sizeof(g1j), exec, KZG_HIP_ERR_HIP);
    KZG_CATCH
}

// bls.LinCombG1 on a cached point set, ONE linear combination per call (bls/bls_kilic.go:132-150; what eth.PolynomialToKZGCommitment and
// CommitToEvalPoly call from many goroutines): concurrent calls on a handle run as one batched bucket MSM.  Uniform rows are read in
// place from the pinned staging buffer; ragged ones are compacted and zero-filled on the device (a zero scalar adds nothing).
static int lincomb_points_coalesced(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (!coalescing_enabled()) return kzg_hip_lincomb_points_batch(pts, scalars_fr, n, 1, out_g1);
    KZG_TRY
    coalescer *co = get_coalescer(pts->fs, pts->co, pts->n * sizeof(fr), sizeof(g1j));
    auto exec = [pts, co](coalesce_buf &b, uint64_t batch) -> int {
        hipSetDevice(pts->fs->device);
        hipStream_t s = b.stream;
        drain_on_exit drain(s);
        uint64_t n_max = 0;
        for (uint64_t i = 0; i < batch; i++) n_max = b.h_meta[i].n > n_max ? b.h_meta[i].n : n_max;
        bool uniform = true;
        for (uint64_t i = 0; i < batch && uniform; i++) uniform = b.h_meta[i].n == n_max;
        dtmp<fr> d_rows(s); dtmp<uint64_t> d_meta(s);
        void *dp_out = nullptr;                                                    // results go straight into the pinned output rows
        HIPCHK(hipHostGetDevicePointer(&dp_out, b.h_out, 0));
        const fr *d_src; uint64_t stride;
        if (uniform) {
            void *dp = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp, b.h_in, 0));
            d_src = (const fr *)dp; stride = co->in_row_bytes() / sizeof(fr);
        } else {
            CHK(d_rows.alloc(batch * n_max)); CHK(d_meta.alloc(2 * batch));
            CHK(coalesce_upload_rows(b, batch, co->in_row_bytes(), n_max, d_rows.p, d_meta.p));
            d_src = d_rows.p; stride = n_max;
        }
        CHK(lincomb_points_rows(pts, s, d_src, n_max, batch, (g1j *)dp_out, stride));
        HIPCHK(hipStreamSynchronize(s));
        return KZG_HIP_OK;
    };
    return co->submit(scalars_fr, n * sizeof(fr), n, 0, out_g1, sizeof(g1j), exec, KZG_HIP_ERR_HIP);
    KZG_CATCH
}

// ComputeProofSingle over `batch` resident polynomials: x[b] -> bls.AsFr (kzg_single_proofs.go:39-40), quotient by (X - x[b])
// (polyLongDiv, poly.go:14-40), commitment of the n - 1 quotient coefficients (:53)
static int proof_single_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_poly, uint64_t n, uint64_t batch, const uint64_t *d_x_u64, uint64_t x_stride, g1j *d_out) {
    dtmp<fr> d_q(s), d_x(s);
    CHK(d_q.alloc(batch * (n - 1))); CHK(d_x.alloc(batch));
    launch_fr_from_u64(s, d_x_u64, x_stride, d_x.p, batch);
    launch_quotient_linear(s, d_poly, n, n, batch, d_x.p, d_q.p, n - 1);
    return commit_rows(ks, s, d_q.p, n - 1, batch, d_out);
}
int kzg_hip_compute_proof_single_batch_dev(kzg_hip_kzg *ks, const void *d_poly_fr, uint64_t n, uint64_t batch, const void *d_x_u64, void *d_out_g1, void *stream) {
    if (!ks || !d_out_g1 || n < 2) return KZG_HIP_ERR_BAD_ARG;
    if (n - 1 > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;   // SecretG1[:len(quotient)], kzg_single_proofs.go:53
    if (!batch) return KZG_HIP_OK;
    if (!d_poly_fr || !d_x_u64) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(ks->fs);
    return proof_single_rows(ks, (hipStream_t)stream, (const fr *)d_poly_fr, n, batch, (const uint64_t *)d_x_u64, 1, (g1j *)d_out_g1);
}
int kzg_hip_compute_proof_single_batch(kzg_hip_kzg *ks, const void *poly_fr, uint64_t n, uint64_t batch, const uint64_t *xs, void *out_g1) {
    if (!ks || !out_g1 || n < 2) return KZG_HIP_ERR_BAD_ARG;
    if (n - 1 > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    if (!poly_fr || !xs) return KZG_HIP_ERR_BAD_ARG;
    dev_guard g(ks->fs);
    hipStream_t s = ks->fs->stream;
    dtmp<fr> d_poly(s); dtmp<uint64_t> d_x(s); dtmp<g1j> d_out(s);
    CHK(d_poly.alloc(n * batch)); CHK(d_x.alloc(batch)); CHK(d_out.alloc(batch));
    HIPCHK(hipMemcpyAsync(d_poly.p, poly_fr, n * batch * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_x.p, xs, batch * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    CHK(proof_single_rows(ks, s, d_poly.p, n, batch, d_x.p, 1, d_out.p));
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_compute_proof_single(kzg_hip_kzg *ks, const void *poly_fr, uint64_t n, uint64_t x, void *out_g1) {
    if (!ks || !poly_fr || !out_g1 || n < 2) return KZG_HIP_ERR_BAD_ARG;
    if (!coalescing_enabled() || n - 1 > ks->n_setup) return kzg_hip_compute_proof_single_batch(ks, poly_fr, n, 1, &x, out_g1);
    KZG_TRY
    coalescer *co = get_coalescer(ks->fs, ks->co_proof, (ks->n_setup + 1) * sizeof(fr), sizeof(g1j));
    auto exec = [ks, co](coalesce_buf &b, uint64_t batch) -> int {
        hipSetDevice(ks->fs->device);
        hipStream_t s = b.stream;
        drain_on_exit drain(s);
        std::shared_lock<std::shared_mutex> tl(ks->tab_mu);
        { dev_guard g(ks->fs); CHK(ensure_fixed_table(ks, s)); }
        uint64_t n_max = 0;
        for (uint64_t i = 0; i < batch; i++) n_max = b.h_meta[i].n > n_max ? b.h_meta[i].n : n_max;
        dtmp<fr> d_rows(s); dtmp<uint64_t> d_meta(s);
        CHK(d_rows.alloc(batch * n_max)); CHK(d_meta.alloc(2 * batch));
        void *dp_out = nullptr;                                                    // results go straight into the pinned output rows
        HIPCHK(hipHostGetDevicePointer(&dp_out, b.h_out, 0));
        CHK(coalesce_upload_rows(b, batch, co->in_row_bytes(), n_max, d_rows.p, d_meta.p));
        // a shorter polynomial padded with zero high coefficients has the same quotient (followed by zeros)
        CHK(proof_single_rows(ks, s, d_rows.p, n_max, batch, d_meta.p + 1, 2, (g1j *)dp_out));
        HIPCHK(hipStreamSynchronize(s));
        return KZG_HIP_OK;
    };
    return co->submit(poly_fr, n * sizeof(fr), n, x, out_g1, sizeof(g1j), exec, KZG_HIP_ERR_HIP);
    KZG_CATCH
}

int kzg_hip_compute_proof_multi(kzg_hip_kzg *ks, const void *poly_fr, uint64_t len, uint64_t x, uint64_t n, void *out_g1) {
    (void)x;   // the reference multiplies a zero-initialised xPowN by x n times (kzg_multi_proofs.go:20-24): it stays zero
    if (!ks || !poly_fr || !out_g1 || len < n + 1) return KZG_HIP_ERR_BAD_ARG;
    uint64_t nq = len - n;                                   // polyLongDiv by X^n: quotient = poly[n:] (poly.go:14-40)
    if (nq > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;  // SecretG1[:len(quotient)], kzg_multi_proofs.go:42
    return kzg_hip_commit_to_poly(ks, (const uint8_t *)poly_fr + n * sizeof(fr), nq, out_g1);
}
int kzg_hip_check_proof_multi_interpolation(kzg_hip_kzg *ks, const void *ys_fr, uint64_t n, const void *x_fr, void *out_is1_g1, void *out_xpow_fr) {
    if (!ks || !ys_fr || !x_fr || !out_is1_g1 || n == 0) return KZG_HIP_ERR_BAD_ARG;
    kzg_hip_fft *fs = ks->fs;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;              // "ys is bad, cannot compute FFT" panic, kzg_multi_proofs.go:50-53
    uint64_t np = next_pow2(n);
    if (np > ks->n_setup) return KZG_HIP_ERR_LEN_MISMATCH;
    dev_guard g(fs);
    hipStream_t s = fs->stream;
    dtmp<fr> d_ys(s), d_ip(s), d_x(s); dtmp<g1j> d_out(s);
    CHK(d_ys.alloc(n)); CHK(d_ip.alloc(np)); CHK(d_x.alloc(2)); CHK(d_out.alloc(1));
    HIPCHK(hipMemcpyAsync(d_ys.p, ys_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_x.p, x_fr, sizeof(fr), hipMemcpyHostToDevice, s));
    fr_fft_rows(fs, s, d_ys.p, n, n, d_ip.p, np, 1, 1);
    launch_fr_scale_by_inv_powers(s, d_ip.p, d_x.p, np, d_x.p + 1);
    CHK(commit_rows(ks, s, d_ip.p, np, 1, d_out.p));
    HIPCHK(hipMemcpyAsync(out_is1_g1, d_out.p, sizeof(g1j), hipMemcpyDeviceToHost, s));
    if (out_xpow_fr) HIPCHK(hipMemcpyAsync(out_xpow_fr, d_x.p + 1, sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_toeplitz_part2(kzg_hip_kzg *ks, const void *coeffs_fr, const void *x_ext_fft_g1, uint64_t n, void *out_g1) {
    if (!ks || !coeffs_fr || !x_ext_fft_g1 || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    kzg_hip_fft *fs = ks->fs;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;              // FFT error -> panic, fk20_single.go:63-66
    if (!is_pow2(n) || n == 0) return KZG_HIP_ERR_LEN_MISMATCH;   // padded FFT length != len(xExtFFT): index panic in the reference
    stream_lease lease(fs);     // its own stream: host-buffer calls from many threads run side by side
    hipStream_t s = lease.s;
    dtmp<fr> d_c(s), d_cf(s); dtmp<g1j> d_x(s), d_h(s);
    CHK(d_c.alloc(n)); CHK(d_cf.alloc(n)); CHK(d_x.alloc(n)); CHK(d_h.alloc(n));
    HIPCHK(hipMemcpyAsync(d_c.p, coeffs_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_x.p, x_ext_fft_g1, n * sizeof(g1j), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_x.p, n);
    fr_fft_rows(fs, s, d_c.p, n, n, d_cf.p, n, 1, 0);
    launch_g1_mul_vec(s, d_x.p, n, d_cf.p, 1, n, d_h.p);
    launch_g1_normalize(s, d_h.p, d_x.p, n, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_x.p, n * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_toeplitz_part3(kzg_hip_kzg *ks, const void *h_ext_fft_g1, uint64_t n, void *out_g1) {
    if (!ks) return KZG_HIP_ERR_BAD_ARG;
    if (n > ks->fs->W) return KZG_HIP_ERR_TOO_WIDE;            // FFTG1 error -> panic, fk20_single.go:80-84
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n == 0 || !h_ext_fft_g1 || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<g1j> full(n);
    CHK(kzg_hip_fft_g1(ks->fs, h_ext_fft_g1, n, 1, full.data()));   // fk20_single.go:80-87
    memcpy(out_g1, full.data(), (n / 2) * sizeof(g1j));
    return KZG_HIP_OK;
    KZG_CATCH
}

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
    {   // fixed-base table over the file points, sized by KZG_HIP_FK20_FB_BUDGET_GB (default: min(48 GB, free HBM - 12 GB)):
        // scale 12, l = 1: c = 13, 20 windows, 32 GB;  scale 16, l = 16 (65 536 file points): c = 9, 29 windows, 47 GB.
        // An allocation failure falls back to the next smaller window and finally to the table-free double-and-add path.
        double budget_gb = table_budget_gb("KZG_HIP_FK20_FB_BUDGET_GB", 48.0, 12.0);
        uint64_t npts = l * k2;
        dtmp<g1a> d_fa(s);
        if (npts >= 64) { CHK(d_fa.alloc(npts)); launch_g1_to_affine(s, c->d_files, d_fa.p, npts); }
        for (uint32_t cc = 14; cc >= 4 && npts >= 64; cc--) {
            double bytes = (double)fb_windows(cc) * (double)npts * (double)(1u << (cc - 1)) * sizeof(g1a);
            if (bytes > budget_gb * 1e9) continue;
            g1a *tab = nullptr;
            if (hipMalloc((void **)&tab, (size_t)fb_windows(cc) * npts * (1u << (cc - 1)) * sizeof(g1a)) != hipSuccess) { (void)hipGetLastError(); continue; }
            hipError_t e = launch_fb_build(s, d_fa.p, npts, cc, fb_windows(cc), tab);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipGetLastError(); hipFree(tab); continue; }
            c->d_files_fb = tab; c->fb_c = cc; c->fb_nwin = fb_windows(cc);
            break;
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}

// steps 1-3: Toeplitz coefficients (pre-scaled by 1/2k, which folds the inverse FFT's scale into the scalars),
// FFT_Fr, and hExtFFT[j] = sum_f C_f[j] * X_f[j] for j in [j0, j0 + cnt)
static int fk20_hext(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, uint64_t j0, uint64_t cnt, g1j *d_hext) {
    kzg_hip_fft *fs = c->ks->fs;
    uint64_t l = c->l, k2 = 2 * c->k;
    dtmp<fr> d_tc(s), d_cf(s);
    CHK(d_tc.alloc(batch * l * k2)); CHK(d_cf.alloc(batch * l * k2));
    launch_toeplitz_coeffs(s, d_poly, poly_stride, n, l, batch, d_tc.p, fs->d_inv_pow2 + ilog2(k2));
    fr_fft_rows(fs, s, d_tc.p, k2, k2, d_cf.p, k2, batch * l, 0);
    if (c->d_files_fb) {
        if (l == 1) launch_fb_mul_vec(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_hext);
        else if (batch * cnt >= 65536) launch_fb_mul_vec_files(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_hext);   // enough output positions to fill the GPU: one lane sums all files
        else {   // few positions (one polynomial, a shard): a lane per (file, position), then the sum over the files
            dtmp<g1j> d_tmp(s);
            CHK(d_tmp.alloc(batch * l * cnt));
            launch_fb_mul_vec(s, c->d_files_fb, l * k2, c->fb_c, c->fb_nwin, d_cf.p, k2, j0, cnt, batch, d_tmp.p);
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
static int fk20_finish(fk20_core *c, hipStream_t s, const g1j *d_hext, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
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
    launch_fb_direct_pass1(s, c->d_files_fb, k2, c->fb_c, c->fb_nwin, d_cf.p, fs->d_reversed, fs->W, batch, 4, d_p1.p);
    launch_g1_fft_direct(s, d_p1.p, k2, k2, d_a.p, d_tmp.p, k2, batch, fs->d_reversed, fs->W, nullptr, 4, g1_fft_direct_lanes(k2, batch), 4, c->k);   // only h[:k] is read below
    HIPCHK(hipGetLastError());
    return fk20_finish_from_h(c, s, d_a.p, batch, da, bit_reverse, d_out);
}
// DA form of a single-file settings object with its table resident: the Toeplitz stage absorbs the first two stages of the inverse
// transform (k_fb_mul_vec_dif2), the remaining ones run decimation-in-frequency and leave h bit-reversed, which is the layout the
// forward (decimation-in-time) transform reads: 10 instead of 12 multiplying stages for the inverse transform and no reordering
// passes.  Same group elements as the plain pipeline; outputs are normalised, so the bytes are identical (tests compare both).
static bool fk20_fused_ok(const fk20_core *c, uint64_t batch, int da) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_FK20_FUSE"); return e && e[0] == '0'; }();
    const uint64_t k2 = 2 * c->k;
    return !off && da && c->l == 1 && c->d_files_fb && k2 >= 8 && !g1_fft_direct_mode(k2, batch);
}
static int fk20_run_fused(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int bit_reverse, g1j *d_out) {
    kzg_hip_fft *fs = c->ks->fs;
    const uint64_t k2 = 2 * c->k;
    dtmp<fr> d_tc(s), d_cf(s); dtmp<g1j> d_a(s), d_b(s);
    CHK(d_tc.alloc(batch * k2)); CHK(d_cf.alloc(batch * k2)); CHK(d_a.alloc(batch * k2)); CHK(d_b.alloc(batch * k2));
    launch_toeplitz_coeffs(s, d_poly, poly_stride, n, 1, batch, d_tc.p, fs->d_inv_pow2 + ilog2(k2));   // 1 / 2k folded into the scalars
    fr_fft_rows(fs, s, d_tc.p, k2, k2, d_cf.p, k2, batch, 0);
    launch_fb_mul_vec_dif2(s, c->d_files_fb, k2, c->fb_c, c->fb_nwin, d_cf.p, fs->d_reversed, fs->W, batch, d_a.p);
    for (uint64_t m = k2 / 8; m >= 1; m >>= 1) launch_g1_fft_stage_dif(s, d_a.p, k2, batch, m, fs->d_glv_reversed, fs->d_wnaf_reversed, fs->W);
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
static int fk20_run_dev(fk20_core *c, hipStream_t s, const fr *d_poly, uint64_t poly_stride, uint64_t n, uint64_t batch, int da, int bit_reverse, g1j *d_out) {
    if (const uint64_t padded = fk20_padded_batch(batch); padded != batch) {
        const uint64_t on = da ? 2 * c->k : c->k;
        dtmp<fr> d_p2(s); dtmp<g1j> d_o2(s);
        CHK(d_p2.alloc(padded * n)); CHK(d_o2.alloc(padded * on));
        HIPCHK(hipMemcpy2DAsync(d_p2.p, n * sizeof(fr), d_poly, poly_stride * sizeof(fr), n * sizeof(fr), batch, hipMemcpyDeviceToDevice, s));
        for (uint64_t b = batch; b < padded; b++)
            HIPCHK(hipMemcpyAsync(d_p2.p + b * n, d_poly + (batch - 1) * poly_stride, n * sizeof(fr), hipMemcpyDeviceToDevice, s));
        if (fk20_fused_ok(c, padded, da)) CHK(fk20_run_fused(c, s, d_p2.p, n, n, padded, bit_reverse, d_o2.p));
        else {
            dtmp<g1j> d_hext(s);
            CHK(d_hext.alloc(padded * 2 * c->k));
            CHK(fk20_hext(c, s, d_p2.p, n, n, padded, 0, 2 * c->k, d_hext.p));
            CHK(fk20_finish(c, s, d_hext.p, padded, da, bit_reverse, d_o2.p));
        }
        HIPCHK(hipMemcpyAsync(d_out, d_o2.p, batch * on * sizeof(g1j), hipMemcpyDeviceToDevice, s));
        return KZG_HIP_OK;
    }
    if (fk20_fused_ok(c, batch, da)) return fk20_run_fused(c, s, d_poly, poly_stride, n, batch, bit_reverse, d_out);
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
    coalescer *co = get_coalescer(c->ks->fs, c->co_da, n * sizeof(fr), on * sizeof(g1j));
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

// ---------------------------------------------------------------------------------------------------------
// eth/ byte-level prover path (row f1)
// ---------------------------------------------------------------------------------------------------------
struct kzg_hip_eth {
    kzg_hip_fft *fs = nullptr;
    kzg_hip_kzg *ks = nullptr;     // "SecretG1" = bit-reversed Lagrange setup (kzgSetupLagrange, eth/globals.go:48)
    uint64_t n = 0;
    fr *d_domain = nullptr;        // DomainFr: w^bitrev(i) (eth/globals.go:61-66)
    std::unique_ptr<coalescer> co_blob;   // concurrent one-blob BlobToKZGCommitment calls (eth/eth.go:145-151) merge into batched launches
    std::unique_ptr<coalescer> co_proof;  // concurrent ComputeKZGProof calls (eth/helpers.go:179-203)
};

int kzg_hip_eth_settings_new(kzg_hip_fft *fs, const void *lagrange_g1, uint64_t n, kzg_hip_eth **out) {
    if (!fs || !lagrange_g1 || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (n == 0 || !is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    KZG_TRY
    std::vector<g1j> br(n);
    std::vector<fr> dom(n);
    const g1j *src = (const g1j *)lagrange_g1;
    uint32_t logn = ilog2(n);
    for (uint64_t i = 0; i < n; i++) {   // bitReversalPermutation (eth/helpers.go, used at eth/globals.go:48)
        uint64_t r = 0;
        for (uint32_t b = 0; b < logn; b++) if (i & (1ull << b)) r |= 1ull << (logn - 1 - b);
        br[i] = src[r];
        // natural-order scale-log2(n) domain = every (W / n)-th expanded root; DomainFr[i] = domain[bitrev(i)] (eth/globals.go:61-66)
        dom[i] = fs->h_expanded[r * (fs->W / n)];
    }
    std::unique_ptr<kzg_hip_eth, void (*)(kzg_hip_eth *)> own(new kzg_hip_eth, kzg_hip_eth_settings_free);
    kzg_hip_eth *eth = own.get();
    eth->fs = fs; eth->n = n;
    // KZGSettings requires len(setup) >= MaxWidth (kzg.go:25-27); the eth setup is exactly its own width, so build it directly
    CHK(kzg_settings_build(fs, br.data(), n, &eth->ks));
    {
        dev_guard g(fs);
        HIPCHK(hipMalloc((void **)&eth->d_domain, n * sizeof(fr)));
        HIPCHK(hipMemcpy(eth->d_domain, dom.data(), n * sizeof(fr), hipMemcpyHostToDevice));
    }
    *out = own.release();
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_eth_settings_free(kzg_hip_eth *eth) {
    if (!eth) return;
    hipSetDevice(eth->fs->device);
    hipDeviceSynchronize();
    eth->co_blob.reset(); eth->co_proof.reset();
    kzg_hip_kzg_settings_free(eth->ks);   // drains the device first
    hipFree(eth->d_domain);
    (void)hipGetLastError();
    delete eth;
}
int kzg_hip_eth_blob_to_kzg_commitment_batch(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out48, uint8_t *ok) {
    if (!eth || !blobs_le32 || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    if (batch == 1 && coalescing_enabled()) {
        // eth.BlobToKZGCommitment takes ONE blob per call (eth/eth.go:145-151): concurrent callers share batched launches.  A row is
        // the blob's 32-byte little-endian elements (read in place from the pinned staging buffer by the conversion kernel); a result
        // row is the 48 compressed bytes + the "invalid element" flag of BlobToPolynomial.
        const uint64_t n = eth->n;
        coalescer *co = get_coalescer(eth->fs, eth->co_blob, n * 32, 64);
        auto exec = [eth, n](coalesce_buf &b, uint64_t rows) -> int {
            hipSetDevice(eth->fs->device);
            hipStream_t s = b.stream;
            drain_on_exit drain(s);
            std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
            { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
            dtmp<uint8_t> d_c(s); dtmp<fr> d_poly(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
            CHK(d_c.alloc(rows * 48)); CHK(d_poly.alloc(rows * n)); CHK(d_out.alloc(rows)); CHK(d_bad.alloc(rows));
            HIPCHK(hipMemsetAsync(d_bad.p, 0, rows * 4, s));
            void *dp_in = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp_in, b.h_in, 0));
            launch_fr_from_le32(s, (const uint8_t *)dp_in, d_poly.p, n, rows, d_bad.p);
            CHK(commit_rows(eth->ks, s, d_poly.p, n, rows, d_out.p));
            launch_g1_from_kilic(s, d_out.p, rows);
            launch_g1_compress(s, d_out.p, d_c.p, rows);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy2DAsync(b.h_out, 64, d_c.p, 48, 48, rows, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpy2DAsync(b.h_out + 48, 64, d_bad.p, 4, 4, rows, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            return KZG_HIP_OK;
        };
        uint8_t row[64];
        int st = co->submit(blobs_le32, n * 32, n, 0, row, 64, exec, KZG_HIP_ERR_HIP);
        if (st != KZG_HIP_OK) return st;
        uint32_t bad; memcpy(&bad, row + 48, 4);
        ok[0] = bad ? 0 : 1;
        if (bad) memset(out48, 0, 48); else memcpy(out48, row, 48);
        return KZG_HIP_OK;
    }
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    uint64_t n = eth->n;
    dtmp<uint8_t> d_in(s), d_c(s); dtmp<fr> d_poly(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
    CHK(d_in.alloc(batch * n * 32)); CHK(d_c.alloc(batch * 48)); CHK(d_poly.alloc(batch * n)); CHK(d_out.alloc(batch)); CHK(d_bad.alloc(batch));
    HIPCHK(hipMemsetAsync(d_bad.p, 0, batch * 4, s));
    HIPCHK(hipMemcpyAsync(d_in.p, blobs_le32, batch * n * 32, hipMemcpyHostToDevice, s));
    launch_fr_from_le32(s, d_in.p, d_poly.p, n, batch, d_bad.p);                 // BlobToPolynomial, eth/helpers.go:264-273
    CHK(commit_rows(eth->ks, s, d_poly.p, n, batch, d_out.p));                  // PolynomialToKZGCommitment, eth/helpers.go:98-103
    launch_g1_from_kilic(s, d_out.p, batch);                                    // commit_rows leaves Kilic images; compress wants internal
    launch_g1_compress(s, d_out.p, d_c.p, batch);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> bad(batch);
    HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out48, d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) {
        ok[b] = bad[b] ? 0 : 1;
        if (bad[b]) memset((uint8_t *)out48 + 48 * b, 0, 48);
    }
    return KZG_HIP_OK;
    KZG_CATCH
}
// ComputeKZGProof over resident rows (eth/helpers.go:179-203): quotients in evaluation form, their commitment over the Lagrange setup, 48-byte
// compression.  No host round trip in between: a row whose z lies in the domain gets bad[row] = 1 and a zero quotient.
static int eth_proof_rows(kzg_hip_eth *eth, hipStream_t s, const fr *d_poly, uint64_t poly_stride, const fr *d_z, uint64_t z_stride, uint64_t batch, uint8_t *d_out48,
                          fr *d_y, uint32_t *d_bad) {
    const uint64_t n = eth->n;
    dtmp<fr> d_q(s); dtmp<g1j> d_out(s);
    CHK(d_q.alloc(batch * n)); CHK(d_out.alloc(batch));
    HIPCHK(hipMemsetAsync(d_bad, 0, batch * 4, s));
    launch_eth_quotient(s, d_poly, poly_stride, eth->d_domain, n, batch, d_z, z_stride, eth->fs->d_inv_pow2 + ilog2(n), d_q.p, d_y, d_bad);
    CHK(commit_rows(eth->ks, s, d_q.p, n, batch, d_out.p));                      // bls.LinCombG1(kzgSetupLagrange, quotient), eth/helpers.go:199
    launch_g1_from_kilic(s, d_out.p, batch);
    launch_g1_compress(s, d_out.p, d_out48, batch);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_eth_compute_kzg_proof_batch_dev(kzg_hip_eth *eth, const void *d_polys_fr, uint64_t n, uint64_t batch, const void *d_zs_fr, void *d_out48, void *d_ys_fr,
                                            void *d_bad_u32, void *stream) {
    if (!eth || !d_out48 || !d_bad_u32) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    if (!d_polys_fr || !d_zs_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    hipStream_t s = (hipStream_t)stream;
    std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
    { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
    dtmp<fr> d_y(s);
    fr *yp = (fr *)d_ys_fr;
    if (!yp) { CHK(d_y.alloc(batch)); yp = d_y.p; }
    return eth_proof_rows(eth, s, (const fr *)d_polys_fr, n, (const fr *)d_zs_fr, 1, batch, (uint8_t *)d_out48, yp, (uint32_t *)d_bad_u32);
    KZG_CATCH
}
int kzg_hip_eth_compute_kzg_proof_batch(kzg_hip_eth *eth, const void *polys_fr, uint64_t n, uint64_t batch, const void *zs_fr, void *out48, void *ys_fr, uint8_t *ok) {
    if (!eth || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;                            // "polynomial has invalid length", eth/helpers.go:186-188
    if (!batch) return KZG_HIP_OK;
    if (!polys_fr || !zs_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    dtmp<fr> d_poly(s), d_z(s), d_y(s); dtmp<uint8_t> d_c(s); dtmp<uint32_t> d_bad(s);
    CHK(d_poly.alloc(batch * n)); CHK(d_z.alloc(batch)); CHK(d_y.alloc(batch)); CHK(d_c.alloc(batch * 48)); CHK(d_bad.alloc(batch));
    HIPCHK(hipMemcpyAsync(d_poly.p, polys_fr, batch * n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_z.p, zs_fr, batch * sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(ensure_fixed_table(eth->ks, s));
    CHK(eth_proof_rows(eth, s, d_poly.p, n, d_z.p, 1, batch, d_c.p, d_y.p, d_bad.p));
    std::vector<uint32_t> bad(batch);
    HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out48, d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
    if (ys_fr) HIPCHK(hipMemcpyAsync(ys_fr, d_y.p, batch * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) {
        ok[b] = bad[b] ? 0 : 1;
        if (bad[b]) { memset((uint8_t *)out48 + 48 * b, 0, 48); if (ys_fr) memset((uint8_t *)ys_fr + 32 * b, 0, 32); }
    }
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_eth_compute_kzg_proof(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *z_fr, void *out48, void *y_fr) {
    if (!eth || !poly_fr || !z_fr || !out48) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;                            // "polynomial has invalid length", eth/helpers.go:186-188
    KZG_TRY
    uint8_t row[128];
    if (!coalescing_enabled()) {
        uint8_t ok = 0;
        CHK(kzg_hip_eth_compute_kzg_proof_batch(eth, poly_fr, n, 1, z_fr, row, row + 48, &ok));
        if (!ok) return KZG_HIP_ERR_BAD_ARG;
    } else {
        // eth.ComputeKZGProof takes ONE polynomial per call: concurrent callers share batched launches.  A request's row is its polynomial
        // followed by z (read in place from the pinned staging buffer); a result row is 48 proof bytes | y | the "invalid z" flag.
        coalescer *co = get_coalescer(eth->fs, eth->co_proof, (n + 1) * sizeof(fr), 128);
        auto exec = [eth, n, co](coalesce_buf &b, uint64_t rows) -> int {
            hipSetDevice(eth->fs->device);
            hipStream_t s = b.stream;
            drain_on_exit drain(s);
            std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
            { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
            dtmp<uint8_t> d_c(s); dtmp<fr> d_y(s); dtmp<uint32_t> d_bad(s);
            CHK(d_c.alloc(rows * 48)); CHK(d_y.alloc(rows)); CHK(d_bad.alloc(rows));
            void *dp_in = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp_in, b.h_in, 0));
            const uint64_t stride = co->in_row_bytes() / sizeof(fr);
            CHK(eth_proof_rows(eth, s, (const fr *)dp_in, stride, (const fr *)dp_in + n, stride, rows, d_c.p, d_y.p, d_bad.p));
            HIPCHK(hipMemcpy2DAsync(b.h_out, 128, d_c.p, 48, 48, rows, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpy2DAsync(b.h_out + 48, 128, d_y.p, 32, 32, rows, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpy2DAsync(b.h_out + 80, 128, d_bad.p, 4, 4, rows, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            return KZG_HIP_OK;
        };
        int st = co->submit(poly_fr, n * sizeof(fr), n, 0, row, 128, exec, KZG_HIP_ERR_HIP, z_fr, sizeof(fr));
        if (st != KZG_HIP_OK) return st;
        uint32_t bad; memcpy(&bad, row + 80, 4);
        if (bad) return KZG_HIP_ERR_BAD_ARG;                                     // "invalid z challenge", eth/helpers.go:190-192
    }
    memcpy(out48, row, 48);
    if (y_fr) memcpy(y_fr, row + 48, sizeof(fr));
    return KZG_HIP_OK;
    KZG_CATCH
}

// ---------------------------------------------------------------------------------------------------------
// eth.ComputeAggregateKZGProof / the prover-side pieces of eth.VerifyAggregateKZGProof (eth/eth.go:155-182, eth/helpers.go:137-176,215-260):
// the block-level caller of the commitment path.  Device: blobs -> polynomials -> commitments -> aggregated polynomial -> proof; host: the
// Fiat-Shamir transcript (one SHA-256 chain over every blob of the block), hashed while the device commits.
// ---------------------------------------------------------------------------------------------------------
// hashToBLSField (eth/helpers.go:113-133): SHA-256, digest read as a little-endian integer, reduced mod r
static fr hash_to_bls_field(const uint8_t *input, size_t len) {
    sha256 h;
    h.update(input, len);
    uint8_t d[32];
    h.final(d);
    uint64_t v[4], m[4];
    memcpy(v, d, 32);
    for (int i = 0; i < 4; i++) m[i] = (uint64_t)FrP::mod(2 * i) | (uint64_t)FrP::mod(2 * i + 1) << 32;
    for (int k = 0; k < 3; k++) {                                   // 2^256 < 3 r
        bool ge = true;
        for (int i = 3; i >= 0; i--) { if (v[i] != m[i]) { ge = v[i] > m[i]; break; } }
        if (!ge) break;
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)v[i] - m[i] - (uint64_t)br; v[i] = (uint64_t)t; br = (t >> 64) & 1; }
    }
    fr c;
    memcpy(c.l, v, 32);
    return to_mont<FrP>(c);
}
// ComputeAggregatedPolyAndCommitment (eth/helpers.go:137-162) up to the aggregated polynomial: BlobsToPolynomials (:275-285), the commitments
// (taken from `comm_in`, or PolynomialToKZGCommitment of every blob, :166-169, into `comm`), ComputeChallenges (:215-232), bls.PolyLinComb.
// Leaves polynomials, powers (device + host) and the aggregated polynomial resident; z_out = the evaluation challenge.
static int eth_aggregate(kzg_hip_eth *eth, hipStream_t s, const uint8_t *blobs, const uint8_t *comm_in, uint64_t batch, dtmp<fr> &d_poly, dtmp<fr> &d_agg,
                         dtmp<fr> &d_pow, std::vector<fr> &pw, std::vector<uint8_t> &comm, fr &z_out) {
    const uint64_t n = eth->n;
    CHK(d_agg.alloc(n));
    std::vector<uint32_t> bad(batch);
    drain_on_exit drain(s);                                          // an error return must not leave a copy into `bad` in flight
    dtmp<uint8_t> d_in(s), d_c(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
    if (batch) {
        CHK(d_in.alloc(batch * n * 32)); CHK(d_poly.alloc(batch * n)); CHK(d_bad.alloc(batch)); CHK(d_pow.alloc(batch));
        HIPCHK(hipMemsetAsync(d_bad.p, 0, batch * 4, s));
        HIPCHK(hipMemcpyAsync(d_in.p, blobs, batch * n * 32, hipMemcpyHostToDevice, s));
        launch_fr_from_le32(s, d_in.p, d_poly.p, n, batch, d_bad.p);
        HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
        if (!comm_in) {
            comm.resize(batch * 48);
            CHK(d_c.alloc(batch * 48)); CHK(d_out.alloc(batch));
            CHK(commit_rows(eth->ks, s, d_poly.p, n, batch, d_out.p));
            launch_g1_from_kilic(s, d_out.p, batch);
            launch_g1_compress(s, d_out.p, d_c.p, batch);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(comm.data(), d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
        }
    }
    // hashPolysComms (eth/helpers.go:235-260), while the device works: FrTo32 of a valid element is the blob's own 32 bytes
    sha256 h;
    h.update("FSBLOBVERIFY_V1_", 16);
    h.update_u64_le(n);                                              // FieldElementsPerBlob
    h.update_u64_le(batch);
    if (batch) h.update(blobs, batch * n * 32);
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) if (bad[b]) return KZG_HIP_ERR_BAD_BLOB;   // "could not convert blobs to polynomials"
    if (batch) h.update(comm_in ? comm_in : comm.data(), batch * 48);
    uint8_t tr[33];
    h.final(tr);
    tr[32] = 0;
    const fr r_chal = hash_to_bls_field(tr, 33);                     // linCombChallenge
    tr[32] = 1;
    z_out = hash_to_bls_field(tr, 33);                               // evalChallenge
    pw.resize(batch);                                                // ComputePowers (eth/helpers.go:87-96)
    fr cur = one<FrP>();
    for (uint64_t i = 0; i < batch; i++) { pw[i] = cur; cur = mul(cur, r_chal); }
    if (batch) {
        HIPCHK(hipMemcpyAsync(d_pow.p, pw.data(), batch * sizeof(fr), hipMemcpyHostToDevice, s));
        launch_poly_lincomb(s, d_poly.p, n, d_pow.p, batch, n, d_agg.p);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemsetAsync(d_agg.p, 0, n * sizeof(fr), s));       // PolyLinComb of no vector: zeros (bls/globals.go:157-159)
    }
    return KZG_HIP_OK;
}
int kzg_hip_eth_compute_aggregate_kzg_proof(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out_proof48, void *out_commitments48) {
    if (!eth || !out_proof48 || (batch && !blobs_le32)) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    std::vector<fr> pw; std::vector<uint8_t> comm; fr z;
    uint32_t bad = 0;
    uint8_t proof[48];
    drain_on_exit drain(s);                                          // declared after the host buffers the stream copies from / into
    CHK(ensure_fixed_table(eth->ks, s));
    dtmp<fr> d_poly(s), d_agg(s), d_pow(s), d_z(s), d_y(s); dtmp<uint8_t> d_c(s); dtmp<uint32_t> d_bad(s);
    CHK(eth_aggregate(eth, s, (const uint8_t *)blobs_le32, nullptr, batch, d_poly, d_agg, d_pow, pw, comm, z));
    CHK(d_z.alloc(1)); CHK(d_y.alloc(1)); CHK(d_c.alloc(48)); CHK(d_bad.alloc(1));
    HIPCHK(hipMemcpyAsync(d_z.p, &z, sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(eth_proof_rows(eth, s, d_agg.p, eth->n, d_z.p, 1, 1, d_c.p, d_y.p, d_bad.p));   // ComputeKZGProof(aggregatedPoly, evaluationChallenge), eth/helpers.go:175
    HIPCHK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(proof, d_c.p, 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return KZG_HIP_ERR_BAD_ARG;                             // "invalid z challenge"
    memcpy(out_proof48, proof, 48);
    if (out_commitments48 && batch) memcpy(out_commitments48, comm.data(), batch * 48);
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_eth_compute_aggregated_poly_and_commitment(kzg_hip_eth *eth, const void *blobs_le32, const void *commitments48, uint64_t batch, void *out_poly_fr,
                                                       void *out_commitment_g1, void *out_z_fr, void *out_y_fr) {
    if (!eth || !out_commitment_g1 || !out_z_fr || (batch && (!blobs_le32 || !commitments48))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    const uint64_t n = eth->n;
    std::vector<fr> pw; std::vector<uint8_t> comm; fr z, y;
    g1j agg_c;
    uint32_t flag[2] = {0, 0};
    drain_on_exit drain(s);                                          // declared after the host buffers the stream copies from / into
    dtmp<fr> d_poly(s), d_agg(s), d_pow(s), d_z(s), d_y(s), d_q(s); dtmp<uint32_t> d_flag(s);
    CHK(eth_aggregate(eth, s, (const uint8_t *)blobs_le32, (const uint8_t *)commitments48, batch, d_poly, d_agg, d_pow, pw, comm, z));
    // aggregatedCommitmentG1 = LinCombG1(FromCompressedG1(commitments), powers), eth/helpers.go:149-160
    set_inf_image(&agg_c);
    CHK(d_flag.alloc(2));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 8, s));
    dtmp<g1j> d_pts(s), d_out(s); dtmp<g1a> d_tab(s); dtmp<uint8_t> d_cin(s), d_ws(s);
    if (batch) {
        msm_plan p = classic_plan(batch);
        if (!msm_index_range_ok(p, batch)) return KZG_HIP_ERR_TOO_WIDE;
        CHK(d_pts.alloc(batch)); CHK(d_out.alloc(1)); CHK(d_tab.alloc(batch)); CHK(d_cin.alloc(batch * 48)); CHK(d_ws.alloc(msm_workspace_bytes(p, batch, 1)));
        HIPCHK(hipMemcpyAsync(d_cin.p, commitments48, batch * 48, hipMemcpyHostToDevice, s));
        launch_g1_decompress(s, d_cin.p, d_pts.p, batch, d_flag.p);
        launch_g1_from_kilic(s, d_pts.p, batch);
        launch_g1_to_affine(s, d_pts.p, d_tab.p, batch);
        launch_msm(s, p, d_tab.p, d_pow.p, batch, batch, 1, d_ws.p, d_out.p, true);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&agg_c, d_out.p, sizeof(g1j), hipMemcpyDeviceToHost, s));
    }
    // y = EvaluatePolynomialInEvaluationForm(aggregatedPoly, evaluationChallenge) (eth/eth.go:166): the quotient kernel's first half
    CHK(d_z.alloc(1)); CHK(d_y.alloc(1)); CHK(d_q.alloc(n));
    HIPCHK(hipMemcpyAsync(d_z.p, &z, sizeof(fr), hipMemcpyHostToDevice, s));
    launch_eth_quotient(s, d_agg.p, n, eth->d_domain, n, 1, d_z.p, 1, eth->fs->d_inv_pow2 + ilog2(n), d_q.p, d_y.p, d_flag.p + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(flag, d_flag.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&y, d_y.p, sizeof(fr), hipMemcpyDeviceToHost, s));
    if (out_poly_fr) HIPCHK(hipMemcpyAsync(out_poly_fr, d_agg.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag[0]) return KZG_HIP_ERR_BAD_POINT;                       // FromCompressedG1 failed, eth/helpers.go:153-156
    if (flag[1]) return KZG_HIP_ERR_BAD_ARG;                         // evaluation challenge inside the domain (probability 2^-243): the barycentric formula divides by zero
    memcpy(out_commitment_g1, &agg_c, sizeof(g1j));
    memcpy(out_z_fr, &z, sizeof(fr));
    if (out_y_fr) memcpy(out_y_fr, &y, sizeof(fr));
    return KZG_HIP_OK;
    KZG_CATCH
}
// bls.EvaluatePolyInEvaluationForm (bls/globals.go:106-153) with rootsOfUnity = the settings' ExpandedRootsOfUnity[:MaxWidth] (the form of
// fft_fr_test.go:73-99), and eth.EvaluatePolynomialInEvaluationForm (eth/helpers.go:207-211: DomainFr, scale 0): the first half of the quotient kernel
static int evaluate_in_evaluation_form(kzg_hip_fft *fs, const fr *d_roots, uint64_t root_stride, const void *poly_fr, uint64_t n, const void *x_fr, void *out_y_fr) {
    stream_lease lease(fs);
    hipStream_t s = lease.s;
    fr y; uint32_t flag = 0;
    drain_on_exit drain(s);
    dtmp<fr> d_poly(s), d_x(s), d_y(s), d_q(s); dtmp<uint32_t> d_flag(s);
    CHK(d_poly.alloc(n)); CHK(d_x.alloc(1)); CHK(d_y.alloc(1)); CHK(d_q.alloc(n)); CHK(d_flag.alloc(1));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(d_poly.p, poly_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_x.p, x_fr, sizeof(fr), hipMemcpyHostToDevice, s));
    launch_eth_quotient(s, d_poly.p, n, d_roots, n, 1, d_x.p, 1, fs->d_inv_pow2 + ilog2(n), d_q.p, d_y.p, d_flag.p, root_stride);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&y, d_y.p, sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag) return KZG_HIP_ERR_BAD_ARG;                            // x inside the domain: the barycentric formula divides by zero
    memcpy(out_y_fr, &y, sizeof(fr));
    return KZG_HIP_OK;
}
int kzg_hip_evaluate_poly_in_evaluation_form(kzg_hip_fft *fs, const void *poly_fr, uint64_t n, const void *x_fr, uint32_t scale, void *out_y_fr) {
    if (!fs || !poly_fr || !x_fr || !out_y_fr) return KZG_HIP_ERR_BAD_ARG;
    if (scale > 63 || n != fs->W >> scale || !n) return KZG_HIP_ERR_LEN_MISMATCH;   // "expected roots of unity ... to match polynomial size", bls/globals.go:107-109
    KZG_TRY
    return evaluate_in_evaluation_form(fs, fs->d_expanded, 1ull << scale, poly_fr, n, x_fr, out_y_fr);
    KZG_CATCH
}
int kzg_hip_eth_evaluate_polynomial_in_evaluation_form(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *x_fr, void *out_y_fr) {
    if (!eth || !poly_fr || !x_fr || !out_y_fr) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;
    KZG_TRY
    return evaluate_in_evaluation_form(eth->fs, eth->d_domain, 1, poly_fr, n, x_fr, out_y_fr);
    KZG_CATCH
}
// test hook: SHA-256 of a host buffer through the transcript's implementation (needs no device)
void kzg_hip_test_sha256(const void *data, uint64_t len, void *out32) {
    sha256 h;
    h.update(data, len);
    h.final((uint8_t *)out32);
}

// ---------------------------------------------------------------------------------------------------------
// setup (un)marshalling (row f4): G1Point.MarshalText / UnmarshalText (bls/bls_all.go:20-39) and the JSON trusted setup of
// eth/globals.go:33-49.  Hex coding and JSON scanning are host work; decompression + subgroup check run on the device.
// ---------------------------------------------------------------------------------------------------------
static int hex_nibble(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
int kzg_hip_g1_marshal_text(kzg_hip_fft *fs, const void *points_g1, uint64_t n, char *out_hex96) {
    if (!fs || (n && (!points_g1 || !out_hex96))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> raw(48 * n);
    CHK(kzg_hip_g1_to_compressed(fs, points_g1, n, raw.data()));
    static const char dig[] = "0123456789abcdef";                       // hex.EncodeToString: lower case, no 0x prefix
    for (uint64_t i = 0; i < 48 * n; i++) { out_hex96[2 * i] = dig[raw[i] >> 4]; out_hex96[2 * i + 1] = dig[raw[i] & 15]; }
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_g1_unmarshal_text(kzg_hip_fft *fs, const char *hex96, uint64_t n, void *out_g1) {
    if (!fs || (n && (!hex96 || !out_g1))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> raw(48 * n);
    for (uint64_t i = 0; i < 48 * n; i++) {
        int hi = hex_nibble(hex96[2 * i]), lo = hex_nibble(hex96[2 * i + 1]);
        if (hi < 0 || lo < 0) return KZG_HIP_ERR_BAD_POINT;             // hex.DecodeString error (bls/bls_all.go:29-32)
        raw[i] = (uint8_t)(hi << 4 | lo);
    }
    return kzg_hip_g1_from_compressed(fs, raw.data(), n, out_g1);
    KZG_CATCH
}
// finds "key" : [ "..." , ... ] in `js` and appends the decoded 48-byte strings; *found = 0 when the key is absent
static int json_hex48_array(const char *js, uint64_t len, const char *key, std::vector<uint8_t> &out, uint64_t *count, int *found) {
    *count = 0; *found = 0;
    std::string pat = std::string("\"") + key + "\"";
    const char *end = js + len, *p = js;
    for (;;) {                                                           // the key must be followed by ':' (skips "setup_G1" inside "setup_G1_lagrange")
        p = std::search(p, end, pat.begin(), pat.end());
        if (p == end) return KZG_HIP_OK;
        p += pat.size();
        const char *q = p;
        while (q < end && (*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r')) q++;
        if (q < end && *q == ':') { p = q + 1; break; }
    }
    while (p < end && *p != '[') { if (*p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') return KZG_HIP_ERR_BAD_ARG; p++; }
    if (p == end) return KZG_HIP_ERR_BAD_ARG;
    p++;
    *found = 1;
    for (;;) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == ',')) p++;
        if (p == end) return KZG_HIP_ERR_BAD_ARG;
        if (*p == ']') return KZG_HIP_OK;
        if (*p != '"') return KZG_HIP_ERR_BAD_ARG;
        p++;
        const char *q = p;
        while (q < end && *q != '"') q++;
        if (q == end) return KZG_HIP_ERR_BAD_ARG;
        if (q - p != 96) return KZG_HIP_ERR_BAD_POINT;                   // FromCompressedG1 wants exactly 48 bytes
        for (int i = 0; i < 48; i++) {
            int hi = hex_nibble(p[2 * i]), lo = hex_nibble(p[2 * i + 1]);
            if (hi < 0 || lo < 0) return KZG_HIP_ERR_BAD_POINT;
            out.push_back((uint8_t)(hi << 4 | lo));
        }
        (*count)++;
        p = q + 1;
    }
}
int kzg_hip_trusted_setup_from_json(kzg_hip_fft *fs, const char *json, uint64_t json_len, void *out_setup_g1, void *out_lagrange_g1, uint64_t capacity,
                                    uint64_t *n_setup_g1, uint64_t *n_lagrange_g1) {
    if (!fs || !json || !n_setup_g1 || !n_lagrange_g1) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> mono, lagr;
    int f1 = 0, f2 = 0;
    CHK(json_hex48_array(json, json_len, "setup_G1", mono, n_setup_g1, &f1));
    CHK(json_hex48_array(json, json_len, "setup_G1_lagrange", lagr, n_lagrange_g1, &f2));
    if (!f1 && !f2) return KZG_HIP_ERR_BAD_ARG;                          // not a trusted-setup document
    if (out_setup_g1 && *n_setup_g1) {
        if (*n_setup_g1 > capacity) return KZG_HIP_ERR_LEN_MISMATCH;
        CHK(kzg_hip_g1_from_compressed(fs, mono.data(), *n_setup_g1, out_setup_g1));
    }
    if (out_lagrange_g1 && *n_lagrange_g1) {
        if (*n_lagrange_g1 > capacity) return KZG_HIP_ERR_LEN_MISMATCH;
        CHK(kzg_hip_g1_from_compressed(fs, lagr.data(), *n_lagrange_g1, out_lagrange_g1));
    }
    return KZG_HIP_OK;
    KZG_CATCH
}

// ---------------------------------------------------------------------------------------------------------
// instrumentation
// ---------------------------------------------------------------------------------------------------------
int kzg_hip_kzg_table_info(kzg_hip_kzg *ks, uint32_t *window_bits, uint32_t *windows, uint64_t *table_bytes) {
    if (!ks || !window_bits || !windows || !table_bytes) return KZG_HIP_ERR_BAD_ARG;
    bool have = ks->d_fixed != nullptr;
    *window_bits = have ? ks->fixed_plan.c : 0; *windows = have ? ks->fixed_plan.nwin : 0;
    *table_bytes = have ? (uint64_t)ks->fixed_plan.nwin * ks->fixed_plan.table_n * ks->fixed_plan.nb * sizeof(g1a) : 0;
    return KZG_HIP_OK;
}
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

}  // extern "C"
