// capi_core.hip -- library / device, FFTSettings and its transforms (a1-a5), conversions, bls.LinCombG1 and cached point sets (a6)
#include "capi_common.hpp"
#include <map>

thread_local std::string g_last_error;


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
    if (max_scale > KZG_HIP_MAX_SCALE) return KZG_HIP_ERR_UNSUPPORTED;   // refused, not attempted: the tables of scale 25+ are tens of GB per settings object
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
    stream_cache_own(fs->stream);
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
// ---- per-stream cache of small stream-ordered blocks (capi_common.hpp, dtmp) ----
namespace {
struct stream_blocks { std::vector<void *> cls[16]; };              // class k: 256 B << k (256 B ... 8 MiB)
std::mutex g_sc_mu;
std::map<hipStream_t, stream_blocks> g_sc;
inline int sc_index(size_t class_bytes) { int k = 0; while ((256u << k) < class_bytes) k++; return k; }
constexpr size_t SC_KEEP = 4;                                       // blocks kept per stream and class
}
void stream_cache_own(hipStream_t s) {   // (never throws: a stream that cannot be entered simply is not cached)
    try { std::lock_guard<std::mutex> lk(g_sc_mu); g_sc[s]; } catch (...) {}
}
void stream_cache_disown(hipStream_t s) {
    stream_blocks b;
    {
        std::lock_guard<std::mutex> lk(g_sc_mu);
        auto it = g_sc.find(s);
        if (it == g_sc.end()) return;
        b = std::move(it->second);
        g_sc.erase(it);
    }
    for (auto &v : b.cls) for (void *p : v) (void)hipFreeAsync(p, s);
}
void *stream_cache_take(hipStream_t s, size_t class_bytes) {
    std::lock_guard<std::mutex> lk(g_sc_mu);
    auto it = g_sc.find(s);
    if (it == g_sc.end()) return nullptr;
    auto &v = it->second.cls[sc_index(class_bytes)];
    if (v.empty()) return nullptr;
    void *p = v.back(); v.pop_back();
    return p;
}
bool stream_cache_give(hipStream_t s, void *p, size_t class_bytes) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_STREAM_CACHE"); return e && e[0] == '0'; }();   // A/B and test hook
    if (off) return false;
    std::lock_guard<std::mutex> lk(g_sc_mu);
    auto it = g_sc.find(s);
    if (it == g_sc.end()) return false;
    auto &v = it->second.cls[sc_index(class_bytes)];
    if (v.size() >= SC_KEEP) return false;
    try { v.push_back(p); } catch (...) { return false; }          // (called from destructors: the block is then freed the ordinary way)
    return true;
}
void lincomb_promo_free(kzg_hip_fft *fs);
void kzg_hip_fft_settings_free(kzg_hip_fft *fs) {
    if (!fs) return;
    hipSetDevice(fs->device);
    lincomb_promo_free(fs);                                 // promoted point sets hold tables and coalescers on this handle: they go first
    if (fs->stream) hipStreamSynchronize(fs->stream);
    hipFree(fs->d_expanded); hipFree(fs->d_reversed); hipFree(fs->d_expanded_l); hipFree(fs->d_reversed_l); hipFree(fs->d_inv_pow2); hipFree(fs->d_tw4096[0]); hipFree(fs->d_tw4096[1]); hipFree(fs->d_tw_das2048); hipFree(fs->d_glv_expanded); hipFree(fs->d_glv_reversed); hipFree(fs->d_wnaf_expanded); hipFree(fs->d_wnaf_reversed);
    if (fs->stream) { stream_cache_disown(fs->stream); hipStreamDestroy(fs->stream); }
    if (fs->h_stage) hipHostFree(fs->h_stage);
    for (auto &ps : fs->pool_idle) { hipStreamSynchronize(ps.s); stream_cache_disown(ps.s); hipStreamDestroy(ps.s); if (ps.h_pin) hipHostFree(ps.h_pin); }
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
void fr_fft_rows(kzg_hip_fft *fs, hipStream_t s, const fr *d_in, uint64_t in_stride, uint64_t n_in, fr *d_out, uint64_t n, uint64_t batch, int inv) {
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
uint32_t g1_fft_direct_logr(uint64_t n, uint64_t batch) {
    static const int forced = [] { const char *e = getenv("KZG_HIP_G1_FFT"); return !e ? 0 : (e[0] == 'd' ? 1 : 2); }();   // (initialised once, thread-safe)
    if (forced) return forced == 1 ? 4u : 0u;
    if (n < 2) return 0;
    // with four lanes per butterfly (g1_quad.hpp) the radix-2 network beats the direct passes from two transforms on (DAUsingFK20 on 2 / 4 polynomials:
    // 20.8 / 21.0 ms against 23.3 / 30.5 ms); a lone transform stays direct (15.1 ms against 20.7 ms)
    // (the passes themselves run on quads or pairs where that leaves no SIMD with two wavefronts, i.e. up to 2048 points: g1_fft_direct_lanes; 4096 points on
    // pairs would be four radix-8 passes of 1.8 ms, the same 7.1 ms as three radix-16 passes of 2.4 ms on single lanes: measured, not used)
    const uint64_t unit = device_simd_lanes() / 16;             // 4096 on 256 CUs: the points whose radix-16 pass is one wavefront per SIMD
    if (g1_quad_enabled()) return n * batch <= unit ? 4 : 0;
    if (n * batch <= 2 * unit) return 4;
    if (n * batch <= 4 * unit) return 3;
    return 0;
}
bool g1_fft_direct_mode(uint64_t n, uint64_t batch) { return g1_fft_direct_logr(n, batch) != 0; }
// lanes per (output, term) of a direct pass: as many as keep the pass at one wavefront per SIMD (65 536 lanes)
int g1_fft_direct_lanes(uint64_t n, uint64_t batch) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_G1_DIRECT_COOP"); return e && e[0] == '0'; }();
    if (!g1_quad_enabled() || off) return 1;
    const uint64_t items = (n * batch) << g1_fft_direct_logr(n, batch);
    const uint64_t one_round = device_simd_lanes();
    return items * 4 <= one_round ? 4 : items * 2 <= one_round ? 2 : 1;
}
// (n_out: the caller only reads the first n_out outputs -- the direct passes then skip the rest of their last pass; 0 = all)
int g1_fft_rows(kzg_hip_fft *fs, hipStream_t s, const g1j *d_in, uint64_t in_stride, uint64_t n_valid, g1j *d_data, uint64_t n, uint64_t batch, int inv,
                       const fr *scale, uint64_t n_out) {
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
int das_ext_rows(kzg_hip_fft *fs, hipStream_t s, fr *d, uint64_t n, uint64_t batch) {
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
// `batch` rows of n points each from host buffers: FFTG1 (fft_g1.go:58-94) on every row, one launch chain (a lone transform is latency-bound: 6.8 ms for 4096
// points against 0.46 ms per transform in a batch of 64)
int kzg_hip_fft_g1_batch(kzg_hip_fft *fs, const void *vals_g1, uint64_t n, uint64_t batch, int inv, void *out_g1) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;          // fft_g1.go:60-62
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;        // fft_g1.go:63-65
    if (n == 0 || !vals_g1 || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    stream_lease lease(fs);
    hipStream_t s = lease.s;
    dtmp<g1j> d_in(s), d_data(s);
    CHK(d_in.alloc(n * batch)); CHK(d_data.alloc(n * batch));
    HIPCHK(hipMemcpyAsync(d_in.p, vals_g1, n * batch * sizeof(g1j), hipMemcpyHostToDevice, s));
    launch_g1_from_kilic(s, d_in.p, n * batch);
    CHK(g1_fft_rows(fs, s, d_in.p, n, n, d_data.p, n, batch, inv, inv ? fs->d_inv_pow2 + ilog2(n) : nullptr));
    launch_g1_normalize(s, d_data.p, d_in.p, n * batch, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_g1, d_in.p, n * batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
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
msm_plan classic_plan(uint64_t n, bool folded) {
    msm_plan p{};
    p.c = 8; p.nwin = 16; p.nb = 128; p.ngroups = folded ? 8 : 16; p.fixed = 0; p.table_n = n;
    return p;
}
void set_inf_image(void *out_g1) { g1j z = g1_to_kilic(g1_inf()); memcpy(out_g1, &z, sizeof z); }   // Kilic Zero(): (0, R, 0)

// ---- cached point sets for bls.LinCombG1 (bls/bls_kilic.go:132-150): callers such as CommitToEvalPoly (kzg_single_proofs.go:12-14,
// the IFFT of the setup) and eth/helpers.go:99,159,199 (the Lagrange setup) multiply the SAME points by fresh scalars every call.
// The handle keeps them in HBM as affine device-internal images together with 2^64 P_i, which folds the 16 windows of each GLV
// half onto 8 bucket groups: 56 instead of 120 doublings on the critical path of a lone MSM.
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
// holds_mu: the caller already owns pts->fs->mu (a stream_lease in fallback mode): std::mutex is not recursive
int lincomb_points_rows(kzg_hip_points *pts, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride, bool holds_mu) {
    if (pts->ks) {   // the cached set's fixed-base table, when its budget allows one: n x windows mixed additions per combination, no sort, no buckets
        if (holds_mu) CHK(ensure_fixed_table(pts->ks, s));
        else { dev_guard g(pts->fs); CHK(ensure_fixed_table(pts->ks, s)); }
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
    CHK(lincomb_points_rows(pts, s, d_sc.p, n, batch, d_out.p, 0, lease.fallback.owns_lock()));
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
int kzg_hip_lincomb_points(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (pts && scalars_fr && out_g1 && n && n <= pts->n) return lincomb_points_coalesced(pts, scalars_fr, n, out_g1);
    return kzg_hip_lincomb_points_batch(pts, scalars_fr, n, 1, out_g1);
}

// ---- bls.LinCombG1 on caller-supplied points: promotion of a REPEATED point set to a cached one (round 6) ----
// The reference's call sites hand the same slice again and again (bls.LinCombG1(setup, coeffs): kzg_single_proofs.go:17-19, eth/helpers.go:99,159,199), but the C
// signature carries no identity, so every call paid the one-shot pipeline: upload 590 KB of points, convert, sort into buckets, 120 dependent doublings -- 1.3 ms for
// 4096 points against 0.25 ms on a cached set (kzg_hip_points: resident rows + a fixed-base table).  The handle therefore remembers the last few point sets it was
// given and promotes one that keeps coming back:
//   sighting 1: a 64-bit fingerprint of (n, first and last 4 KiB) -- costs a microsecond, lets one-off callers pass untouched;
//   sighting 2: fingerprint known -> the points are COPIED (host memory, n x 144 B);
//   sighting 3 (KZG_HIP_LINCOMB_PROMOTE_AFTER + 1): fingerprint known and memcmp against the copy equal -> kzg_hip_points_new on the copy, once (0.1-0.3 s: the
//              table of the set is built; this call is the slow one), and from then on every call whose points compare EQUAL, byte for byte, to the copy runs on the set.
// Identity is established by the full comparison on EVERY call (n x 144 B memcmp: 30-50 us for 4096 points), never by the fingerprint: a caller that changes one
// coordinate between calls compares unequal and takes the one-shot path with its new points (tests/test_gpu_parity.py).  Results are the same group element either way
// and both paths return the normalised image, so the bytes do not depend on which ran.  At most LINCOMB_PROMO_SETS promoted sets per handle (least recently used is
// freed), each within KZG_HIP_POINTS_FB_BUDGET_GB of HBM (default min(32 GB, free - 24 GB)).  KZG_HIP_LINCOMB_PROMOTE=0 switches the whole mechanism off.
struct lincomb_promo {
    static constexpr int SLOTS = 6, LINCOMB_PROMO_SETS = 2;
    struct entry {
        uint64_t n = 0, fp = 0, last_use = 0; uint32_t sightings = 0;
        std::shared_ptr<const std::vector<uint8_t>> copy;                  // the points as first compared (sighting 2 on); immutable once taken, so calls compare against it OUTSIDE the lock
        std::shared_ptr<kzg_hip_points> set;                               // promoted: the cached set (kept alive by calls in flight)
        bool building = false;
    };
    std::mutex mu; std::condition_variable built;
    entry e[SLOTS];
    uint64_t tick = 0, promoted = 0;
    std::atomic<uint64_t> served{0};
};
namespace {
uint64_t lincomb_fingerprint(const uint8_t *p, uint64_t n) {
    const size_t bytes = (size_t)n * sizeof(g1j), head = bytes < 4096 ? bytes : 4096;
    uint64_t h = 0x9e3779b97f4a7c15ull ^ n;
    auto eat = [&](const uint8_t *q, size_t len) {
        for (size_t i = 0; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, q + i, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
    };
    eat(p, head);
    if (bytes > head) eat(p + bytes - head, head);
    return h;
}
bool lincomb_promotion_enabled() { static const bool on = [] { const char *e = getenv("KZG_HIP_LINCOMB_PROMOTE"); return !(e && e[0] == '0'); }(); return on; }
uint32_t lincomb_promote_after() { static const uint32_t v = [] { const char *e = getenv("KZG_HIP_LINCOMB_PROMOTE_AFTER"); long x = e ? atol(e) : 2; return (uint32_t)(x < 1 ? 1 : x > 1000 ? 1000 : x); }(); return v; }
// the cached set for these points if they are a promoted set (or become one with this call); null: take the one-shot path
std::shared_ptr<kzg_hip_points> lincomb_promoted_set(kzg_hip_fft *fs, const void *points_g1, uint64_t n) {
    if (!lincomb_promotion_enabled() || n < 64 || n > (1u << 20)) return nullptr;
    const uint8_t *pb = (const uint8_t *)points_g1;
    const size_t bytes = (size_t)n * sizeof(g1j);
    const uint64_t fpv = lincomb_fingerprint(pb, n);
    {   // the memory is created under the handle's mutex (first call only)
        std::lock_guard<std::mutex> lk(fs->mu);
        if (!fs->promo) fs->promo = new lincomb_promo;
    }
    lincomb_promo &pr = *fs->promo;
    std::unique_lock<std::mutex> lk(pr.mu);
    pr.tick++;
    lincomb_promo::entry *hit = nullptr;
    for (auto &x : pr.e) if (x.n == n && x.fp == fpv && x.sightings) { hit = &x; break; }
    if (!hit) {   // sighting 1: remember the fingerprint in the least recently used slot that holds no promoted set (those leave only through the set limit below)
        lincomb_promo::entry *v = nullptr;
        for (auto &x : pr.e) if (!x.set && !x.building && (!v || x.last_use < v->last_use)) v = &x;
        if (v) { v->n = n; v->fp = fpv; v->sightings = 1; v->last_use = pr.tick; v->copy.reset(); }
        return nullptr;
    }
    hit->last_use = pr.tick;
    if (!hit->copy) {   // sighting 2: take the copy every later call is compared with
        hit->copy = std::make_shared<const std::vector<uint8_t>>(pb, pb + bytes);
        hit->sightings = 2;
        return nullptr;
    }
    while (hit->building) pr.built.wait(lk);                              // another thread is promoting this very set: wait for its table rather than build a second one
    if (hit->set) {   // the steady state: compare OUTSIDE the lock (the copy is immutable, both objects are kept alive by the shared pointers), so concurrent callers of one set do not queue behind each other's 590 KB memcmp
        std::shared_ptr<const std::vector<uint8_t>> cp = hit->copy;
        std::shared_ptr<kzg_hip_points> set = hit->set;
        lk.unlock();
        if (cp->size() != bytes || memcmp(cp->data(), pb, bytes) != 0) return nullptr;          // same fingerprint, different points (or changed in place): one-shot
        pr.served.fetch_add(1, std::memory_order_relaxed);
        return set;
    }
    if (hit->copy->size() != bytes || memcmp(hit->copy->data(), pb, bytes) != 0) return nullptr;   // same fingerprint, different points (or changed in place): one-shot
    if (++hit->sightings <= lincomb_promote_after() + 0u) return nullptr;
    // promote: build the cached set from the copy (outside the lock: other sets keep being served)
    hit->building = true;
    std::vector<std::shared_ptr<kzg_hip_points>> evicted;                  // freed outside the lock, after their last call in flight
    {
        int have = 0; lincomb_promo::entry *old = nullptr;
        for (auto &x : pr.e) if (x.set) { have++; if (!old || x.last_use < old->last_use) old = &x; }
        if (have >= lincomb_promo::LINCOMB_PROMO_SETS && old) { evicted.push_back(std::move(old->set)); old->set.reset(); old->sightings = 0; old->n = 0; old->copy.reset(); }
    }
    std::shared_ptr<const std::vector<uint8_t>> src_keep = hit->copy;
    const uint8_t *src = src_keep->data();
    lk.unlock();
    evicted.clear();
    kzg_hip_points *raw = nullptr;
    const int st = kzg_hip_points_new(fs, src, n, &raw);
    lk.lock();
    hit->building = false;
    if (st == KZG_HIP_OK && raw) { hit->set = std::shared_ptr<kzg_hip_points>(raw, kzg_hip_points_free); pr.promoted++; }
    else hit->sightings = 2;                                               // could not be built (memory): stays on the one-shot path, tried again later
    pr.built.notify_all();
    return hit->set;
}
}  // namespace
void lincomb_promo_free(kzg_hip_fft *fs) { delete fs->promo; fs->promo = nullptr; }
int kzg_hip_lincomb_promotions(kzg_hip_fft *fs, uint64_t *promoted, uint64_t *served) {
    if (!fs) return KZG_HIP_ERR_BAD_ARG;
    uint64_t a = 0, b = 0;
    { std::lock_guard<std::mutex> lk0(fs->mu); if (fs->promo) { std::lock_guard<std::mutex> lk(fs->promo->mu); a = fs->promo->promoted; b = fs->promo->served.load(); } }
    if (promoted) *promoted = a;
    if (served) *served = b;
    return KZG_HIP_OK;
}

int kzg_hip_lincomb_g1(kzg_hip_fft *fs, const void *points_g1, const void *scalars_fr, uint64_t n, void *out_g1) {
    if (!fs || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n == 0) { set_inf_image(out_g1); return KZG_HIP_OK; }   // bls/bls_test.go:69-78
    if (!points_g1 || !scalars_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    if (std::shared_ptr<kzg_hip_points> set = lincomb_promoted_set(fs, points_g1, n))   // the same points as before, byte for byte: the cached set's table walk
        return kzg_hip_lincomb_points(set.get(), scalars_fr, n, out_g1);
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
    KZG_CATCH
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
