// capi_kzg.hip -- KZGSettings: commitments, single and multi proofs, Toeplitz parts, the coalesced one-polynomial entry points (a7-a9, f2)
#include "capi_common.hpp"

// ---------------------------------------------------------------------------------------------------------
// KZGSettings
// ---------------------------------------------------------------------------------------------------------
// uploads n Kilic images, converts to the device-internal domain, normalises and keeps Jacobian + affine copies resident.
// Every error path frees what was built (the handle is owned by a unique_ptr until the last step).
int kzg_settings_build(kzg_hip_fft *fs, const void *points_g1, uint64_t n, kzg_hip_kzg **out) {
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
    if (ks->copy_stream) { stream_cache_disown(ks->copy_stream); hipStreamDestroy(ks->copy_stream); }
    for (int i = 0; i < 2; i++) if (ks->copy_done[i]) hipEventDestroy(ks->copy_done[i]);
    (void)hipGetLastError();
    delete ks;
}

// number of signed c-bit windows of a canonical scalar (< r < 2^255): ceil(255 / c), plus one only if the top window's
// digit (top bits of r - 1, plus the incoming carry) can exceed 2^(c-1) and carry out (c = 15 carries: 18 windows, c = 16 does not: 16)
// table budget in GB: the environment override, else min(cap, free HBM - headroom)
double table_budget_gb(const char *env, double cap_gb, double headroom_gb) {
    if (const char *e = getenv(env)) return atof(e);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0.0;
    double g = (double)free_b / 1e9 - headroom_gb;
    return g > cap_gb ? cap_gb : (g > 0.0 ? g : 0.0);
}
uint32_t fb_windows(uint32_t c) {
    uint32_t nw = (255 + c - 1) / c, sh = c * (nw - 1);
    uint64_t top = (0x73eda753299d7d48ull >> (sh - 192)) + 1;
    return top > (1ull << (c - 1)) ? nw + 1 : nw;
}

// Lazily builds the fixed-base table T[(w n + i) D + d - 1] = d 2^(c w) SecretG1[i] (k_msm.hip).  The window size is the
// largest whose table fits the HBM budget.  Budget, in this order: kzg_hip_kzg_set_table_budget_gb (per handle), the KZG_HIP_FB_BUDGET_GB
// environment variable, else the DEFAULT of 110 GB clipped to free HBM - 24 GB.  Since round 5 the walk uses the endomorphism (k_fb_accumulate_glv:
// both GLV halves of a scalar walk the SAME rows), so n = 4096 gets c = 16 with 8 windows = 103 GB and 16 additions per point by default -- what
// took the 206 GB opt-in before -- and the monomial setup (103 GB), eth's Lagrange setup (103 GB) and FK20 settings (32-48 GB) still co-reside in
// 288 GB.  Smaller budgets: 58 GB c = 15 (2 x 9 additions), 16 GB c = 13 (2 x 10), 8.9 GB c = 12 (2 x 11), 4.8 GB c = 11 (2 x 12).
// (KZG_HIP_FB_GLV=0 restores the plain layout: 206 GB c = 16 / 61 GB c = 14 (19) / 32 GB c = 13 (20) / 9.7 GB c = 11 (24).)
// If the allocation fails (another process on the GPU, fragmentation)
// the next smaller window is tried, and finally the bucket path, which needs no table: a commitment never fails for lack of HBM.
// The build runs on the HANDLE's stream and only that stream is waited for (by the host thread that found no table): a caller's stream
// passed to a _dev entry point is never synchronised here -- its work already enqueued keeps running under the build.  Callers hold fs->mu.
#ifndef FB_DEFAULT_BUDGET_GB
#define FB_DEFAULT_BUDGET_GB 110.0
#endif
// windows of the GLV walk: both halves of a split scalar are below 2^126.5 (glv_split_signed), so the top signed digit cannot carry out as soon as
// c * nwin >= 128
uint32_t fb_windows_glv(uint32_t c) { return (128 + c - 1) / c; }
bool fb_glv_enabled() {
    static const bool on = [] { const char *e = getenv("KZG_HIP_FB_GLV"); return !(e && !strcmp(e, "0")); }();   // "0": the round-1..4 layout (one window per c bits of the whole scalar), A/B runs and tests
    return on;
}
int ensure_fixed_table(kzg_hip_kzg *ks, hipStream_t) {
    if (ks->d_fixed || ks->fixed_plan.c == 0xffffffffu) return KZG_HIP_OK;
    hipStream_t s = ks->fs->stream;
    double budget_gb = ks->budget_gb >= 0.0 ? ks->budget_gb : table_budget_gb("KZG_HIP_FB_BUDGET_GB", FB_DEFAULT_BUDGET_GB, 24.0);
    if (ks->n_setup < 64) { ks->fixed_plan.c = 0xffffffffu; return KZG_HIP_OK; }   // classic path only
    const bool glv = fb_glv_enabled();
    for (uint32_t c = 16; c >= 5; c--) {
        // window sizes that are dominated by a smaller table with the same number of additions per point are skipped: GLV c = 14 (2 x 10 windows, 32 GB)
        // against c = 13 (2 x 10, 16 GB); plain c = 15 (18 windows, 116 GB) measured slower than c = 14 (19 windows, 61 GB)
        if (glv ? (c == 14) : (c == 15)) continue;
        const uint32_t nwin = glv ? fb_windows_glv(c) : fb_windows(c);
        double bytes = (double)nwin * (double)ks->n_setup * (double)(1u << (c - 1)) * sizeof(g1a);
        if (bytes > budget_gb * 1e9) continue;
        msm_plan p{};
        p.c = c; p.nwin = nwin; p.nb = 1u << (c - 1); p.ngroups = 1; p.fixed = 1; p.glv = glv ? 1 : 0; p.table_n = ks->n_setup;
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
int commit_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_sc, uint64_t n, uint64_t batch, g1j *d_out, uint64_t sc_stride, bool to_kilic) {
    CHK(ensure_fixed_table(ks, s));
    bool fixed = ks->d_fixed != nullptr;
    if (!sc_stride) sc_stride = n;
    msm_plan p = fixed ? ks->fixed_plan : classic_plan(ks->n_setup);
    if (!fixed && !msm_index_range_ok(p, n)) return KZG_HIP_ERR_TOO_WIDE;
    size_t ws_main = fixed ? fb_partials_bytes(n, batch) : msm_workspace_bytes(p, n, batch);
    dtmp<uint8_t> d_ws(s);
    CHK(d_ws.alloc(ws_main));
    // to_kilic = false (the eth byte paths: their results go on to the compression kernel): normalised points stay in the device-internal domain, no conversion back and forth
    if (fixed) launch_fb_msm(s, ks->d_fixed, p.table_n, p.c, p.nwin, d_sc, sc_stride, n, batch, d_ws.p, d_out, to_kilic, p.glv != 0, to_kilic && ks->projective.load(std::memory_order_relaxed));   // sums, normalises (unless projective), converts
    else launch_msm(s, p, ks->d_secret_a, d_sc, sc_stride, n, batch, d_ws.p, d_out, to_kilic);
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
// Ranges pinned through kzg_hip_host_register: the in-place paths need the EXTENT of a pinned range, which hipPointerGetAttributes does not report.
namespace {
std::mutex g_reg_mu;
std::map<uintptr_t, size_t> g_registered;   // base -> bytes
}
// device-visible address of the host range [host, host + bytes) if ALL of it lies in pinned, mapped memory (registered with kzg_hip_host_register, or
// from hipHostMalloc / hipHostRegister where the runtime reports the allocation's extent); null for pageable memory and for a range that starts in a
// pinned allocation but runs past its end (a kernel reading it in place would fault: such input takes the staged copy).  The current device must be the handle's.
const void *host_mapped_pointer(const void *host, size_t bytes) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // pageable memory is "invalid value" for this query
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    const uintptr_t h = (uintptr_t)host;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_registered.upper_bound(h);
        if (it != g_registered.begin()) {
            --it;
            if (h >= it->first && h < it->first + it->second) return (h + bytes <= it->first + it->second) ? a.devicePointer : nullptr;
        }
    }
    // pinned by someone else (hipHostMalloc, a framework's pinned allocator): trust the runtime's extent if it has one, otherwise stage
    hipDeviceptr_t base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const uintptr_t d = (uintptr_t)a.devicePointer, b = (uintptr_t)base;
    return (d >= b && d + bytes <= b + size) ? a.devicePointer : nullptr;
}
// Host -> device copy of a range that may STRADDLE the end (or the start) of a range pinned through kzg_hip_host_register: the runtime refuses one
// hipMemcpyAsync over pinned and pageable pages together ("invalid argument"), so the copy is cut at the registered boundaries.
int h2d_copy(void *dst, const void *src, size_t bytes, hipStream_t s) {
    uintptr_t p = (uintptr_t)src; uint8_t *d = (uint8_t *)dst;
    while (bytes) {
        size_t chunk = bytes;
        {
            std::lock_guard<std::mutex> lk(g_reg_mu);
            auto it = g_registered.upper_bound(p);                         // first range that starts after p
            if (it != g_registered.end() && it->first - p < chunk) chunk = it->first - p;
            if (it != g_registered.begin()) {
                --it;
                if (p < it->first + it->second && it->first + it->second - p < chunk) chunk = it->first + it->second - p;   // p lies inside this range: stop at its end
            }
        }
        HIPCHK(hipMemcpyAsync(d, (const void *)p, chunk, hipMemcpyHostToDevice, s));
        p += chunk; d += chunk; bytes -= chunk;
    }
    return KZG_HIP_OK;
}
int kzg_hip_host_register(void *host, uint64_t bytes) {
    if (!host || !bytes) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    {   // a range that overlaps a live registration is refused before the runtime sees it: overwriting the tracked extent of a live base (or nesting two extents)
        // would make h2d_copy cut copies at the wrong boundary and host_mapped_pointer vouch for pages that are not pinned
        std::lock_guard<std::mutex> lk(g_reg_mu);
        const uintptr_t h = (uintptr_t)host;
        auto it = g_registered.lower_bound(h);
        if (it != g_registered.end() && it->first < h + bytes) { g_last_error = "kzg_hip_host_register: the range overlaps a registered one (unregister it first)"; return KZG_HIP_ERR_BAD_ARG; }
        if (it != g_registered.begin()) { --it; if (it->first + it->second > h) { g_last_error = "kzg_hip_host_register: the range overlaps a registered one (unregister it first)"; return KZG_HIP_ERR_BAD_ARG; } }
    }
    HIPCHK(hipHostRegister(host, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_registered[(uintptr_t)host] = bytes;
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_host_unregister(void *host) {
    if (!host) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    // the extent is forgotten only once the runtime has let go of the pages: after a failed hipHostUnregister they are still pinned, and h2d_copy /
    // host_mapped_pointer must keep cutting and checking at this range's boundaries.  An address INSIDE a tracked range that is not its base never reaches the
    // runtime (ROCm 7.2 aborts the process on it instead of returning an error)
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        const uintptr_t h = (uintptr_t)host;
        auto it = g_registered.upper_bound(h);
        if (it != g_registered.begin()) {
            --it;
            if (h > it->first && h < it->first + it->second) { g_last_error = "kzg_hip_host_unregister: the address lies inside a registered range but is not its base"; return KZG_HIP_ERR_BAD_ARG; }
        }
    }
    HIPCHK(hipHostUnregister(host));
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_registered.erase((uintptr_t)host);
    return KZG_HIP_OK;
    KZG_CATCH
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
    CHK(d_out.alloc(batch));
    // Pinned input (kzg_hip_host_register, hipHostMalloc): the walk reads the coefficients IN PLACE over PCIe -- each scalar is loaded exactly once, 128 KiB per
    // blob = 13 GB/s at 100 k commitments/s -- instead of waiting for a staged copy of pageable memory (the calling thread copies at ~10 GB/s: 67-78 k/s)
    if (const fr *mapped = (const fr *)host_mapped_pointer(coeffs_fr, (size_t)n * batch * sizeof(fr))) {
        CHK(commit_rows(ks, s, mapped, n, batch, d_out.p));
        HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return KZG_HIP_OK;
    }
    CHK(d_sc.alloc(n * batch));
    // Large batches are uploaded in chunks on a second stream: the copy of chunk i + 1 (from pageable host memory it occupies the
    // calling thread) runs while the GPU walks chunk i.  Chunks keep >= 256 blobs so that a walk still fills one round of waves.
    uint64_t chunk = batch >= 1024 ? 512 : (batch >= 512 ? 256 : batch);
    if (chunk < batch && n * sizeof(fr) >= (64u << 10)) {
        if (!ks->copy_stream) {
            HIPCHK(hipStreamCreateWithFlags(&ks->copy_stream, hipStreamNonBlocking));
            stream_cache_own(ks->copy_stream);
            for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ks->copy_done[i], hipEventDisableTiming));
        }
        CHK(ensure_fixed_table(ks, s));
        HIPCHK(hipStreamSynchronize(s));                     // d_sc was allocated in order on s: make it visible to the copy stream
        // The copy of the FIRST chunk overlaps with nothing, so the schedule ramps up: 64 blobs, then 192, then `chunk` at a time (KZG_HIP_UPLOAD_RAMP=0: equal
        // chunks, as through round 5) -- the walk starts after 8 MiB instead of 64 MiB of pageable copy.
        static const bool ramp_on = [] { const char *e = getenv("KZG_HIP_UPLOAD_RAMP"); return !(e && e[0] == '0'); }();
        const bool ramp = ramp_on && batch >= 1024;          // (measured: 1024 blobs 78.5 -> 83.9 k/s, 4096 blobs 86.1 -> 86.9 k/s; 512 blobs are better off with two equal chunks: 79.4 vs 76.5 k/s)
        int slot = 0;
        uint64_t step = 0;
        for (uint64_t b0 = 0, cnt = 0; b0 < batch; b0 += cnt, slot ^= 1, step++) {
            uint64_t want = chunk;
            if (ramp && step == 0) want = 64; else if (ramp && step == 1) want = 192;
            cnt = batch - b0 < want ? batch - b0 : want;
            CHK(h2d_copy(d_sc.p + b0 * n, (const fr *)coeffs_fr + b0 * n, n * cnt * sizeof(fr), ks->copy_stream));
            HIPCHK(hipEventRecord(ks->copy_done[slot], ks->copy_stream));
            HIPCHK(hipStreamWaitEvent(s, ks->copy_done[slot], 0));
            CHK(commit_rows(ks, s, d_sc.p + b0 * n, n, cnt, d_out.p + b0));
        }
    } else {
        CHK(h2d_copy(d_sc.p, coeffs_fr, n * batch * sizeof(fr), s));
        CHK(commit_rows(ks, s, d_sc.p, n, batch, d_out.p));
    }
    HIPCHK(hipMemcpyAsync(out_g1, d_out.p, batch * sizeof(g1j), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return KZG_HIP_OK;
}
// ---- one-polynomial calls: concurrent callers on a handle are merged into batched launches (coalesce.hpp) ----
bool coalescing_enabled() {
    static const bool on = [] { const char *e = getenv("KZG_HIP_COALESCE"); return !(e && e[0] == '0'); }();
    return on;
}
// rows per staging buffer: as many as fit 32 MiB of pinned memory per direction, within [4, 256]
static uint64_t coalesce_rows(size_t in_row, size_t out_row) {
    size_t row = in_row > out_row ? in_row : out_row;
    uint64_t r = (64u << 20) / (row ? row : 1);          // 64 MiB of pinned rows per staging buffer: 113 rows of 4096 proofs (a 56-row cap split 64 callers 56 + 8)
    return r < 4 ? 4 : (r > 256 ? 256 : r);
}
coalescer *get_coalescer(kzg_hip_fft *fs, std::unique_ptr<coalescer> &slot, size_t in_row, size_t out_row, int callers_per_batch) {
    std::lock_guard<std::mutex> lk(fs->mu);
    if (!slot) slot.reset(new coalescer(fs->device, in_row, out_row, coalesce_rows(in_row, out_row), callers_per_batch));
    return slot.get();
}
// uploads the batch's rows (pinned, row stride in_row_bytes) as dense n_max-wide rows and zero-fills the tails
int coalesce_upload_rows(coalesce_buf &b, uint64_t batch, size_t in_row_bytes, uint64_t n_max, fr *d_rows, uint64_t *d_meta) {
    hipStream_t s = b.stream;
    HIPCHK(hipMemcpyAsync(d_meta, b.h_meta, batch * sizeof(coalesce_row), hipMemcpyHostToDevice, s));
    // (plain copies where they can do it: hipMemcpy2DAsync costs the host and the copy engine more than a few 1-D copies -- its device -> host form held a lone
    // eth.ComputeKZGProof at 0.449 ms in round 6 until it was replaced)
    if (in_row_bytes == n_max * sizeof(fr)) HIPCHK(hipMemcpyAsync(d_rows, b.h_in, batch * in_row_bytes, hipMemcpyHostToDevice, s));
    else if (batch <= 4) { for (uint64_t r = 0; r < batch; r++) HIPCHK(hipMemcpyAsync(d_rows + r * n_max, b.h_in + r * in_row_bytes, n_max * sizeof(fr), hipMemcpyHostToDevice, s)); }
    else HIPCHK(hipMemcpy2DAsync(d_rows, n_max * sizeof(fr), b.h_in, in_row_bytes, n_max * sizeof(fr), batch, hipMemcpyHostToDevice, s));
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
int lincomb_points_coalesced(kzg_hip_points *pts, const void *scalars_fr, uint64_t n, void *out_g1) {
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
int proof_single_rows(kzg_hip_kzg *ks, hipStream_t s, const fr *d_poly, uint64_t n, uint64_t batch, const uint64_t *d_x_u64, uint64_t x_stride, g1j *d_out) {
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
        void *dp_out = nullptr;                                                    // results go straight into the pinned output rows
        HIPCHK(hipHostGetDevicePointer(&dp_out, b.h_out, 0));
        if (batch == 1) {
            // a lone caller: the quotient kernel reads each coefficient once, so the one row (and its x) is read IN PLACE from the pinned staging area -- no upload, no
            // zero-tail launch (a single row has no tail: n_max is its own length)
            void *dp_in = nullptr, *dp_meta = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp_in, b.h_in, 0));
            HIPCHK(hipHostGetDevicePointer(&dp_meta, b.h_meta, 0));
            CHK(proof_single_rows(ks, s, (const fr *)dp_in, n_max, 1, (const uint64_t *)dp_meta + 1, 2, (g1j *)dp_out));
            HIPCHK(hipStreamSynchronize(s));
            return KZG_HIP_OK;
        }
        dtmp<fr> d_rows(s); dtmp<uint64_t> d_meta(s);
        CHK(d_rows.alloc(batch * n_max)); CHK(d_meta.alloc(2 * batch));
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

// shape of the fixed-base table (window bits, windows, bytes); zeros before the first commitment or when no table fits
int kzg_hip_kzg_table_info(kzg_hip_kzg *ks, uint32_t *window_bits, uint32_t *windows, uint64_t *table_bytes) {
    if (!ks || !window_bits || !windows || !table_bytes) return KZG_HIP_ERR_BAD_ARG;
    bool have = ks->d_fixed != nullptr;
    *window_bits = have ? ks->fixed_plan.c : 0; *windows = have ? ks->fixed_plan.nwin : 0;
    *table_bytes = have ? (uint64_t)ks->fixed_plan.nwin * ks->fixed_plan.table_n * ks->fixed_plan.nb * sizeof(g1a) : 0;
    return KZG_HIP_OK;
}
// The reference's G1Point IS a Jacobian triple (bls/bls_kilic.go:30-35) and CommitToPoly / ComputeProofSingle return whatever Z their additions left; this
// library normalises every result (Z = one) by default, which costs one F_p inversion per result: ~110 us of pure latency on one lane, a third of a lone
// CommitToPoly.  on != 0: results of the fixed-base table walk on THIS settings object (CommitToPoly, ComputeProofSingle, their batch, _dev and coalesced
// forms) leave as (X ZZ, Y ZZZ, ZZ) -- the same group element, Z != one; the bucket fallback (no table) still normalises.  eth and cached point sets hold
// settings objects of their own and are not affected.
int kzg_hip_kzg_set_projective_outputs(kzg_hip_kzg *ks, int on) {
    if (!ks) return KZG_HIP_ERR_BAD_ARG;
    ks->projective.store(on != 0, std::memory_order_relaxed);
    return KZG_HIP_OK;
}
// mixed additions per coefficient of a commitment on that table: 2 x windows when both GLV halves of a scalar walk it (the default), else windows; 0 without a table
uint32_t kzg_hip_kzg_table_additions(kzg_hip_kzg *ks) {
    if (!ks || !ks->d_fixed) return 0;
    return ks->fixed_plan.glv ? 2 * ks->fixed_plan.nwin : ks->fixed_plan.nwin;
}
