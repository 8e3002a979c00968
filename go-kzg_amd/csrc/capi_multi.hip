// capi_multi.hip -- several GPUs behind ONE handle of the C ABI (threading row: "multi-GPU handle owns one context per device"; 8e).
//
// The reference is a single-process library (kzg.go:11-19, fk20_multi.go:58-109): a drop-in that uses the 8 GPUs of a node has to do so
// inside the library.  A kzg_hip_multi owns one FFTSettings + KZGSettings (+ tables) per listed device and offers
//   * batch calls, sharded by polynomial (contiguous shares, a host thread per device on the single-device entry points, results written
//     straight into the caller's buffer): no data-path collective;
//   * ONE DAUsingFK20 / DAUsingFK20Multi split over the devices: the Toeplitz stage by output position, an ALL-GATHER of the
//     hExtFFT slices, and then either the two G1 transforms on the first device ("gather") or both transforms sharded by decimation with
//     two more all-gathers each (5 in total, the scheme of SURVEY.md 8e; "sharded").
// The all-gather is ncclAllGather on ncclUint8 over single-process communicators (ncclCommInitAll; RCCL over xGMI) when every listed device
// is distinct; a list that repeats a device (a 1-GPU box testing [0, 0]) or a process without RCCL exchanges the same slices with
// hipMemcpyPeerAsync.  RCCL is bound at the first multi-device handle, not at load time: librccl.so is 573 MB, single-GPU callers never
// need it, and a process that already holds a copy (PyTorch bundles one with the same SONAME) must not get a second one.
//
// What may be handed to another device (round 5).  Every buffer that an all-gather reads or writes lives in the handle's EXCHANGE ARENA of
// its device: one block from hipMalloc (never the stream-ordered pool, whose memory is not mapped to peers unless hipMemPoolSetAccess says
// so), with hipDeviceEnablePeerAccess granted for every distinct pair of listed devices at creation.  all_gather_bytes only accepts `xbuf`
// values, and only the arena makes them, so no hipMallocAsync pointer can reach ncclAllGather or a peer copy.  The handle PROVES its
// transport when it is created (transport_self_test): every entry writes a pattern into its slice, one all_gather_bytes, every entry's
// buffer is read back and compared; a transport that errs or delivers wrong bytes is replaced by the next one -- rccl -> peer-copy ->
// host-staged (device -> pinned host -> device, no peer mapping involved) -- with the reason in kzg_hip_multi_transport_note.
// KZG_HIP_MULTI_FAULT (tests only; comma list of "rccl", "rccl-corrupt", "rccl-hang", "peer", "peer-corrupt", "peer-hang") makes the named leg fail so
// that the fall-backs run on a one-GPU box ("rccl-block" / "rccl-init-block": the RCCL calls block on the HOST side).
//
// A transport that HANGS (round 6).  The usual failure of a misconfigured RCCL / P2P path is not an error code but a collective that never completes.  The
// probe therefore never blocks on a stream: it polls hipStreamQuery on every entry's stream against a deadline (KZG_HIP_MULTI_PROBE_TIMEOUT_MS, default
// 10 000).  On a timeout the communicators are aborted (ncclCommAbort, on a helper thread that is itself given the deadline), the entries' streams, events
// and arenas that the stuck work may still touch are ABANDONED (never synchronised, freed only if they have drained by the time the handle is freed) and
// replaced, and the next transport is probed on the fresh ones; the note names the timeout.  "rccl-hang" / "peer-hang" enqueue a kernel that spins on a
// host flag nobody sets (bounded by its own clock: at most ~2 x the probe deadline, so it cannot wedge a box) in front of the exchange; the flag is set where a real
// abort would take effect (before ncclCommAbort) resp. after the streams have been abandoned (the stuck peer exchange then completes late, into the abandoned arenas).
// A stream that stays stuck for good may hold the hardware queue it shares with fresh streams: then the next probes time out too and the constructor returns the
// error after at most three deadlines -- bounded either way.
#include "capi_common.hpp"
#include <rccl/rccl.h>   // types and prototypes; the functions are resolved at run time (rccl_api)
#include <dlfcn.h>
#include <atomic>
#include <deque>
#include <functional>
#include <future>

namespace {

struct rccl_api {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;     // optional: without it a timed-out communicator is leaked instead of aborted
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
// one binding per process; nullptr (with the reason in *why) when no RCCL can be found
rccl_api *rccl_bind(std::string *why) {
    static std::mutex mu;
    static rccl_api api;
    static bool tried = false;
    static std::string err;
    std::lock_guard<std::mutex> lk(mu);
    if (!tried) {
        tried = true;
        const char *names[] = {getenv("KZG_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            api.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.h) break;
            err = dlerror();
        }
        if (api.h) {
            api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.h, "ncclCommInitAll");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
            api.CommAbort = (decltype(api.CommAbort))dlsym(api.h, "ncclCommAbort");
            api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
            api.GroupStart = (decltype(api.GroupStart))dlsym(api.h, "ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.h, "ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
            if (!api.CommInitAll || !api.CommDestroy || !api.AllGather || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
                err = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
                dlclose(api.h); api.h = nullptr;
            }
        }
    }
    if (!api.h) { if (why) *why = err; return nullptr; }
    return &api;
}

}  // namespace

// a buffer that may cross devices: only exch_arena::take makes one (hipMalloc memory, peer access granted)
struct xbuf { uint8_t *p = nullptr; };
// One block of hipMalloc memory per entry of the handle, bump-allocated per sharded call (calls are serialised by kzg_hip_multi::mu and every
// stream has drained when one returns, so a call starts from an empty arena and may grow it).
struct exch_arena {
    int device = 0; uint8_t *base = nullptr; size_t cap = 0, used = 0;
    int reserve(size_t bytes) {            // nothing of this arena is in flight when this is called
        used = 0;
        if (bytes <= cap) return KZG_HIP_OK;
        HIPCHK(hipSetDevice(device));
        if (base) { HIPCHK(hipFree(base)); base = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 1u << 20);
        HIPCHK(hipMalloc((void **)&base, want));
        cap = want;
        return KZG_HIP_OK;
    }
    int take(size_t bytes, xbuf *out) {
        size_t at = (used + 255) & ~(size_t)255;
        if (at + bytes > cap) { g_last_error = "exchange arena too small (internal sizing error)"; return KZG_HIP_ERR_HIP; }
        out->p = base + at; used = at + bytes;
        return KZG_HIP_OK;
    }
    void release() { if (base) { (void)hipSetDevice(device); (void)hipFree(base); base = nullptr; cap = used = 0; } }
    uint8_t *abandon() { uint8_t *b = base; base = nullptr; cap = used = 0; return b; }   // stuck work may still write it: never reused, freed by the handle if that work ever drains
};
// a host thread bound to one entry of the handle for its lifetime: batch calls hand it that entry's share (no thread creation per call)
struct dev_worker {
    std::thread th; std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; bool stop = false;
    void loop() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); }
            f();
        }
    }
    void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
    void shutdown() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_one(); if (th.joinable()) th.join(); }
};
struct multi_dev {
    int device = 0;
    kzg_hip_fft *fs = nullptr;
    kzg_hip_kzg *ks = nullptr;
    hipStream_t s = nullptr;       // this device's stream for sharded calls
    hipEvent_t ev = nullptr;       // ... and its event for the peer-copy exchange and the cross-stream barrier
    exch_arena arena;              // the memory other devices may read or write
};
enum multi_transport { T_RCCL = 0, T_PEER = 1, T_HOST = 2 };
struct kzg_hip_multi {
    std::vector<multi_dev> d;
    std::vector<std::unique_ptr<dev_worker>> workers;   // entry i > 0 is served by workers[i - 1]; entry 0 by the calling thread
    rccl_api *nccl = nullptr;              // non-null: the exchange is ncclAllGather
    std::vector<ncclComm_t> comms;
    int tkind = T_PEER;
    std::string transport = "peer-copy", transport_note, self_test;
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;   // pinned, portable: the host-staged exchange
    unsigned fault = 0;                    // KZG_HIP_MULTI_FAULT bits (tests)
    long probe_timeout_ms = 10000;         // deadline of one self-test exchange (KZG_HIP_MULTI_PROBE_TIMEOUT_MS)
    struct abandoned_t { int device; hipStream_t s; hipEvent_t ev; uint8_t *arena; };
    std::vector<abandoned_t> abandoned;    // streams / events / arenas of entries whose probe timed out: never synchronised
    uint32_t *h_spin_flag = nullptr;       // pinned, mapped: what the injected "hang" kernels spin on
    int fft_mode = -1;                     // -1 default policy, 0 gather, 1 sharded transforms
    std::mutex mu;                         // sharded calls (collectives) on a handle run one at a time
    std::atomic<uint64_t> n_allgather{0};  // exchanges performed (tests and bench read it)
};
enum { FAULT_RCCL = 1, FAULT_RCCL_CORRUPT = 2, FAULT_PEER = 4, FAULT_PEER_CORRUPT = 8, FAULT_RCCL_HANG = 16, FAULT_PEER_HANG = 32, FAULT_PEER_STUCK = 64, FAULT_RCCL_BLOCK = 128, FAULT_RCCL_INIT_BLOCK = 256 };
struct kzg_hip_multi_eth { kzg_hip_multi *m = nullptr; std::vector<kzg_hip_eth *> eth; uint64_t n = 0; };
struct kzg_hip_multi_fk20s { kzg_hip_multi *m = nullptr; std::vector<kzg_hip_fk20s *> fk; uint64_t n2 = 0; };
struct kzg_hip_multi_fk20m { kzg_hip_multi *m = nullptr; std::vector<kzg_hip_fk20m *> fk; uint64_t n2 = 0, l = 1; };

namespace {

// contiguous share of `total` units for part i of `parts` (the split multi_gpu.py's shard_units makes between ranks)
inline void share(uint64_t total, uint64_t parts, uint64_t i, uint64_t *lo, uint64_t *hi) {
    uint64_t base = total / parts, rem = total % parts;
    *lo = i * base + std::min<uint64_t>(i, rem);
    *hi = *lo + base + (i < rem ? 1 : 0);
}

// runs f(i) for every entry -- entry 0 on the calling thread, entry i > 0 on its worker -- and returns the first non-zero status.  Nothing
// here creates a thread, so nothing here can throw std::system_error past the callers' KZG_CATCH with threads still joinable.
template <class F> int per_device(kzg_hip_multi *m, F f) {
    const size_t D = m->d.size();
    std::vector<int> st(D, KZG_HIP_OK);
    std::vector<std::string> errs(D);
    std::mutex done_mu; std::condition_variable done_cv; size_t pending = 0;
    auto body = [&](size_t i) {
        try { st[i] = f(i); }
        catch (const std::exception &e) { g_last_error = e.what(); st[i] = KZG_HIP_ERR_HIP; }
        catch (...) { g_last_error = "unknown exception"; st[i] = KZG_HIP_ERR_HIP; }
        if (st[i] == KZG_HIP_ERR_HIP) errs[i] = g_last_error;   // g_last_error is thread-local: carry the text to the caller's thread
    };
    for (size_t i = 1; i < D; i++) {
        if (i - 1 >= m->workers.size()) { body(i); continue; }   // (a handle under construction whose workers could not all be started)
        { std::lock_guard<std::mutex> lk(done_mu); pending++; }
        try {
            m->workers[i - 1]->post([&, i] { body(i); std::lock_guard<std::mutex> lk(done_mu); if (--pending == 0) done_cv.notify_one(); });
        } catch (...) {                                          // the queue could not grow: run the share here instead
            { std::lock_guard<std::mutex> lk(done_mu); pending--; }
            body(i);
        }
    }
    body(0);
    { std::unique_lock<std::mutex> lk(done_mu); done_cv.wait(lk, [&] { return pending == 0; }); }   // every posted share has finished, whatever its status
    for (size_t i = 0; i < D; i++)
        if (st[i] != KZG_HIP_OK) { if (st[i] == KZG_HIP_ERR_HIP) g_last_error = errs[i]; return st[i]; }
    return KZG_HIP_OK;
}

// stream-ordered temporaries of one device of a sharded call (device-LOCAL data only: what crosses devices comes from exch_arena); frees on
// that device whatever the calling thread's current device is
struct mtmp {
    int device; hipStream_t s; std::vector<void *> ptrs;
    mtmp(int dev, hipStream_t st) : device(dev), s(st) {}
    mtmp(mtmp &&) = default;
    mtmp(const mtmp &) = delete;
    template <class T> int alloc(T **p, size_t count) {
        *p = nullptr;
        if (!count) count = 1;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipMallocAsync((void **)p, count * sizeof(T), s));
        ptrs.push_back(*p);
        return KZG_HIP_OK;
    }
    ~mtmp() { if (ptrs.empty()) return; (void)hipSetDevice(device); for (void *p : ptrs) (void)hipFreeAsync(p, s); }
};
// whatever way a sharded call returns, every device's stream has drained before its temporaries are released and the handle is unlocked
struct drain_all {
    kzg_hip_multi *m;
    explicit drain_all(kzg_hip_multi *mm) : m(mm) {}
    ~drain_all() { for (auto &d : m->d) { (void)hipSetDevice(d.device); (void)hipStreamSynchronize(d.s); } }
};

// every stream waits for everything enqueued so far on every other stream
int cross_barrier(kzg_hip_multi *m) {
    for (auto &d : m->d) { HIPCHK(hipSetDevice(d.device)); HIPCHK(hipEventRecord(d.ev, d.s)); }
    for (auto &d : m->d) {
        HIPCHK(hipSetDevice(d.device));
        for (auto &o : m->d) if (&o != &d) HIPCHK(hipStreamWaitEvent(d.s, o.ev, 0));
    }
    return KZG_HIP_OK;
}

// In-place all-gather of bytes: device i holds its slice at buf[i].p + i * bytes_each and ends with all D slices in buf[i].p.  The buffers are
// arena memory (xbuf); `kind` is the transport to use (the handle's, or the one transport_self_test is probing).
int all_gather_bytes(kzg_hip_multi *m, const std::vector<xbuf> &buf, size_t bytes_each, int kind) {
    const size_t D = m->d.size();
    m->n_allgather++;
    if (kind == T_RCCL) {
        if (m->fault & FAULT_RCCL) { g_last_error = "ncclAllGather failed: injected fault (KZG_HIP_MULTI_FAULT=rccl)"; return KZG_HIP_ERR_HIP; }
        rccl_api *n = m->nccl;
        ncclResult_t r = n->GroupStart();
        hipError_t he = hipSuccess;
        for (size_t i = 0; i < D && r == ncclSuccess && he == hipSuccess; i++) {   // one thread drives every communicator: the calls form one group
            he = hipSetDevice(m->d[i].device);
            if (he == hipSuccess) r = n->AllGather(buf[i].p + i * bytes_each, buf[i].p, bytes_each, ncclUint8, m->comms[i], m->d[i].s);
        }
        ncclResult_t r2 = n->GroupEnd();   // always closed, also after a failure inside the group
        if (r == ncclSuccess) r = r2;
        HIPCHK(he);
        if (r != ncclSuccess) { g_last_error = std::string("ncclAllGather failed: ") + n->GetErrorString(r); return KZG_HIP_ERR_HIP; }
        if (m->fault & FAULT_RCCL_CORRUPT) { HIPCHK(hipSetDevice(m->d[D - 1].device)); HIPCHK(hipMemsetAsync(buf[D - 1].p, 0x5a, 1, m->d[D - 1].s)); }
        return KZG_HIP_OK;
    }
    if (kind == T_HOST) {
        // device -> pinned host -> device: no peer mapping involved.  The staging area belongs to the handle (sharded calls are serialised).
        const size_t total = D * bytes_each;
        if (m->h_stage_cap < total) {
            if (m->h_stage) {   // copies of an earlier exchange may still be reading the old area: drain before it goes
                for (auto &d : m->d) { HIPCHK(hipSetDevice(d.device)); HIPCHK(hipStreamSynchronize(d.s)); }
                HIPCHK(hipHostFree(m->h_stage)); m->h_stage = nullptr; m->h_stage_cap = 0;
            }
            size_t cap = std::max<size_t>(total, 1u << 20);
            HIPCHK(hipHostMalloc((void **)&m->h_stage, cap, hipHostMallocPortable));
            m->h_stage_cap = cap;
        }
        for (size_t i = 0; i < D; i++) {
            HIPCHK(hipSetDevice(m->d[i].device));
            HIPCHK(hipMemcpyAsync(m->h_stage + i * bytes_each, buf[i].p + i * bytes_each, bytes_each, hipMemcpyDeviceToHost, m->d[i].s));
            HIPCHK(hipEventRecord(m->d[i].ev, m->d[i].s));
        }
        for (size_t j = 0; j < D; j++) {
            HIPCHK(hipSetDevice(m->d[j].device));
            for (size_t i = 0; i < D; i++) {
                if (i == j) continue;
                HIPCHK(hipStreamWaitEvent(m->d[j].s, m->d[i].ev, 0));
                HIPCHK(hipMemcpyAsync(buf[j].p + i * bytes_each, m->h_stage + i * bytes_each, bytes_each, hipMemcpyHostToDevice, m->d[j].s));
            }
        }
        return cross_barrier(m);   // the staging area is reused by the next exchange only after every reader is done
    }
    // peer copies: device j pulls slice i from device i once i's stream has produced it
    if (m->fault & FAULT_PEER) { g_last_error = "peer copy failed: injected fault (KZG_HIP_MULTI_FAULT=peer)"; return KZG_HIP_ERR_HIP; }
    for (auto &d : m->d) { HIPCHK(hipSetDevice(d.device)); HIPCHK(hipEventRecord(d.ev, d.s)); }
    for (size_t j = 0; j < D; j++) {
        HIPCHK(hipSetDevice(m->d[j].device));
        for (size_t i = 0; i < D; i++) {
            if (i == j) continue;
            HIPCHK(hipStreamWaitEvent(m->d[j].s, m->d[i].ev, 0));
            if (m->d[i].device == m->d[j].device) HIPCHK(hipMemcpyAsync(buf[j].p + i * bytes_each, buf[i].p + i * bytes_each, bytes_each, hipMemcpyDeviceToDevice, m->d[j].s));
            else HIPCHK(hipMemcpyPeerAsync(buf[j].p + i * bytes_each, m->d[j].device, buf[i].p + i * bytes_each, m->d[i].device, bytes_each, m->d[j].s));
        }
    }
    if (m->fault & FAULT_PEER_CORRUPT) { HIPCHK(hipSetDevice(m->d[D - 1].device)); HIPCHK(hipMemsetAsync(buf[D - 1].p, 0x5a, 1, m->d[D - 1].s)); }
    return cross_barrier(m);   // a source buffer may be reused only after every reader is done
}
inline int all_gather_bytes(kzg_hip_multi *m, const std::vector<xbuf> &buf, size_t bytes_each) { return all_gather_bytes(m, buf, bytes_each, m->tkind); }

// fault injection: holds a stream until the host sets *flag -- or until its own clock runs out, so that a test can never wedge the device
__global__ void k_multi_spin(uint32_t *flag, uint64_t max_ticks) {
    const uint64_t t0 = wall_clock64();                        // constant 100 MHz counter
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(127);
}
int enqueue_injected_hang(kzg_hip_multi *m, size_t entry) {
    if (!m->h_spin_flag) {
        HIPCHK(hipHostMalloc((void **)&m->h_spin_flag, sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped));
        *m->h_spin_flag = 0;
    }
    __atomic_store_n(m->h_spin_flag, 0u, __ATOMIC_SEQ_CST);
    uint32_t *dflag = nullptr;
    HIPCHK(hipSetDevice(m->d[entry].device));
    HIPCHK(hipHostGetDevicePointer((void **)&dflag, m->h_spin_flag, 0));
    const uint64_t ticks = (uint64_t)(2 * m->probe_timeout_ms + 500) * 100000ull;          // 100 MHz: 1e5 ticks per millisecond
    k_multi_spin<<<1, 1, 0, m->d[entry].s>>>(dflag, ticks);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
inline void release_injected_hang(kzg_hip_multi *m) { if (m->h_spin_flag) __atomic_store_n(m->h_spin_flag, 1u, __ATOMIC_SEQ_CST); }

// Polls every entry's stream until all have drained: 0 = drained, 1 = the deadline passed (some stream still has work), -1 = a stream reports an error
// (g_last_error set).  Never blocks in the runtime: hipStreamQuery returns at once.
int wait_streams_deadline(kzg_hip_multi *m, long timeout_ms) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (long spins = 0;; spins++) {
        bool all = true;
        for (auto &d : m->d) {
            (void)hipSetDevice(d.device);
            hipError_t e = hipStreamQuery(d.s);
            if (e == hipErrorNotReady) { (void)hipGetLastError(); all = false; continue; }
            if (e != hipSuccess) { (void)hipGetLastError(); g_last_error = std::string("stream of device ") + std::to_string(d.device) + ": " + hipGetErrorString(e); return -1; }
        }
        if (all) return 0;
        if (std::chrono::steady_clock::now() >= until) return 1;
        if (spins < 200) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}
// The entries' streams hold work that will not finish: give every entry a fresh stream, event and (empty) arena; the old ones go to `abandoned`
// (nothing here waits for them).  False when a replacement could not be created (g_last_error set).
bool abandon_streams(kzg_hip_multi *m) {
    for (auto &d : m->d) {
        m->abandoned.push_back({d.device, d.s, d.ev, d.arena.abandon()});
        d.s = nullptr; d.ev = nullptr;
        if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) {
            g_last_error = std::string("could not replace the stream of device ") + std::to_string(d.device) + ": " + hipGetErrorString(hipGetLastError());
            return false;
        }
    }
    return true;
}
// ncclCommAbort on every communicator, on a helper thread that gets the same deadline (an abort that blocks -- it synchronises RCCL's internal streams -- must
// not wedge the constructor either); the communicators are gone afterwards whatever happened
std::string abort_communicators(kzg_hip_multi *m) {
    std::string how;
    rccl_api *api = m->nccl;
    std::vector<ncclComm_t> comms; comms.swap(m->comms);
    m->nccl = nullptr;
    if (!api || comms.empty()) return "no communicator to abort";
    if (!api->CommAbort) return "librccl has no ncclCommAbort: communicators leaked";
    auto done = std::make_shared<std::promise<int>>();
    std::future<int> fut = done->get_future();
    try {
        std::thread([api, comms, done] { int bad = 0; for (ncclComm_t c : comms) if (c && api->CommAbort(c) != ncclSuccess) bad++; done->set_value(bad); }).detach();
    } catch (const std::exception &) { return "could not start the abort thread: communicators leaked"; }
    if (fut.wait_for(std::chrono::milliseconds(m->probe_timeout_ms)) != std::future_status::ready) return "ncclCommAbort did not return within the deadline either (left behind on its thread)";
    const int bad = fut.get();
    return bad ? "ncclCommAbort reported an error on " + std::to_string(bad) + " communicator(s)" : "communicators aborted";
}

// The host side of RCCL can block too (ncclGroupEnd waiting for a proxy connection that never comes up; ncclCommInitAll on a fabric that does not answer).  The probe's
// ncclAllGather group and the communicators' creation therefore run on a helper thread that owns COPIES of everything it touches (never the handle), and the constructor
// waits for it against a deadline.  A helper that is still inside RCCL when the deadline passes is left behind (detached; ncclCommAbort from the constructor's thread is
// NCCL's documented way to unblock it); what it references -- the streams it enqueues on -- is abandoned, not destroyed.
struct rccl_probe_job {
    rccl_api *api; std::vector<ncclComm_t> comms; std::vector<int> devices; std::vector<hipStream_t> streams; std::vector<uint8_t *> bufs; size_t each; long block_ms;
    std::atomic<bool> cancelled{false};                            // set by the constructor when it stops waiting: an injected block then returns without touching RCCL
    std::promise<std::pair<int, std::string>> done;
};
void rccl_probe_enqueue(std::shared_ptr<rccl_probe_job> j) {
    if (j->block_ms > 0) {                                                                           // injected: "rccl-block"
        std::this_thread::sleep_for(std::chrono::milliseconds(j->block_ms));
        if (j->cancelled.load()) { j->done.set_value({KZG_HIP_ERR_HIP, "cancelled"}); return; }       // (the communicators it holds have been aborted meanwhile)
    }
    ncclResult_t r = j->api->GroupStart();
    hipError_t he = hipSuccess;
    for (size_t i = 0; i < j->comms.size() && r == ncclSuccess && he == hipSuccess; i++) {
        he = hipSetDevice(j->devices[i]);
        if (he == hipSuccess) r = j->api->AllGather(j->bufs[i] + i * j->each, j->bufs[i], j->each, ncclUint8, j->comms[i], j->streams[i]);
    }
    const ncclResult_t r2 = j->api->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (he != hipSuccess) { j->done.set_value({KZG_HIP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he)}); return; }
    if (r != ncclSuccess) { j->done.set_value({KZG_HIP_ERR_HIP, std::string("ncclAllGather failed: ") + j->api->GetErrorString(r)}); return; }
    j->done.set_value({KZG_HIP_OK, std::string()});
}

// One exchange with known contents on transport `kind`: entry i fills its slice with bytes that depend on (i, offset), one all_gather_bytes,
// every entry's whole buffer is read back and compared on the host.  KZG_HIP_OK and *why empty when every byte arrived everywhere.  *timed_out: the
// exchange did not complete within the deadline -- the entries' streams still hold it and must not be waited for (the caller abandons them).
int transport_probe(kzg_hip_multi *m, int kind, std::string *why, bool *timed_out) {
    const size_t D = m->d.size(), each = 4096 + 144;            // not a power of two: a slice boundary inside a cache line
    std::vector<xbuf> buf(D);
    std::vector<uint8_t> pat(D * each), got(D * each);
    *timed_out = false;
    for (size_t i = 0; i < D; i++) for (size_t b = 0; b < each; b++) pat[i * each + b] = (uint8_t)(0x31 * (i + 1) + 7 * b + (b >> 8));
    // every way out drains the streams -- against the deadline, never with a blocking synchronise
    auto fail = [&](int st) {
        *why = g_last_error;
        if (wait_streams_deadline(m, m->probe_timeout_ms) == 1) { *timed_out = true; *why += " [and the streams did not drain within the deadline]"; }
        (void)hipGetLastError();
        return st;
    };
    for (size_t i = 0; i < D; i++) {
        multi_dev &d = m->d[i];
        int st = d.arena.reserve(D * each + 256); if (st) return fail(st);
        st = d.arena.take(D * each, &buf[i]); if (st) return fail(st);
        if (hipSetDevice(d.device) != hipSuccess || hipMemsetAsync(buf[i].p, 0, D * each, d.s) != hipSuccess ||
            hipMemcpyAsync(buf[i].p + i * each, pat.data() + i * each, each, hipMemcpyHostToDevice, d.s) != hipSuccess) { g_last_error = "self-test: could not fill the pattern"; return fail(KZG_HIP_ERR_HIP); }
    }
    // (the pattern upload from pageable memory has completed on return of hipMemcpyAsync or is stream-ordered: either way the streams are idle or short here)
    if ((kind == T_RCCL && (m->fault & FAULT_RCCL_HANG)) || (kind == T_PEER && (m->fault & FAULT_PEER_HANG))) {
        int st = enqueue_injected_hang(m, D - 1); if (st) return fail(st);
    }
    int st;
    if (kind == T_RCCL && !(m->fault & (FAULT_RCCL | FAULT_RCCL_CORRUPT))) {
        // the enqueue itself against the deadline (rccl_probe_enqueue, above); the sharded calls later use all_gather_bytes directly: by then the transport is proven
        auto job = std::make_shared<rccl_probe_job>();
        job->api = m->nccl; job->comms = m->comms; job->each = each; job->block_ms = (m->fault & FAULT_RCCL_BLOCK) ? 3 * m->probe_timeout_ms : 0;
        for (size_t i = 0; i < D; i++) { job->devices.push_back(m->d[i].device); job->streams.push_back(m->d[i].s); job->bufs.push_back(buf[i].p); }
        std::future<std::pair<int, std::string>> fut = job->done.get_future();
        m->n_allgather++;
        try { std::thread(rccl_probe_enqueue, job).detach(); }
        catch (const std::exception &e) { g_last_error = std::string("self-test: could not start the RCCL helper thread: ") + e.what(); return fail(KZG_HIP_ERR_HIP); }
        if (fut.wait_for(std::chrono::milliseconds(m->probe_timeout_ms)) != std::future_status::ready) {
            char t[220]; snprintf(t, sizeof t, "self-test: the RCCL calls did not return within %ld ms (KZG_HIP_MULTI_PROBE_TIMEOUT_MS): timeout on the host side", m->probe_timeout_ms);
            *why = t; g_last_error = t; *timed_out = true;
            job->cancelled.store(true);
            return KZG_HIP_ERR_HIP;
        }
        const std::pair<int, std::string> res = fut.get();
        st = res.first;
        if (st) g_last_error = res.second;
    } else st = all_gather_bytes(m, buf, each, kind);
    if (st) return fail(st);
    const int w = wait_streams_deadline(m, m->probe_timeout_ms);
    if (w == 1) {
        char t[200]; snprintf(t, sizeof t, "self-test: the all-gather did not complete within %ld ms (KZG_HIP_MULTI_PROBE_TIMEOUT_MS): timeout", m->probe_timeout_ms);
        *why = t; g_last_error = t; *timed_out = true;
        return KZG_HIP_ERR_HIP;
    }
    if (w < 0) return fail(KZG_HIP_ERR_HIP);
    for (size_t i = 0; i < D; i++) {      // the streams are idle: these copies cannot queue behind anything
        multi_dev &d = m->d[i];
        if (hipSetDevice(d.device) != hipSuccess || hipMemcpyAsync(got.data(), buf[i].p, D * each, hipMemcpyDeviceToHost, d.s) != hipSuccess ||
            hipStreamSynchronize(d.s) != hipSuccess) { g_last_error = std::string("self-test: read-back failed: ") + hipGetErrorString(hipGetLastError()); return fail(KZG_HIP_ERR_HIP); }
        if (got != pat) {
            size_t at = 0; while (got[at] == pat[at]) at++;
            char t[160]; snprintf(t, sizeof t, "self-test: entry %zu holds wrong bytes after the all-gather (first at slice %zu, offset %zu)", i, at / each, at % each);
            g_last_error = t;
            return fail(KZG_HIP_ERR_HIP);
        }
    }
    why->clear();
    return KZG_HIP_OK;
}

const char *transport_name(int k) { return k == T_RCCL ? "rccl" : k == T_PEER ? "peer-copy" : "host-staged"; }

// Creation-time proof of the exchange: probes the preferred transport and steps down (rccl -> peer-copy -> host-staged) until one delivers.
int transport_self_test(kzg_hip_multi *m) {
    std::lock_guard<std::mutex> lk(m->mu);
    for (int kind = m->tkind; kind <= T_HOST; kind++) {
        if (kind == T_RCCL && !m->nccl) continue;
        std::string why;
        bool timed_out = false;
        int st = transport_probe(m, kind, &why, &timed_out);
        if (st == KZG_HIP_OK) {
            m->tkind = kind; m->transport = transport_name(kind);
            char t[160]; snprintf(t, sizeof t, "ok: %s, %zu entries, %d B per slice, every byte verified on every entry", transport_name(kind), m->d.size(), 4096 + 144);
            m->self_test = t;
            return KZG_HIP_OK;
        }
        if (!m->transport_note.empty()) m->transport_note += "; ";
        m->transport_note += std::string(transport_name(kind)) + " failed its self-test (" + why + ")";
        if (timed_out) {
            // injected hang: what ncclCommAbort does to RCCL's kernels (they poll the abort flag) the test does to its own spinning kernel -- BEFORE the abort, which
            // synchronises internal streams that wait for the user stream; a peer copy cannot be cancelled at all: its stream is simply left behind
            if (kind == T_RCCL && (m->fault & FAULT_RCCL_HANG)) release_injected_hang(m);
            if (kind == T_RCCL) m->transport_note += "; " + abort_communicators(m);
            bool drained = kind == T_RCCL && wait_streams_deadline(m, m->probe_timeout_ms) == 0;      // aborted collectives return: the streams are usable again
            if (!drained) {
                if (!abandon_streams(m)) { m->self_test = "failed: a transport hung and its streams could not be replaced"; g_last_error = "multi-device exchange: " + m->transport_note + "; " + g_last_error; return KZG_HIP_ERR_HIP; }
                m->transport_note += "; streams and arenas of the hung exchange abandoned";
                // injected peer hang: the stuck exchange is let go only NOW, so that it completes late, next to the probe of the next transport, and writes into the
                // abandoned arenas -- the case the abandoning exists for.  (Left spinning it would also hold whatever hardware queue its stream shares with the fresh
                // ones -- ROCm maps a process's streams onto a few queues -- which is a property of the injection, not of the code under test.)
                if (kind == T_PEER && (m->fault & FAULT_PEER_HANG) && !(m->fault & FAULT_PEER_STUCK)) release_injected_hang(m);
            }
        } else if (kind == T_RCCL) {   // a communicator that failed once is not used again
            for (ncclComm_t c : m->comms) if (c) (void)m->nccl->CommDestroy(c);
            m->comms.clear(); m->nccl = nullptr;
        }
    }
    m->self_test = "failed: no transport delivered";
    g_last_error = "multi-device exchange: " + m->transport_note;
    return KZG_HIP_ERR_HIP;
}

// peer access between every distinct pair of listed devices (both directions); what could not be granted goes into the note and the
// exchange then runs host-staged at worst (the self-test decides)
void enable_peer_access(kzg_hip_multi *m) {
    std::vector<int> devs;
    for (auto &d : m->d) if (std::find(devs.begin(), devs.end(), d.device) == devs.end()) devs.push_back(d.device);
    for (int a : devs) for (int b : devs) {
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        hipError_t e = hipErrorPeerAccessUnsupported;
        if (can && hipSetDevice(a) == hipSuccess) e = hipDeviceEnablePeerAccess(b, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (!m->transport_note.empty()) m->transport_note += "; ";
            m->transport_note += "no peer access " + std::to_string(a) + " -> " + std::to_string(b) + " (" + hipGetErrorString(e) + ")";
        }
    }
}

// x[a] = src[D a + r], a < count: the residue class r of a sequence (decimation in time by D)
__global__ void k_multi_gather_stride(const g1j *src, uint64_t D, uint64_t r, uint64_t count, g1j *x) {
    uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (a < count) x[a] = src[D * a + r];
}
// operands of the radix-D combine of a transform of size N = D M from its D sub-transforms Y_r (M points each):
//   out[t] = sum_r w_N^(r t) Y_r[t mod M]   for t in [t0, t0 + cnt)
// row r of pts / sc (cnt entries each) holds Y_r[t mod M] and w^(r t); roots = Expanded / ReverseRootsOfUnity with stride W / N
__global__ void k_multi_combine_operands(const g1j *Y, const fr *roots, uint64_t root_stride, uint64_t N, uint64_t M, uint64_t D, uint64_t t0, uint64_t cnt, g1j *pts, fr *sc) {
    uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (e >= D * cnt) return;
    uint64_t r = e / cnt, t = t0 + e % cnt;
    pts[e] = Y[r * M + (t & (M - 1))];
    sc[e] = roots[((r * t) & (N - 1)) * root_stride];
}

// One G1 transform of size N (inv: with ReverseRootsOfUnity, unscaled) of the sequence x (replicated on every device; only x[:n_valid] is
// non-trivial) sharded by decimation in time over the D devices: device r transforms the residue class r (size M = N / D), the sub-results
// are all-gathered, device r combines the output slice [r cnt, (r + 1) cnt) of the first n_out = D cnt outputs.  out[r] receives that slice
// at out[r] + r * cnt (the caller all-gathers or reads it).  Two launches of scalar multiplications per device: M-point transform, D - 1
// products per output.
int sharded_g1_fft(kzg_hip_multi *m, std::vector<mtmp> &tmp, const std::vector<const g1j *> &x, uint64_t n_valid, uint64_t N, int inv, uint64_t n_out,
                   const std::vector<g1j *> &out) {
    const uint64_t D = m->d.size(), M = N / D, cnt = n_out / D;
    std::vector<xbuf> ybuf(D);
    for (uint64_t r = 0; r < D; r++) {
        multi_dev &d = m->d[r];
        g1j *xr = nullptr;
        const uint64_t valid_r = n_valid > r ? (n_valid - r + D - 1) / D : 0;   // elements D a + r below n_valid
        CHK(tmp[r].alloc(&xr, std::max<uint64_t>(valid_r, 1)));
        CHK(d.arena.take(N * sizeof(g1j), &ybuf[r]));                           // exchanged: arena memory
        g1j *Y = (g1j *)ybuf[r].p;
        HIPCHK(hipSetDevice(d.device));
        if (valid_r) hipLaunchKernelGGL(k_multi_gather_stride, dim3((uint32_t)((valid_r + 255) / 256)), dim3(256), 0, d.s, x[r], D, r, valid_r, xr);
        CHK(g1_fft_rows(d.fs, d.s, xr, valid_r, valid_r, Y + r * M, M, 1, inv));
        HIPCHK(hipGetLastError());
    }
    CHK(all_gather_bytes(m, ybuf, M * sizeof(g1j)));
    for (uint64_t r = 0; r < D; r++) {
        multi_dev &d = m->d[r];
        g1j *pts = nullptr, *prod = nullptr; fr *sc = nullptr;
        CHK(tmp[r].alloc(&pts, D * cnt)); CHK(tmp[r].alloc(&prod, D * cnt)); CHK(tmp[r].alloc(&sc, D * cnt));
        HIPCHK(hipSetDevice(d.device));
        const fr *roots = inv ? d.fs->d_reversed : d.fs->d_expanded;
        hipLaunchKernelGGL(k_multi_combine_operands, dim3((uint32_t)((D * cnt + 255) / 256)), dim3(256), 0, d.s, (const g1j *)ybuf[r].p, roots, d.fs->W / N, N, M, D, r * cnt, cnt, pts, sc);
        HIPCHK(hipMemcpyAsync(prod, pts, cnt * sizeof(g1j), hipMemcpyDeviceToDevice, d.s));           // the term of class 0 has twiddle one
        launch_g1_mul_vec(d.s, pts + cnt, (D - 1) * cnt, sc + cnt, 1, (D - 1) * cnt, prod + cnt);
        launch_g1_sum_files(d.s, prod, D, cnt, 1, out[r] + r * cnt);
        HIPCHK(hipGetLastError());
    }
    return KZG_HIP_OK;
}

bool sharded_fft_default(const kzg_hip_multi *m) {
    if (const char *e = getenv("KZG_HIP_MULTI_FFT")) return !strcmp(e, "sharded");
    if (m->fft_mode >= 0) return m->fft_mode == 1;
    return m->d.size() >= 4;   // two devices halve one of three launches of a lone transform and pay two exchanges for it
}

// DAUsingFK20 / DAUsingFK20Multi of ONE polynomial over all devices of the handle (fk20_single.go:176-196, fk20_multi.go:113-133)
int fk20_da_sharded(kzg_hip_multi *m, const std::vector<fk20_core *> &core, const void *poly_fr, uint64_t n, void *out_g1) {
    KZG_TRY
    std::lock_guard<std::mutex> lk(m->mu);
    const uint64_t D = m->d.size(), k = core[0]->k, k2 = 2 * k;
    const uint64_t cnt = (k2 + D - 1) / D;                       // output positions per device (the last share may be short)
    std::vector<mtmp> tmp; tmp.reserve(D);
    for (auto &d : m->d) tmp.emplace_back(d.device, d.s);
    drain_all drain(m);                                          // declared after tmp: streams drain before the temporaries go
    // everything that crosses devices: hExtFFT (D cnt), two sub-transform files (2k each), h (k), the normalised proofs (2k) -- + alignment slack
    const size_t arena_need = (D * cnt + 2 * k2 + k + k2) * sizeof(g1j) + 8 * 256;
    for (auto &d : m->d) CHK(d.arena.reserve(arena_need));       // the previous sharded call drained every stream: the arena is idle
    std::vector<xbuf> hext(D);
    // (1) Toeplitz stage, sharded by output position; every device needs the coefficients (n x 32 B)
    for (uint64_t i = 0; i < D; i++) {
        multi_dev &d = m->d[i];
        fr *d_poly = nullptr;
        CHK(tmp[i].alloc(&d_poly, n));
        CHK(d.arena.take(D * cnt * sizeof(g1j), &hext[i]));
        g1j *d_hext = (g1j *)hext[i].p;
        HIPCHK(hipSetDevice(d.device));
        HIPCHK(hipMemcpyAsync(d_poly, poly_fr, n * sizeof(fr), hipMemcpyHostToDevice, d.s));
        const uint64_t j0 = std::min(i * cnt, k2), j1 = std::min(j0 + cnt, k2);
        if (j1 > j0) CHK(fk20_hext(core[i], d.s, d_poly, n, n, 1, j0, j1 - j0, d_hext + j0));
    }
    CHK(all_gather_bytes(m, hext, cnt * sizeof(g1j)));           // all-gather #1: hExtFFT
    const bool pow2_devs = (D & (D - 1)) == 0;
    g1j *d_res = nullptr;                                        // the 2k proofs, reverse-bit order, Kilic images, on device 0
    if (D > 1 && pow2_devs && k2 >= 8 * D && sharded_fft_default(m)) {
        // (2) h = IFFT_G1(hExtFFT)[:k] (1 / 2k is folded into the Toeplitz coefficients): sub-transforms, all-gather #2, combine, all-gather #3
        std::vector<const g1j *> x(D); std::vector<g1j *> hbuf(D); std::vector<xbuf> hb(D);
        for (uint64_t i = 0; i < D; i++) { x[i] = (const g1j *)hext[i].p; CHK(m->d[i].arena.take(k * sizeof(g1j), &hb[i])); hbuf[i] = (g1j *)hb[i].p; }
        CHK(sharded_g1_fft(m, tmp, x, k2, k2, 1, k, hbuf));
        CHK(all_gather_bytes(m, hb, (k / D) * sizeof(g1j)));
        // (3) proofs = FFT_G1(h || inf^k): sub-transforms on the k / D valid entries of each class, all-gather #4, combine,
        //     normalise the own slice, all-gather #5 of the proof points
        std::vector<g1j *> pbuf(D), nbuf(D); std::vector<xbuf> nb(D);
        for (uint64_t i = 0; i < D; i++) { x[i] = hbuf[i]; CHK(tmp[i].alloc(&pbuf[i], k2)); CHK(m->d[i].arena.take(k2 * sizeof(g1j), &nb[i])); nbuf[i] = (g1j *)nb[i].p; }
        CHK(sharded_g1_fft(m, tmp, x, k, k2, 0, k2, pbuf));
        const uint64_t M = k2 / D;
        for (uint64_t i = 0; i < D; i++) {
            HIPCHK(hipSetDevice(m->d[i].device));
            launch_g1_normalize(m->d[i].s, pbuf[i] + i * M, nbuf[i] + i * M, M, true);
            HIPCHK(hipGetLastError());
        }
        CHK(all_gather_bytes(m, nb, M * sizeof(g1j)));
        CHK(tmp[0].alloc(&d_res, k2));
        HIPCHK(hipSetDevice(m->d[0].device));
        launch_g1_bitrev_copy(m->d[0].s, nbuf[0], k2, k2, d_res, k2, 1);   // reverseBitOrderG1 (fk20_single.go:192-194)
        HIPCHK(hipGetLastError());
    } else {
        CHK(tmp[0].alloc(&d_res, k2));
        HIPCHK(hipSetDevice(m->d[0].device));
        CHK(fk20_finish(core[0], m->d[0].s, (const g1j *)hext[0].p, 1, 1, 1, d_res));
    }
    HIPCHK(hipSetDevice(m->d[0].device));
    HIPCHK(hipMemcpyAsync(out_g1, d_res, k2 * sizeof(g1j), hipMemcpyDeviceToHost, m->d[0].s));
    HIPCHK(hipStreamSynchronize(m->d[0].s));
    return KZG_HIP_OK;
    KZG_CATCH
}

}  // namespace

extern "C" {

int kzg_hip_multi_settings_new(const int *devices, uint32_t n_devices, unsigned max_scale, const void *secret_g1, uint64_t n_setup, kzg_hip_multi **out) {
    if (!out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (!devices || !n_devices || n_devices > 64 || !secret_g1) return KZG_HIP_ERR_BAD_ARG;
    if (kzg_hip_device_count() < 1) return KZG_HIP_ERR_NO_DEVICE;
    // ordinals are bounded by what the runtime enumerates (a mixed node may have a gfx950 device at ordinal 1 behind another architecture at
    // ordinal 0); kzg_hip_fft_settings_new rejects a device that is not gfx950
    int ordinals = 0;
    if (hipGetDeviceCount(&ordinals) != hipSuccess) { (void)hipGetLastError(); return KZG_HIP_ERR_NO_DEVICE; }
    for (uint32_t i = 0; i < n_devices; i++) if (devices[i] < 0 || devices[i] >= ordinals) return KZG_HIP_ERR_NO_DEVICE;
    KZG_TRY
    // owned until it is handed out: whatever throws below (std::bad_alloc in a resize, a string concatenation, the probe's vectors) frees the handle with its worker
    // threads, streams, arenas and per-device settings on the way to KZG_CATCH
    std::unique_ptr<kzg_hip_multi, void (*)(kzg_hip_multi *)> owner(new kzg_hip_multi, kzg_hip_multi_settings_free);
    kzg_hip_multi *m = owner.get();
    m->d.resize(n_devices);
    if (const char *t = getenv("KZG_HIP_MULTI_PROBE_TIMEOUT_MS")) { long v = atol(t); if (v >= 10 && v <= 600000) m->probe_timeout_ms = v; }
    for (uint32_t i = 0; i < n_devices; i++) { m->d[i].device = devices[i]; m->d[i].arena.device = devices[i]; }
    if (const char *f = getenv("KZG_HIP_MULTI_FAULT")) {   // tests: make a leg of the exchange fail so that the fall-backs run on one GPU
        std::string fs(f);
        auto has = [&](const char *w) { size_t at = 0; const size_t n = strlen(w); while ((at = fs.find(w, at)) != std::string::npos) { const size_t e = at + n; if ((at == 0 || fs[at - 1] == ',') && (e == fs.size() || fs[e] == ',')) return true; at = e; } return false; };
        if (has("rccl")) m->fault |= FAULT_RCCL;
        if (has("rccl-corrupt")) m->fault |= FAULT_RCCL_CORRUPT;
        if (has("peer")) m->fault |= FAULT_PEER;
        if (has("peer-corrupt")) m->fault |= FAULT_PEER_CORRUPT;
        if (has("rccl-hang")) m->fault |= FAULT_RCCL_HANG;
        if (has("rccl-block")) m->fault |= FAULT_RCCL_BLOCK;            // the RCCL calls of the probe block on the HOST side (the helper thread sleeps three deadlines)
        if (has("rccl-init-block")) m->fault |= FAULT_RCCL_INIT_BLOCK;  // ncclCommInitAll does not return in time
        if (has("peer-hang")) m->fault |= FAULT_PEER_HANG;
        if (has("peer-stuck")) m->fault |= FAULT_PEER_HANG | FAULT_PEER_STUCK;   // ... and is NOT let go after its streams were abandoned: stuck until the kernel's own clock runs out
    }
    try {   // one host thread per further entry, for the lifetime of the handle; a thread that cannot be started fails the constructor cleanly
        for (uint32_t i = 1; i < n_devices; i++) {
            m->workers.emplace_back(new dev_worker);
            dev_worker *w = m->workers.back().get();
            w->th = std::thread([w] { w->loop(); });
        }
    } catch (const std::exception &e) {
        g_last_error = std::string("multi-device handle: worker thread: ") + e.what();
        return KZG_HIP_ERR_HIP;
    }
    int st = per_device(m, [&](size_t i) -> int {
        multi_dev &d = m->d[i];
        CHK(kzg_hip_fft_settings_new(d.device, max_scale, &d.fs));
        CHK(kzg_hip_kzg_settings_new(d.fs, secret_g1, n_setup, &d.ks));
        HIPCHK(hipSetDevice(d.device));
        HIPCHK(hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
        return KZG_HIP_OK;
    });
    if (st) { std::string keep = g_last_error; owner.reset(); g_last_error = keep; return st; }
    // exchange transport: RCCL when every device is listed once (KZG_HIP_MULTI_TRANSPORT=rccl also for a single device: the binding's own test;
    // =peer never binds RCCL, =host goes straight to the host-staged exchange)
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; i++) for (uint32_t j = 0; j < i; j++) if (devices[i] == devices[j]) distinct = false;
    enable_peer_access(m);
    const char *force = getenv("KZG_HIP_MULTI_TRANSPORT");
    const bool want_rccl = force ? !strcmp(force, "rccl") && distinct : (distinct && n_devices >= 2);
    if (force && !strcmp(force, "host")) m->tkind = T_HOST;
    if (want_rccl) {
        std::string why;
        rccl_api *api = rccl_bind(&why);
        if (!api) m->transport_note = "RCCL not bound (" + why + ")";
        else {
            // ncclCommInitAll on a helper thread, against 6 x the probe deadline (creation legitimately takes seconds on 8 devices); a helper that does not come back is
            // left behind with its own copies, and the handle goes on with peer copies
            struct init_job { rccl_api *api; std::vector<int> devs; std::vector<ncclComm_t> comms; long block_ms; std::promise<ncclResult_t> done; };
            auto job = std::make_shared<init_job>();
            job->api = api; job->devs.assign(devices, devices + n_devices); job->comms.assign(n_devices, nullptr);
            job->block_ms = (m->fault & FAULT_RCCL_INIT_BLOCK) ? 8 * m->probe_timeout_ms : 0;
            std::future<ncclResult_t> fut = job->done.get_future();
            std::thread([job] {
                if (job->block_ms > 0) { std::this_thread::sleep_for(std::chrono::milliseconds(job->block_ms)); job->done.set_value(ncclSystemError); return; }
                job->done.set_value(job->api->CommInitAll(job->comms.data(), (int)job->devs.size(), job->devs.data()));
            }).detach();
            if (fut.wait_for(std::chrono::milliseconds(6 * m->probe_timeout_ms)) != std::future_status::ready) {
                char t[200]; snprintf(t, sizeof t, "ncclCommInitAll did not return within %ld ms: timeout (left behind on its thread)", 6 * m->probe_timeout_ms);
                m->transport_note = t;
            } else {
                const ncclResult_t r = fut.get();
                if (r != ncclSuccess) m->transport_note = std::string("ncclCommInitAll: ") + api->GetErrorString(r);
                else { m->comms = job->comms; m->nccl = api; m->tkind = T_RCCL; }
            }
        }
    } else if (!distinct) m->transport_note = "the device list repeats a device";
    m->transport = transport_name(m->tkind);
    // the handle proves its exchange before anyone relies on it: pattern -> all-gather -> verify on every entry; steps down on failure
    st = transport_self_test(m);
    m->n_allgather = 0;                      // kzg_hip_multi_exchanges counts the callers' exchanges
    if (st) { std::string keep = g_last_error; owner.reset(); g_last_error = keep; return st; }
    *out = owner.release();
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_multi_settings_free(kzg_hip_multi *m) {
    if (!m) return;
    release_injected_hang(m);
    for (auto &w : m->workers) w->shutdown();
    if (m->nccl) for (ncclComm_t c : m->comms) if (c) (void)m->nccl->CommDestroy(c);
    for (auto &d : m->d) {
        (void)hipSetDevice(d.device);
        if (d.s) { (void)hipStreamSynchronize(d.s); (void)hipStreamDestroy(d.s); }
        if (d.ev) (void)hipEventDestroy(d.ev);
        d.arena.release();
        if (d.ks) kzg_hip_kzg_settings_free(d.ks);
        if (d.fs) kzg_hip_fft_settings_free(d.fs);
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    // what a hung exchange left behind is freed only if ALL of it has drained by now (hipFree synchronises the device: with one stream still stuck it would block);
    // otherwise it is leaked -- never waited for.  (An injected spin was released at the top and ends within microseconds.)
    bool drained = true;
    if (!m->abandoned.empty()) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(m->h_spin_flag ? 2000 : 0);
        for (auto &a : m->abandoned) {
            (void)hipSetDevice(a.device);
            hipError_t e = hipStreamQuery(a.s);
            while (e == hipErrorNotReady && std::chrono::steady_clock::now() < until) { std::this_thread::sleep_for(std::chrono::microseconds(200)); e = hipStreamQuery(a.s); }
            if (e != hipSuccess) drained = false;
            (void)hipGetLastError();
        }
        if (drained)
            for (auto &a : m->abandoned) { (void)hipSetDevice(a.device); (void)hipStreamDestroy(a.s); if (a.ev) (void)hipEventDestroy(a.ev); if (a.arena) (void)hipFree(a.arena); }
    }
    if (m->h_spin_flag && drained) (void)hipHostFree(m->h_spin_flag);
    (void)hipGetLastError();
    delete m;
}
uint32_t kzg_hip_multi_device_count(const kzg_hip_multi *m) { return m ? (uint32_t)m->d.size() : 0; }
int kzg_hip_multi_device(const kzg_hip_multi *m, uint32_t i) { return (m && i < m->d.size()) ? m->d[i].device : -1; }
kzg_hip_fft *kzg_hip_multi_fft(kzg_hip_multi *m, uint32_t i) { return (m && i < m->d.size()) ? m->d[i].fs : nullptr; }
kzg_hip_kzg *kzg_hip_multi_kzg(kzg_hip_multi *m, uint32_t i) { return (m && i < m->d.size()) ? m->d[i].ks : nullptr; }
const char *kzg_hip_multi_transport(const kzg_hip_multi *m) { return m ? m->transport.c_str() : ""; }
const char *kzg_hip_multi_transport_note(const kzg_hip_multi *m) { return m ? m->transport_note.c_str() : ""; }
const char *kzg_hip_multi_transport_check(const kzg_hip_multi *m) { return m ? m->self_test.c_str() : ""; }
uint64_t kzg_hip_multi_exchanges(const kzg_hip_multi *m) { return m ? m->n_allgather.load() : 0; }
int kzg_hip_multi_set_fft_sharding(kzg_hip_multi *m, int mode) {
    if (!m || mode < -1 || mode > 1) return KZG_HIP_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    m->fft_mode = mode;
    return KZG_HIP_OK;
}
int kzg_hip_multi_set_table_budget_gb(kzg_hip_multi *m, double gb) {
    if (!m) return KZG_HIP_ERR_BAD_ARG;
    for (auto &d : m->d) CHK(kzg_hip_kzg_set_table_budget_gb(d.ks, gb));
    return KZG_HIP_OK;
}

// ---- batches sharded by polynomial: device i takes rows [lo_i, hi_i) and writes its results in place ----
int kzg_hip_multi_commit_to_poly_batch(kzg_hip_multi *m, const void *coeffs_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!m || !coeffs_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_commit_to_poly_batch(m->d[i].ks, (const uint8_t *)coeffs_fr + lo * n * sizeof(fr), n, hi - lo, (uint8_t *)out_g1 + lo * sizeof(g1j));
    });
    KZG_CATCH
}
int kzg_hip_multi_compute_proof_single_batch(kzg_hip_multi *m, const void *poly_fr, uint64_t n, uint64_t batch, const uint64_t *xs, void *out_g1) {
    if (!m || !poly_fr || !xs || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_compute_proof_single_batch(m->d[i].ks, (const uint8_t *)poly_fr + lo * n * sizeof(fr), n, hi - lo, xs + lo, (uint8_t *)out_g1 + lo * sizeof(g1j));
    });
    KZG_CATCH
}

// ---- transforms over F_r on batches of rows (fft_fr.go:55-105, das_extension.go:71-84), rows divided among the devices ----
int kzg_hip_multi_fft_fr_batch(kzg_hip_multi *m, const void *vals_fr, uint64_t n, uint64_t batch, int inv, void *out_fr) {
    if (!m || !vals_fr || !out_fr) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_fft_fr_batch(m->d[i].fs, (const uint8_t *)vals_fr + lo * n * sizeof(fr), n, hi - lo, inv, (uint8_t *)out_fr + lo * n * sizeof(fr));
    });
    KZG_CATCH
}
int kzg_hip_multi_fft_g1_batch(kzg_hip_multi *m, const void *vals_g1, uint64_t n, uint64_t batch, int inv, void *out_g1) {
    if (!m || !vals_g1 || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_fft_g1_batch(m->d[i].fs, (const uint8_t *)vals_g1 + lo * n * sizeof(g1j), n, hi - lo, inv, (uint8_t *)out_g1 + lo * n * sizeof(g1j));
    });
    KZG_CATCH
}
int kzg_hip_multi_das_fft_extension_batch(kzg_hip_multi *m, void *vals_fr, uint64_t n, uint64_t batch) {
    if (!m || !vals_fr) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_das_fft_extension_batch(m->d[i].fs, (uint8_t *)vals_fr + lo * n * sizeof(fr), n, hi - lo);
    });
    KZG_CATCH
}

// ---- package eth on every device (eth/globals.go:39-72): BlobToKZGCommitment / ComputeKZGProof on batches, rows divided among the devices ----
int kzg_hip_multi_eth_settings_new(kzg_hip_multi *m, const void *lagrange_g1, uint64_t n, kzg_hip_multi_eth **out) {
    if (!m || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    KZG_TRY
    kzg_hip_multi_eth *e = new kzg_hip_multi_eth;
    e->m = m; e->n = n; e->eth.assign(m->d.size(), nullptr);
    int st = per_device(m, [&](size_t i) -> int { return kzg_hip_eth_settings_new(m->d[i].fs, lagrange_g1, n, &e->eth[i]); });
    if (st) { kzg_hip_multi_eth_settings_free(e); return st; }
    *out = e;
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_multi_eth_settings_free(kzg_hip_multi_eth *e) {
    if (!e) return;
    for (auto *p : e->eth) kzg_hip_eth_settings_free(p);
    delete e;
}
int kzg_hip_multi_eth_blob_to_kzg_commitment_batch(kzg_hip_multi_eth *e, const void *blobs_le32, uint64_t batch, void *out48, uint8_t *ok) {
    if (!e || !blobs_le32 || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    kzg_hip_multi *m = e->m;
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_eth_blob_to_kzg_commitment_batch(e->eth[i], (const uint8_t *)blobs_le32 + lo * e->n * 32, hi - lo, (uint8_t *)out48 + lo * 48, ok + lo);
    });
    KZG_CATCH
}
int kzg_hip_multi_eth_compute_kzg_proof_batch(kzg_hip_multi_eth *e, const void *polys_fr, uint64_t n, uint64_t batch, const void *zs_fr, void *out48, void *ys_fr, uint8_t *ok) {
    if (!e || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (n != e->n) return KZG_HIP_ERR_LEN_MISMATCH;                               // "polynomial has invalid length", eth/helpers.go:186-188
    if (!batch) return KZG_HIP_OK;
    if (!polys_fr || !zs_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    kzg_hip_multi *m = e->m;
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_eth_compute_kzg_proof_batch(e->eth[i], (const uint8_t *)polys_fr + lo * n * sizeof(fr), n, hi - lo, (const uint8_t *)zs_fr + lo * sizeof(fr),
                                                   (uint8_t *)out48 + lo * 48, ys_fr ? (uint8_t *)ys_fr + lo * sizeof(fr) : nullptr, ok + lo);
    });
    KZG_CATCH
}

// ---- FK20 single ----
int kzg_hip_multi_fk20_single_settings_new(kzg_hip_multi *m, uint64_t n2, kzg_hip_multi_fk20s **out) {
    if (!m || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    KZG_TRY
    kzg_hip_multi_fk20s *f = new kzg_hip_multi_fk20s;
    f->m = m; f->n2 = n2; f->fk.assign(m->d.size(), nullptr);
    int st = per_device(m, [&](size_t i) -> int { return kzg_hip_fk20_single_settings_new(m->d[i].ks, n2, &f->fk[i]); });
    if (st) { kzg_hip_multi_fk20_single_settings_free(f); return st; }
    *out = f;
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_multi_fk20_single_settings_free(kzg_hip_multi_fk20s *f) {
    if (!f) return;
    for (auto *p : f->fk) kzg_hip_fk20_single_settings_free(p);
    delete f;
}
int kzg_hip_multi_da_using_fk20_batch(kzg_hip_multi_fk20s *f, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!f || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    kzg_hip_multi *m = f->m;
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_da_using_fk20_batch(f->fk[i], (const uint8_t *)poly_fr + lo * n * sizeof(fr), n, hi - lo, (uint8_t *)out_g1 + lo * 2 * n * sizeof(g1j));
    });
    KZG_CATCH
}
int kzg_hip_multi_da_using_fk20(kzg_hip_multi_fk20s *f, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!f || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > f->m->d[0].fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;   // fk20_single.go:178-180
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;               // fk20_single.go:181-183
    if (2 * n != f->n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (f->fk.size() == 1 && !f->m->nccl) return kzg_hip_da_using_fk20(f->fk[0], poly_fr, n, out_g1);   // one entry: nothing to exchange, the single-device call has the shorter pipeline
    std::vector<fk20_core *> cores;
    for (auto *p : f->fk) cores.push_back(&p->c);
    return fk20_da_sharded(f->m, cores, poly_fr, n, out_g1);
}

// ---- FK20 multi ----
int kzg_hip_multi_fk20_multi_settings_new(kzg_hip_multi *m, uint64_t n2, uint64_t chunk_len, kzg_hip_multi_fk20m **out) {
    if (!m || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    KZG_TRY
    kzg_hip_multi_fk20m *f = new kzg_hip_multi_fk20m;
    f->m = m; f->n2 = n2; f->l = chunk_len; f->fk.assign(m->d.size(), nullptr);
    int st = per_device(m, [&](size_t i) -> int { return kzg_hip_fk20_multi_settings_new(m->d[i].ks, n2, chunk_len, &f->fk[i]); });
    if (st) { kzg_hip_multi_fk20_multi_settings_free(f); return st; }
    *out = f;
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_multi_fk20_multi_settings_free(kzg_hip_multi_fk20m *f) {
    if (!f) return;
    for (auto *p : f->fk) kzg_hip_fk20_multi_settings_free(p);
    delete f;
}
int kzg_hip_multi_da_using_fk20_multi_batch(kzg_hip_multi_fk20m *f, const void *poly_fr, uint64_t n, uint64_t batch, void *out_g1) {
    if (!f || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    kzg_hip_multi *m = f->m;
    const uint64_t per = 2 * n / f->l;
    return per_device(m, [&](size_t i) -> int {
        uint64_t lo, hi; share(batch, m->d.size(), i, &lo, &hi);
        if (hi == lo) return KZG_HIP_OK;
        return kzg_hip_da_using_fk20_multi_batch(f->fk[i], (const uint8_t *)poly_fr + lo * n * sizeof(fr), n, hi - lo, (uint8_t *)out_g1 + lo * per * sizeof(g1j));
    });
    KZG_CATCH
}
int kzg_hip_multi_da_using_fk20_multi(kzg_hip_multi_fk20m *f, const void *poly_fr, uint64_t n, void *out_g1) {
    if (!f || !poly_fr || !out_g1) return KZG_HIP_ERR_BAD_ARG;
    if (n > f->m->d[0].fs->W / 2) return KZG_HIP_ERR_TOO_WIDE;     // fk20_multi.go:115-117
    if (!is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;                 // fk20_multi.go:118-120
    if (2 * n != f->n2) return KZG_HIP_ERR_LEN_MISMATCH;
    if (f->fk.size() == 1 && !f->m->nccl) return kzg_hip_da_using_fk20_multi(f->fk[0], poly_fr, n, out_g1);
    std::vector<fk20_core *> cores;
    for (auto *p : f->fk) cores.push_back(&p->c);
    return fk20_da_sharded(f->m, cores, poly_fr, n, out_g1);
}

}  // extern "C"
