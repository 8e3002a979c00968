// coalesce.hpp -- request coalescing behind the reference's one-object-per-call API.
//
// The reference's methods take ONE polynomial per call (KZGSettings.CommitToPoly kzg_single_proofs.go:17-19,
// ComputeProofSingle :36-54, FK20SingleSettings.DAUsingFK20 fk20_single.go:176-196) and are called from many goroutines; one
// polynomial fills 1-3 % of an MI355X.  Concurrent calls on a handle are therefore merged into batched launches:
//
//   caller thread:  take a row of the OPEN staging buffer (pinned host memory) with ONE atomic add  ->  copy its input into that row (in
//                   parallel with the other callers)  ->  the caller that took row 0 is the batch's leader, the others sleep on a futex word
//                   ->  when the batch is done, copy its own result out of the pinned output rows.
//   leader:         waits for a device slot (while it waits, callers keep joining its buffer: with every slot busy a batch grows to
//                   whatever arrived during the previous batch -- no timer involved), then for its share of the recent callers (at most
//                   `window` microseconds; a steady lone caller never waits), closes the buffer (later arrivals go to the next one), waits
//                   until every taken row is filled, runs the batch on the buffer's own stream and wakes the batch's callers.
//
// Round 5: NO mutex on the callers' path.  Rounds 2-4 took the coalescer's mutex three times per call (row reservation, "my row is filled",
// leader election); with 256 caller threads on the 16 cores of a bench box the hand-offs of that one mutex cost more than the batches gained
// (64 threads 51 k commitments/s, 256 threads 38 k/s).  Now a call is: fetch_add on the open buffer's state word, memcpy, fetch_add on its
// `ready` counter, futex sleep -- the mutex is taken twice per BATCH (close + open the next buffer; recycle).
//
// One batch is in flight up to ~96 concurrent callers (KZG_COALESCE_CALLERS_PER_BATCH; 48 for the FK20 pipelines), up to MAX_EXEC (on separate streams) beyond: the end of a batch (reduction
// trees, one inversion per polynomial) is latency-bound and uses a fraction of the CUs, so with hundreds of callers the next batch's
// table walk overlaps it.  The executing function is supplied by the handle (commit / proof / FK20); everything here is host-side C++.
//
// Wake-ups.  The end of a batch releases all of its callers at once: they sleep on NWAKE futex words per buffer (`done`), the leader wakes
// ONE sleeper per word and each of those wakes the rest of its word (16 + 3 wake-ups on the critical path for 64 callers instead of 63:
// FUTEX_WAKE costs the waker ~1.5 us per thread).  Before parking, a follower spins briefly -- only while fewer than MAX_SPINNERS do:
// with a handful of callers the batch is back before a park / wake round trip would be, with hundreds the spinners would only take the
// cores from the row copies.  (Linux only, like ROCm.)
#pragma once
#ifndef KZG_COALESCE_SIM
#include <hip/hip_runtime.h>
void stream_cache_own(hipStream_t s);      // capi_core.hip: small stream-ordered temporaries of library-owned streams are recycled per stream
void stream_cache_disown(hipStream_t s);
#endif
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <functional>
#include <mutex>
#include <vector>
#include <time.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#ifndef KZG_COALESCE_CALLERS_PER_BATCH
#define KZG_COALESCE_CALLERS_PER_BATCH 96   // concurrent callers per batch in flight: two half-size walks take 1.3 times one full-size walk, so 64 callers run as ONE batch (round-5 sweep, tools/coalesce_sweep.sh: 48 / 96 -> 53.8 k / 55.3 k at 64 callers, 256 callers equal)
#endif

namespace kzg {

struct coalesce_row {        // what the executor sees for row b of a batch
    uint64_t n;              // valid input elements of this row (the rest of the row is unspecified: the executor zero-fills on device)
    uint64_t arg;            // per-request scalar argument (ComputeProofSingle's x)
};

static inline void futex_wait_u32(std::atomic<uint32_t> *a, uint32_t seen) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
static inline void futex_wait_u32_for(std::atomic<uint32_t> *a, uint32_t seen, long ns) {
    if (ns <= 0) return;
    struct timespec ts; ts.tv_sec = ns / 1000000000L; ts.tv_nsec = ns % 1000000000L;
    syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, seen, &ts, nullptr, 0);
}
static inline long futex_wake_u32(std::atomic<uint32_t> *a, int n) { return syscall(SYS_futex, (uint32_t *)a, FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0); }
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");

struct coalesce_buf {
    // ---- what an executor sees
    uint8_t *h_in = nullptr, *h_out = nullptr;   // pinned: max_batch x in_row_bytes, max_batch x out_row_bytes
    coalesce_row *h_meta = nullptr;              // pinned: row i's (n, arg), written by the caller that took row i
    int status = 0;
#ifndef KZG_COALESCE_SIM
    hipStream_t stream = nullptr;
#else
    void *stream = nullptr;
#endif
    // ---- protocol
    static constexpr uint64_t CLOSED = 1ull << 32;
    std::atomic<uint64_t> state{CLOSED};         // low 32 bits: rows taken so far; CLOSED: no more rows (also while the buffer is not the open one)
    std::atomic<uint32_t> ready{0};              // rows whose input copy has finished
    std::atomic<uint32_t> count{0};              // rows of the closed batch (0 while it is open)
    std::atomic<uint32_t> target{0};             // rows its gathering leader is waiting for (0: not gathering)
    std::atomic<uint32_t> outstanding{0};        // callers of the closed batch that still have to copy their result out
    std::atomic<uint32_t> done_flag{0};          // 1: the batch has executed, `status` and h_out are valid
    std::atomic<uint32_t> lead_word{0};          // futex word of the leader (bumped when the target / the last filled row arrives)
    enum phase_t : int { FREE, OPEN, BUSY };
    int phase = FREE;                            // under the coalescer's mutex
    static constexpr int NWAKE = 16;
    struct alignas(64) wake_word { std::atomic<uint32_t> v{0}; };
    wake_word done[NWAKE];                       // followers sleep on done[row % NWAKE]
};

class coalescer {
  public:
    static constexpr int NBUF = 4, MAX_EXEC = 3, MAX_SPINNERS = 6;
    // exec(buf, batch): inputs are in buf.h_in (row stride in_row_bytes), results go to buf.h_out (row stride out_row_bytes);
    // must block until the results are in host memory; returns a status that every request of the batch receives
    using exec_fn = std::function<int(coalesce_buf &, uint64_t batch)>;

    // callers_per_batch: concurrent callers per batch in flight (0: KZG_COALESCE_CALLERS_PER_BATCH).  The table walk of a commitment wants ONE batch for 64 callers
    // (96); pipelines whose launches fill the chip from ~32 polynomials on and whose batches take tens of milliseconds (FK20) run two half batches side by side (48).
    coalescer(int device, size_t in_row_bytes, size_t out_row_bytes, uint64_t max_batch, int callers_per_batch = 0)
        : device_(device), in_row_(in_row_bytes), out_row_(out_row_bytes), max_batch_(max_batch) {
        if (callers_per_batch > 0) per_batch_ = callers_per_batch;
        if (const char *e = getenv("KZG_HIP_COALESCE_US")) window_us_ = atol(e);
        if (const char *e = getenv("KZG_HIP_COALESCE_EXEC")) { max_exec_ = atoi(e); if (max_exec_ < 1) max_exec_ = 1; if (max_exec_ > NBUF - 1) max_exec_ = NBUF - 1; }
        if (const char *e = getenv("KZG_HIP_COALESCE_SPIN_US")) spin_us_ = atol(e);
        if (const char *e = getenv("KZG_HIP_COALESCE_PER_BATCH")) { per_batch_ = atoi(e); if (per_batch_ < 1) per_batch_ = 1; }
    }
    ~coalescer() {
        if (getenv("KZG_HIP_COALESCE_STATS") && batches_)
            fprintf(stderr, "[coalescer] %llu requests in %llu batches (avg %.1f), per batch: %.3f ms executing, %.3f ms waiting for a device slot, %.3f ms gathering callers, %.3f ms waiting for row copies\n",
                    (unsigned long long)requests_.load(), (unsigned long long)batches_.load(), (double)requests_.load() / batches_.load(), exec_ns_.load() * 1e-6 / batches_.load(),
                    slot_ns_.load() * 1e-6 / batches_.load(), gather_ns_.load() * 1e-6 / batches_.load(), ready_ns_.load() * 1e-6 / batches_.load());
        for (auto &b : bufs_) free_buf(b);
    }
    // cumulative statistics since the handle's first call: {requests, batches, ns executing, ns waiting for a device slot, ns gathering callers, ns waiting for row copies,
    // most callers seen inside submit() at once (decaying estimate), batches allowed in flight right now}
    void stats(uint64_t out[8]) const {
        out[0] = requests_.load(); out[1] = batches_.load(); out[2] = (uint64_t)exec_ns_.load(); out[3] = (uint64_t)slot_ns_.load();
        out[4] = (uint64_t)gather_ns_.load(); out[5] = (uint64_t)ready_ns_.load(); out[6] = peak_seen_.load(); out[7] = (uint64_t)exec_limit_.load();
    }
    size_t in_row_bytes() const { return in_row_; }
    size_t out_row_bytes() const { return out_row_; }

    // one request: `in` holds in_bytes (<= in_row_bytes), the result (out_bytes <= out_row_bytes) is written to `out`.
    // (in2: an optional second input, copied behind the first one in the request's row -- ComputeKZGProof's z after its polynomial)
    int submit(const void *in, size_t in_bytes, uint64_t n, uint64_t arg, void *out, size_t out_bytes, const exec_fn &exec, int alloc_error_status,
               const void *in2 = nullptr, size_t in2_bytes = 0) {
#ifndef KZG_COALESCE_SIM
        hipSetDevice(device_);                   // caller threads (goroutine-backed OS threads) start on device 0
#endif
        const uint32_t in_now = inside_.fetch_add(1, std::memory_order_relaxed) + 1;
        { uint32_t m = inside_max_.load(std::memory_order_relaxed); while (in_now > m && !inside_max_.compare_exchange_weak(m, in_now, std::memory_order_relaxed)) {} }
        // ---- take a row of the open buffer
        coalesce_buf *bp = nullptr; uint32_t row = 0;
        for (;;) {
            const uint32_t seq = open_seq_.load(std::memory_order_acquire);
            int bi = open_.load(std::memory_order_acquire);
            if (bi == OPEN_UNINIT) {                 // first call on this coalescer (or after an allocation failure): open buffer 0
                std::lock_guard<std::mutex> lk(mu_);
                if (open_.load(std::memory_order_relaxed) == OPEN_UNINIT && !open_next_locked(-1)) { inside_.fetch_sub(1, std::memory_order_relaxed); return alloc_error_status; }
                continue;
            }
            if (bi < 0) { futex_wait_u32(&open_seq_, seq); continue; }           // every buffer busy: the next recycle opens one
            coalesce_buf &b = bufs_[bi];
            // seq_cst: with the gathering leader this is a store-buffering (Dekker) pair -- leader: target.store; state.load -- follower: state.fetch_add; target.load --
            // and only a total order over all four accesses excludes "leader reads the old state AND follower reads target == 0" (a missed wake-up that costs the batch
            // its whole gather window)
            const uint64_t old = b.state.fetch_add(1, std::memory_order_seq_cst);
            if (!(old & coalesce_buf::CLOSED) && (uint32_t)old < max_batch_) { bp = &b; row = (uint32_t)old; break; }
            if (!(old & coalesce_buf::CLOSED) && (uint32_t)old == max_batch_) bump_and_wake(b.lead_word);   // full: its leader need not wait for more
            if (open_seq_.load(std::memory_order_acquire) == seq) futex_wait_u32(&open_seq_, seq);      // closed or full: until another buffer opens
        }
        coalesce_buf &b = *bp;
        b.h_meta[row] = coalesce_row{n, arg};
        memcpy(b.h_in + (size_t)row * in_row_, in, in_bytes);                      // parallel across callers
        if (in2_bytes) memcpy(b.h_in + (size_t)row * in_row_ + in_bytes, in2, in2_bytes);
        const uint32_t filled = b.ready.fetch_add(1, std::memory_order_seq_cst) + 1;
        if (row == 0) lead(b, exec, alloc_error_status);
        else {
            // the leader sleeps until its target is reached and, after closing, until the last taken row is filled
            const uint32_t cnt = b.count.load(std::memory_order_seq_cst), tgt = b.target.load(std::memory_order_seq_cst);
            if ((cnt && filled >= cnt) || (tgt && row + 1 == tgt)) bump_and_wake(b.lead_word);
            wait_done(b, row);
        }
        const int st = b.status;
        if (st == 0) memcpy(out, b.h_out + (size_t)row * out_row_, out_bytes);
        inside_.fetch_sub(1, std::memory_order_relaxed);
        if (b.outstanding.fetch_sub(1, std::memory_order_acq_rel) == 1) recycle(b);   // last one out recycles the buffer (only it takes the mutex)
        return st;
    }

  private:
    static constexpr int OPEN_UNINIT = -2;
    static long now_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000L + ts.tv_nsec; }
    static void bump_and_wake(std::atomic<uint32_t> &w) { w.fetch_add(1, std::memory_order_seq_cst); futex_wake_u32(&w, 1); }
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }

    void free_buf(coalesce_buf &b) {
#ifndef KZG_COALESCE_SIM
        if (b.h_in) hipHostFree(b.h_in);
        if (b.h_out) hipHostFree(b.h_out);
        if (b.h_meta) hipHostFree(b.h_meta);
        if (b.stream) { stream_cache_disown(b.stream); hipStreamDestroy(b.stream); }
        (void)hipGetLastError();
#else
        free(b.h_in); free(b.h_out); free(b.h_meta);
#endif
        b.h_in = b.h_out = nullptr; b.h_meta = nullptr; b.stream = nullptr;
    }
    bool alloc_buf(coalesce_buf &b) {            // first use of this buffer: pinned staging + its stream (under the mutex: NBUF times per handle)
        if (b.h_in) return true;
#ifndef KZG_COALESCE_SIM
        if (hipHostMalloc((void **)&b.h_in, in_row_ * max_batch_, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&b.h_out, out_row_ * max_batch_, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&b.h_meta, sizeof(coalesce_row) * max_batch_, hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            free_buf(b);
            return false;
        }
        stream_cache_own(b.stream);
#else
        // (simulation: KZG_COALESCE_SIM_MAX_BUFS staging buffers can be allocated, the next allocation fails -- pinned-memory pressure)
        if (const char *e = getenv("KZG_COALESCE_SIM_MAX_BUFS")) {
            int have = 0;
            for (auto &x : bufs_) have += x.h_in != nullptr;
            if (have >= atoi(e)) return false;
        }
        b.h_in = (uint8_t *)malloc(in_row_ * max_batch_); b.h_out = (uint8_t *)malloc(out_row_ * max_batch_); b.h_meta = (coalesce_row *)malloc(sizeof(coalesce_row) * max_batch_);
#endif
        return true;
    }
    // mu_ held.  Makes a FREE buffer (not `except`) the open one, preferring one whose staging memory exists: a new one is allocated only when no allocated buffer
    // is free, and a failed allocation is an error only when nothing allocated is left to recycle either (then open_ = OPEN_UNINIT: the next caller tries again and
    // reports the failure itself).  With every usable buffer busy open_ = -1 and the next recycle opens one: under pinned-memory pressure the handle runs with fewer
    // staging buffers instead of failing calls.
    bool open_next_locked(int except) {
        int pick = -1, fresh = -1, busy_allocated = 0;
        for (int k = 0; k < NBUF; k++) {
            const int o = (except < 0 ? k : (except + 1 + k) % NBUF);
            coalesce_buf &c = bufs_[o];
            if (o == except || c.phase != coalesce_buf::FREE) { if (c.h_in) busy_allocated++; continue; }
            if (c.h_in) { pick = o; break; }
            if (fresh < 0) fresh = o;
        }
        if (pick < 0 && fresh >= 0) {
            if (alloc_buf(bufs_[fresh])) pick = fresh;
            else if (!busy_allocated) { open_.store(OPEN_UNINIT, std::memory_order_release); bump_open_seq(); return false; }
        }
        if (pick >= 0) {
            coalesce_buf &c = bufs_[pick];
            c.phase = coalesce_buf::OPEN;
            c.ready.store(0, std::memory_order_relaxed); c.count.store(0, std::memory_order_relaxed); c.target.store(0, std::memory_order_relaxed);
            c.done_flag.store(0, std::memory_order_relaxed); c.status = 0;
            c.state.store(0, std::memory_order_release);
            open_.store(pick, std::memory_order_release);
            bump_open_seq();
            return true;
        }
        open_.store(-1, std::memory_order_release);
        bump_open_seq();
        return true;
    }
    void bump_open_seq() { open_seq_.fetch_add(1, std::memory_order_seq_cst); futex_wake_u32(&open_seq_, INT_MAX); }   // (callers that found the open buffer full or closed: rare)
    void recycle(coalesce_buf &b) {
        std::lock_guard<std::mutex> lk(mu_);
        b.phase = coalesce_buf::FREE;
        if (open_.load(std::memory_order_relaxed) == -1) open_next_locked(-1);
    }

    // a follower waits for its batch: brief spinning while few callers do, then a futex sleep on its word; the woken one wakes the rest of its word
    void wait_done(coalesce_buf &b, uint32_t row) {
        std::atomic<uint32_t> *word = &b.done[row % coalesce_buf::NWAKE].v;
        if (spin_us_ > 0 && spinners_.load(std::memory_order_relaxed) < MAX_SPINNERS) {
            spinners_.fetch_add(1, std::memory_order_relaxed);
            const long until = now_ns() + spin_us_ * 1000L;
            bool got = false;
            for (int i = 0;; i++) {
                if (b.done_flag.load(std::memory_order_acquire)) { got = true; break; }
                cpu_relax();
                if ((i & 63) == 63 && now_ns() > until) break;
            }
            spinners_.fetch_sub(1, std::memory_order_relaxed);
            if (got) return;
        }
        for (;;) {
            const uint32_t seen = word->load(std::memory_order_acquire);
            if (b.done_flag.load(std::memory_order_acquire)) break;
            futex_wait_u32(word, seen);
        }
        futex_wake_u32(word, INT_MAX);           // second level of the fan-out (a no-op when nobody else sleeps on this word)
    }

    // the caller that took row 0 runs the batch
    void lead(coalesce_buf &b, const exec_fn &exec, int alloc_error_status) {
        // (1) a device slot.  While this waits, callers keep joining the buffer: under load a batch is whatever arrived during the previous one.
        const long s0 = now_ns();
        for (;;) {
            const uint32_t seen = slot_word_.load(std::memory_order_acquire);
            int e = executing_.load(std::memory_order_acquire);
            if (e < exec_limit_.load(std::memory_order_relaxed)) { if (executing_.compare_exchange_weak(e, e + 1, std::memory_order_acq_rel)) break; continue; }
            futex_wait_u32(&slot_word_, seen);
        }
        const long s1 = now_ns();
        // (2) this batch's share of the concurrent callers gets up to `window_us_` to join, so that N concurrent callers run as exec_limit
        // overlapping batches of N / exec_limit instead of a convoy of tiny ones.  `peak_`: the most callers seen inside submit() since the
        // previous batch was formed -- at the moment a leader gets its slot most of them are between two calls -- decaying by a quarter per
        // batch once they stop coming (a lone caller must end at a target of 1: it never waits).
        uint32_t peak;
        {
            std::lock_guard<std::mutex> lk(mu_);
            // (after an idle gap -- no batch formed for 2 ms and for four batch times -- the estimate is forgotten at once: a lone caller that follows a burst
            // of 64 must not pay the burst's gather window for the dozen calls the decay takes)
            const long idle_ns = s1 - last_close_ns_;
            const bool idle = last_close_ns_ != 0 && idle_ns > 2000000L && idle_ns > 4 * exec_ema_ns_.load(std::memory_order_relaxed);
            last_close_ns_ = s1;
            const uint32_t decayed = idle ? 0u : peak_ - (peak_ + 3) / 4;  // rounds up: 3 -> 2 -> 1 -> 0
            uint32_t seen_max = inside_max_.exchange(inside_.load(std::memory_order_relaxed), std::memory_order_relaxed);
            if (idle) seen_max = inside_.load(std::memory_order_relaxed);  // (the maximum since the last batch is the burst's tail: forgotten with the rest)
            peak_ = seen_max > decayed ? seen_max : decayed;
            peak = peak_;
            if (peak > peak_seen_.load(std::memory_order_relaxed)) peak_seen_.store(peak, std::memory_order_relaxed);
            // Batches in flight: ONE up to ~48 concurrent callers (a table walk over fewer than ~50 polynomials leaves lanes idle and pays its
            // reduction tree in full, so two half-size walks take 1.3 times one full-size walk), a second and third one beyond, where a batch
            // is large enough to walk efficiently and the host side of a batch (hundreds of wake-ups and 128 KiB row copies) is worth overlapping.
            int lim = (int)(1 + peak / (uint32_t)per_batch_);
            if (lim > max_exec_) lim = max_exec_;
            exec_limit_.store(lim, std::memory_order_relaxed);
        }
        uint32_t target = (peak + (uint32_t)exec_limit_.load(std::memory_order_relaxed) - 1) / (uint32_t)exec_limit_.load(std::memory_order_relaxed);
        if (target > max_batch_) target = (uint32_t)max_batch_;
        if (window_us_ > 0 && (uint32_t)b.state.load(std::memory_order_acquire) < target) {
            // the window grows with the work it precedes: 5 % of the recent execution time of a batch (a 36 ms FK20 batch can afford 1.8 ms for
            // callers that need a millisecond to come back), never less than `window_us_`
            const long grown = exec_ema_ns_.load(std::memory_order_relaxed) / 20;
            const long deadline = s1 + (grown > window_us_ * 1000L ? grown : window_us_ * 1000L);
            b.target.store(target, std::memory_order_seq_cst);
            for (;;) {
                const uint32_t seen = b.lead_word.load(std::memory_order_seq_cst);
                if ((uint32_t)b.state.load(std::memory_order_seq_cst) >= target) break;
                const long left = deadline - now_ns();
                if (left <= 0) break;
                futex_wait_u32_for(&b.lead_word, seen, left);
            }
            b.target.store(0, std::memory_order_relaxed);
        }
        const long s2 = now_ns();
        // (3) close this buffer, open a free one
        const uint64_t old = b.state.fetch_or(coalesce_buf::CLOSED, std::memory_order_acq_rel);
        const uint32_t batch = (uint32_t)old < max_batch_ ? (uint32_t)old : (uint32_t)max_batch_;
        b.outstanding.store(batch, std::memory_order_relaxed);
        b.count.store(batch, std::memory_order_seq_cst);
        {
            std::lock_guard<std::mutex> lk(mu_);
            b.phase = coalesce_buf::BUSY;
            open_next_locked((int)(&b - bufs_));
        }
        // (4) every taken row filled
        for (;;) {
            const uint32_t seen = b.lead_word.load(std::memory_order_seq_cst);
            if (b.ready.load(std::memory_order_seq_cst) >= batch) break;
            futex_wait_u32(&b.lead_word, seen);
        }
        const long s3 = now_ns();
        int st;
        try { st = exec(b, batch); } catch (...) { st = alloc_error_status; }   // (std::bad_alloc in the executor must not strand the sleepers)
        const long s4 = now_ns();
        batches_.fetch_add(1, std::memory_order_relaxed); requests_.fetch_add(batch, std::memory_order_relaxed);
        exec_ns_.fetch_add(s4 - s3, std::memory_order_relaxed); slot_ns_.fetch_add(s1 - s0, std::memory_order_relaxed);
        gather_ns_.fetch_add(s2 - s1, std::memory_order_relaxed); ready_ns_.fetch_add(s3 - s2, std::memory_order_relaxed);
        { const long ema = exec_ema_ns_.load(std::memory_order_relaxed); exec_ema_ns_.store(ema == 0 ? (s4 - s3) : (3 * ema + (s4 - s3)) / 4, std::memory_order_relaxed); }
        b.status = st;
        b.done_flag.store(1, std::memory_order_release);
        for (auto &w : b.done) w.v.fetch_add(1, std::memory_order_release);
        // (5) hand the device slot to the next leader first (its batch is complete and waiting), then release this batch's callers
        executing_.fetch_sub(1, std::memory_order_acq_rel);
        bump_and_wake(slot_word_);
        if (batch > 1) for (auto &w : b.done) futex_wake_u32(&w.v, 1);        // one per word, each wakes its word's rest
    }

    std::mutex mu_;                      // buffer phases, open_ transitions, the concurrency estimate: taken by leaders and recyclers only
    coalesce_buf bufs_[NBUF];
    std::atomic<int> open_{OPEN_UNINIT}; // buffer accepting rows; -1 while all are busy
    std::atomic<uint32_t> open_seq_{0};  // futex word: bumped whenever open_ changes
    std::atomic<int> executing_{0};      // batches that hold a device slot
    std::atomic<uint32_t> slot_word_{0}; // futex word: bumped when a slot is released
    int max_exec_ = MAX_EXEC;            // upper bound (KZG_HIP_COALESCE_EXEC)
    std::atomic<int> exec_limit_{1};     // batches allowed in flight right now: 1 .. max_exec_ by the number of concurrent callers
    std::atomic<uint32_t> inside_{0};    // callers currently inside submit()
    std::atomic<uint32_t> inside_max_{0};// its maximum since the last batch was formed
    uint32_t peak_ = 0;                  // decaying maximum of inside_: the concurrency the gather targets are derived from (under mu_)
    long last_close_ns_ = 0;             // when the previous batch's leader got its slot (under mu_)
    std::atomic<int> spinners_{0};
    long window_us_ = 150;               // upper bound of the gather wait (KZG_HIP_COALESCE_US; 0 disables)
    long spin_us_ = 40;                  // a follower's spin before it parks (KZG_HIP_COALESCE_SPIN_US; 0 disables)
    int per_batch_ = KZG_COALESCE_CALLERS_PER_BATCH;   // concurrent callers per batch in flight (KZG_HIP_COALESCE_PER_BATCH)
    std::atomic<uint64_t> batches_{0}, requests_{0};   // statistics (KZG_HIP_COALESCE_STATS=1 prints them when the handle is freed)
    std::atomic<long> exec_ns_{0}, slot_ns_{0}, gather_ns_{0}, ready_ns_{0};
    std::atomic<uint64_t> peak_seen_{0}; // largest concurrency estimate a leader ever used (statistics)
    std::atomic<long> exec_ema_ns_{0};   // recent execution time of a batch (the gather window scales with it)
    int device_;
    size_t in_row_, out_row_;
    uint64_t max_batch_;
};

}  // namespace kzg
