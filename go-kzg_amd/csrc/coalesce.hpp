// coalesce.hpp -- request coalescing behind the reference's one-object-per-call API.
//
// The reference's methods take ONE polynomial per call (KZGSettings.CommitToPoly kzg_single_proofs.go:17-19,
// ComputeProofSingle :36-54, FK20SingleSettings.DAUsingFK20 fk20_single.go:176-196) and are called from many goroutines; one
// polynomial fills 1-3 % of an MI355X.  Concurrent calls on a handle are therefore merged into batched launches:
//
//   caller thread:  reserve a row in the OPEN staging buffer (pinned host memory)  ->  copy its input into that row (in
//                   parallel with the other callers)  ->  if fewer than MAX_EXEC batches are on the device, close the buffer and
//                   become its leader, else sleep  ->  when the batch is done, copy its own result out of the pinned output rows.
//   leader:         waits until every reserved row is filled, then runs the batch on the buffer's own stream (inputs read from
//                   pinned memory, batched kernels, one D2H) and wakes the batch's callers.
//
// One batch is in flight up to ~48 concurrent callers, up to MAX_EXEC (on separate streams) beyond: the end of a batch (reduction
// trees, one inversion per polynomial) is latency-bound and uses a fraction of the CUs, so with hundreds of callers the next batch's
// table walk overlaps it.  The elected leader holds its buffer open for at most `window` microseconds until it has its share
// (concurrency / batches in flight) of the recent callers; a steady lone caller never waits: nothing is added to its latency.
// The executing function is supplied by the handle (commit / proof / FK20); everything here is host-side C++.
//
// Wake-ups.  The end of a batch releases all of its callers at once.  Through a condition variable every one of them re-acquires the
// coalescer's mutex on the way out -- a convoy of futex hand-offs that was measured at ~0.4 ms per batch with 64 callers (the device
// sat idle 28 % of the time and the next leader's gather window expired with 50 of the 64 callers back).  The callers of a batch
// therefore sleep on a per-buffer futex word (`epoch`) and leave WITHOUT the mutex when they find the batch done; only leader
// election and the row reservation take it.  (Linux only, like ROCm.)
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <functional>
#include <mutex>
#include <vector>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#ifndef KZG_COALESCE_CALLERS_PER_BATCH
#define KZG_COALESCE_CALLERS_PER_BATCH 48   // concurrent callers per batch in flight (96 until the small-batch walk got cheaper: 48 measured +2 % at 64 callers, +10 % for proofs)
#endif

namespace kzg {

struct coalesce_row {        // what the executor sees for row b of a batch
    uint64_t n;              // valid input elements of this row (the rest of the row is unspecified: the executor zero-fills on device)
    uint64_t arg;            // per-request scalar argument (ComputeProofSingle's x)
};

static inline void futex_wait_u32(std::atomic<uint32_t> *a, uint32_t seen) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
static inline long futex_wake_u32(std::atomic<uint32_t> *a, int n) { return syscall(SYS_futex, (uint32_t *)a, FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0); }
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");

struct coalesce_buf {
    enum state_t : int { FREE_OPEN, CLOSED, DRAINING };
    std::atomic<int> state{FREE_OPEN};           // written under the coalescer's mutex; DRAINING is also read without it (after `status`)
    // futex words: bumped (under the mutex) by every event this buffer's sleepers care about.  A caller sleeps on word (row % NWAKE); the
    // end of a batch wakes ONE sleeper per word and each of those wakes the rest of its word: a two-level fan-out (16 + 3 wake-ups on
    // the critical path for 64 callers instead of 63: FUTEX_WAKE costs the waker ~1.5 us per thread)
    static constexpr int NWAKE = 16;
    struct alignas(64) wake_word { std::atomic<uint32_t> v{0}; };
    wake_word epoch[NWAKE];
    void bump_all() { for (auto &w : epoch) w.v.fetch_add(1, std::memory_order_release); }
    uint8_t *h_in = nullptr, *h_out = nullptr;   // pinned: max_batch x in_row_bytes, max_batch x out_row_bytes
    coalesce_row *h_meta = nullptr;              // pinned copy of `rows` for the executor's H2D (filled by the leader)
    std::vector<coalesce_row> rows;              // reserved rows, in order
    uint64_t ready = 0;                          // rows whose input copy has finished
    std::atomic<uint64_t> outstanding{0};        // callers that still have to copy their result out
    int status = 0;
    hipStream_t stream = nullptr;
    uint64_t target = 0;                         // rows its gathering leader is waiting for
};

class coalescer {
  public:
    static constexpr int NBUF = 4, MAX_EXEC = 3;
    // exec(buf, batch): inputs are in buf.h_in (row stride in_row_bytes), results go to buf.h_out (row stride out_row_bytes);
    // must block until the results are in host memory; returns a status that every request of the batch receives
    using exec_fn = std::function<int(coalesce_buf &, uint64_t batch)>;

    coalescer(int device, size_t in_row_bytes, size_t out_row_bytes, uint64_t max_batch)
        : device_(device), in_row_(in_row_bytes), out_row_(out_row_bytes), max_batch_(max_batch) {
        if (const char *e = getenv("KZG_HIP_COALESCE_US")) window_us_ = atol(e);
        if (const char *e = getenv("KZG_HIP_COALESCE_EXEC")) { max_exec_ = atoi(e); if (max_exec_ < 1) max_exec_ = 1; if (max_exec_ > NBUF - 1) max_exec_ = NBUF - 1; }
    }
    ~coalescer() {
        if (getenv("KZG_HIP_COALESCE_STATS") && batches_)
            fprintf(stderr, "[coalescer] %llu requests in %llu batches (avg %.1f), per batch: %.3f ms executing, %.3f ms gathering callers, %.3f ms waiting for row copies\n",
                    (unsigned long long)requests_, (unsigned long long)batches_, (double)requests_ / batches_, exec_s_ / batches_ * 1e3, gather_s_ / batches_ * 1e3,
                    ready_s_ / batches_ * 1e3);
        for (auto &b : bufs_) {
            if (b.h_in) hipHostFree(b.h_in);
            if (b.h_out) hipHostFree(b.h_out);
            if (b.h_meta) hipHostFree(b.h_meta);
            if (b.stream) hipStreamDestroy(b.stream);
        }
        (void)hipGetLastError();
    }
    size_t in_row_bytes() const { return in_row_; }
    size_t out_row_bytes() const { return out_row_; }

    // one request: `in` holds in_bytes (<= in_row_bytes), the result (out_bytes <= out_row_bytes) is written to `out`.
    // Wake-ups are targeted: a futex word per buffer for its callers, one condition variable for the (single) gathering leader and
    // one for callers that found every buffer busy.
    // (in2: an optional second input, copied behind the first one in the request's row -- ComputeKZGProof's z after its polynomial)
    int submit(const void *in, size_t in_bytes, uint64_t n, uint64_t arg, void *out, size_t out_bytes, const exec_fn &exec, int alloc_error_status,
               const void *in2 = nullptr, size_t in2_bytes = 0) {
        hipSetDevice(device_);                   // caller threads (goroutine-backed OS threads) start on device 0
        std::unique_lock<std::mutex> lk(mu_);
        const uint64_t in_now0 = inside_.fetch_add(1, std::memory_order_relaxed) + 1;
        if (in_now0 > inside_max_) inside_max_ = in_now0;
        // reserve a row in the open buffer
        int bi;
        for (;;) {
            bi = open_;
            if (bi >= 0 && bufs_[bi].rows.size() < max_batch_) break;
            cv_reserve_.wait(lk);
        }
        coalesce_buf &b = bufs_[bi];
        if (!b.h_in) {   // first use of this buffer: pinned staging + its stream (under the lock: NBUF times per handle)
            if (hipHostMalloc((void **)&b.h_in, in_row_ * max_batch_, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&b.h_out, out_row_ * max_batch_, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&b.h_meta, sizeof(coalesce_row) * max_batch_, hipHostMallocDefault) != hipSuccess ||
                hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                if (b.h_in) { hipHostFree(b.h_in); b.h_in = nullptr; }
                if (b.h_out) { hipHostFree(b.h_out); b.h_out = nullptr; }
                if (b.h_meta) { hipHostFree(b.h_meta); b.h_meta = nullptr; }
                inside_.fetch_sub(1, std::memory_order_relaxed);
                return alloc_error_status;
            }
        }
        const uint64_t row = b.rows.size();
        b.rows.push_back(coalesce_row{n, arg});
        b.outstanding++;
        if (gathering_ == bi && b.rows.size() >= b.target) cv_leader_.notify_all();   // the leader holding this buffer open has its share
        lk.unlock();
        memcpy(b.h_in + row * in_row_, in, in_bytes);                      // parallel across callers
        if (in2_bytes) memcpy(b.h_in + row * in_row_ + in_bytes, in2, in2_bytes);
        lk.lock();
        b.ready++;
        if (b.state.load(std::memory_order_relaxed) == coalesce_buf::CLOSED && b.ready == b.rows.size()) cv_leader_.notify_all();   // its leader waits for the last row
        // wait for the batch; lead it if a device slot is free
        bool locked = true;
        for (;;) {
            if (b.state.load(std::memory_order_relaxed) == coalesce_buf::DRAINING) break;
            if (b.state.load(std::memory_order_relaxed) == coalesce_buf::FREE_OPEN && executing_ < exec_limit_ && gathering_ < 0) {
                executing_++;
                // Gather: this batch's share of the concurrent callers gets up to `window_us_` to join, so that N concurrent callers
                // run as MAX_EXEC overlapping batches of N / MAX_EXEC instead of a convoy of tiny ones.
                // (`peak_`: the most callers seen inside submit() since the previous batch was formed -- at the moment a leader is
                // elected most of them are between two calls -- decaying by a quarter per batch once they stop coming)
                const uint64_t decayed = peak_ - (peak_ + 3) / 4;          // rounds up: 3 -> 2 -> 1 -> 0 (a lone caller must end at a target of 1)
                peak_ = inside_max_ > decayed ? inside_max_ : decayed;
                inside_max_ = inside_.load(std::memory_order_relaxed);
                // Batches in flight: ONE up to ~48 concurrent callers (96 when this was measured) (a table walk over fewer than ~50 polynomials leaves lanes idle
                // and pays its reduction tree in full, so two half-size walks take 1.3 times one full-size walk: measured 44.6 k/s
                // against 38.2 k/s with 64 callers), a second and third one beyond, where a batch is large enough to walk
                // efficiently and the host side of a batch (hundreds of wake-ups and 128 KiB row copies) is worth overlapping
                // (256 callers: 55.8 k/s with three against 44.0 k/s with one).
                exec_limit_ = (int)(1 + peak_ / KZG_COALESCE_CALLERS_PER_BATCH);
                if (exec_limit_ > max_exec_) exec_limit_ = max_exec_;
                uint64_t target = (peak_ + exec_limit_ - 1) / exec_limit_;
                if (target > max_batch_) target = max_batch_;
                if (window_us_ > 0 && b.rows.size() < target) {
                    gathering_ = bi; b.target = target;
                    const auto g0 = std::chrono::steady_clock::now();
                    // the window grows with the work it precedes: 5 % of the recent execution time of a batch (a 36 ms FK20 batch can
                    // afford 1.8 ms for callers that need a millisecond to come back -- 32 Python threads re-entered over ~2 ms and ran
                    // as two alternating half batches, 75 ms per round instead of 40), never less than `window_us_`
                    const long grown = (long)(exec_ema_s_ * 0.05 * 1e6);
                    const auto deadline = g0 + std::chrono::microseconds(grown > window_us_ ? grown : window_us_);
                    while (b.rows.size() < target)
                        if (cv_leader_.wait_until(lk, deadline) == std::cv_status::timeout) break;
                    gathering_ = -1;
                    gather_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
                }
                // close this buffer, open a free one
                b.state.store(coalesce_buf::CLOSED, std::memory_order_relaxed);
                open_ = -1;
                for (int k = 1; k < NBUF; k++) {
                    const int o = (bi + k) % NBUF;
                    if (bufs_[o].state.load(std::memory_order_relaxed) == coalesce_buf::FREE_OPEN && bufs_[o].outstanding.load() == 0) { open_ = o; break; }
                }
                if (open_ >= 0) cv_reserve_.notify_all();                  // callers that found this buffer full
                const auto r0 = std::chrono::steady_clock::now();
                while (b.ready < b.rows.size()) cv_leader_.wait(lk);       // every reserved row filled
                ready_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
                const uint64_t batch = b.rows.size();
                for (uint64_t i = 0; i < batch; i++) b.h_meta[i] = b.rows[i];
                lk.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                int st;
                try { st = exec(b, batch); } catch (...) { st = alloc_error_status; }   // (std::bad_alloc in the executor must not strand the sleepers)
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                lk.lock();
                batches_++; requests_ += batch; exec_s_ += dt;
                exec_ema_s_ = exec_ema_s_ == 0 ? dt : 0.75 * exec_ema_s_ + 0.25 * dt;
                b.status = st;
                b.state.store(coalesce_buf::DRAINING, std::memory_order_release);
                b.bump_all();
                executing_--;
                coalesce_buf *next = (open_ >= 0 && open_ != bi) ? &bufs_[open_] : nullptr;
                if (next) next->bump_all();
                lk.unlock(); locked = false;
                if (next)                                                  // one caller of the accumulating batch becomes its leader
                    for (auto &w : next->epoch) if (futex_wake_u32(&w.v, 1) > 0) break;
                for (auto &w : b.epoch) futex_wake_u32(&w.v, 1);           // this batch's callers: one per word, each wakes its word's rest
                break;
            }
            std::atomic<uint32_t> *word = &b.epoch[row % coalesce_buf::NWAKE].v;
            const uint32_t seen = word->load(std::memory_order_relaxed);   // under the mutex: every later event bumps it
            lk.unlock();
            futex_wait_u32(word, seen);
            if (b.state.load(std::memory_order_acquire) == coalesce_buf::DRAINING) {   // the common wake-up: done; leave without the mutex
                futex_wake_u32(word, INT_MAX);
                locked = false;
                break;
            }
            lk.lock();
        }
        const int st = b.status;
        inside_.fetch_sub(1, std::memory_order_relaxed);
        if (locked) lk.unlock();
        if (st == 0) memcpy(out, b.h_out + row * out_row_, out_bytes);
        if (b.outstanding.fetch_sub(1) == 1) {                              // last one out recycles the buffer (only it re-takes the lock)
            lk.lock();
            b.rows.clear(); b.ready = 0; b.state.store(coalesce_buf::FREE_OPEN, std::memory_order_relaxed);
            if (open_ < 0) { open_ = bi; cv_reserve_.notify_all(); }
        }
        return st;
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_reserve_, cv_leader_;
    coalesce_buf bufs_[NBUF];
    int open_ = 0;               // buffer accepting reservations, -1 while all are busy
    int executing_ = 0;          // batches on the device
    int max_exec_ = MAX_EXEC;    // upper bound (KZG_HIP_COALESCE_EXEC)
    int exec_limit_ = 1;         // batches allowed in flight right now: 1 .. max_exec_ by the number of concurrent callers
    std::atomic<uint64_t> inside_{0};   // callers currently inside submit()
    uint64_t inside_max_ = 0;    // its maximum since the last batch was formed (under the mutex)
    uint64_t peak_ = 0;          // decaying maximum of inside_: the concurrency the gather targets are derived from
    int gathering_ = -1;         // buffer whose elected leader is still waiting for stragglers
    long window_us_ = 150;       // upper bound of that wait (KZG_HIP_COALESCE_US; 0 disables)
    uint64_t batches_ = 0, requests_ = 0;   // statistics (KZG_HIP_COALESCE_STATS=1 prints them when the handle is freed)
    double exec_s_ = 0, gather_s_ = 0, ready_s_ = 0;
    double exec_ema_s_ = 0;      // recent execution time of a batch (the gather window scales with it)
    int device_;
    size_t in_row_, out_row_;
    uint64_t max_batch_;
};

}  // namespace kzg
