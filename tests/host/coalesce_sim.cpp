// coalesce_sim.cpp -- the request coalescer (go-kzg_amd/csrc/coalesce.hpp) against a SIMULATED device, on the CPU: T caller threads make blocking
// one-row calls; the executor models a GPU that needs `a` microseconds of latency per batch plus `b` microseconds per row, concurrent batches
// sharing the per-row throughput.  Checks that every caller gets ITS result (rows are tagged), that ragged metadata arrives with the right row,
// that nothing deadlocks when callers come and go, and prints calls per second -- the protocol (row tickets, leader = row 0, device slots,
// futex sleeps) has no GPU in it, so its races are found here, under ThreadSanitizer if wanted (g++ -fsanitize=thread).
//   usage: coalesce_sim <threads> <calls per thread> [a_us b_us]      exit code 0 = every result correct
#define KZG_COALESCE_SIM 1
#include "../../go-kzg_amd/csrc/coalesce.hpp"
#include <thread>
#include <cstdint>

using namespace kzg;

static std::atomic<int> g_running{0};
static std::atomic<uint64_t> g_batches{0}, g_rows{0}, g_maxbatch{0};

int main(int argc, char **argv) {
    const unsigned T = argc > 1 ? (unsigned)atoi(argv[1]) : 64, calls = argc > 2 ? (unsigned)atoi(argv[2]) : 200;
    const double a_us = argc > 3 ? atof(argv[3]) : 280.0, b_us = argc > 4 ? atof(argv[4]) : 9.5;
    const size_t in_row = 128 << 10, out_row = 144;
    coalescer co(0, in_row, out_row, 256);
    auto exec = [&](coalesce_buf &b, uint64_t batch) -> int {
        g_batches++; g_rows += batch;
        { uint64_t m = g_maxbatch.load(); while (batch > m && !g_maxbatch.compare_exchange_weak(m, batch)) {} }
        g_running++;
        double work = b_us * (double)batch;                       // shared throughput: k concurrent batches progress at 1 / k each
        auto last = std::chrono::steady_clock::now();
        while (work > 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            auto now = std::chrono::steady_clock::now();
            work -= std::chrono::duration<double, std::micro>(now - last).count() / (double)g_running.load();
            last = now;
        }
        g_running--;
        std::this_thread::sleep_for(std::chrono::microseconds((long)a_us));
        for (uint64_t i = 0; i < batch; i++) {                    // result of row i: its tag, its n, its arg, and the last word of its input
            uint64_t tag, tail;
            memcpy(&tag, b.h_in + i * in_row, 8);
            memcpy(&tail, b.h_in + i * in_row + b.h_meta[i].n * 8 - 8, 8);
            uint64_t out[4] = {tag * 0x9e3779b97f4a7c15ull, b.h_meta[i].n, b.h_meta[i].arg, tail};
            memcpy(b.h_out + i * out_row, out, sizeof out);
        }
        return 0;
    };
    std::atomic<uint64_t> bad{0};
    std::vector<std::thread> ts;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned t = 0; t < T; t++)
        ts.emplace_back([&, t] {
            std::vector<uint64_t> in(in_row / 8);
            const unsigned mine = calls - (t % 3 == 2 ? calls / 2 : 0);     // a third of the callers leave early
            for (unsigned c = 0; c < mine; c++) {
                const uint64_t tag = ((uint64_t)t << 32) | c, n = 1 + (tag * 2654435761u) % (in_row / 8);   // ragged lengths
                in[0] = tag; in[n - 1] = ~tag; if (n == 1) in[0] = tag;
                uint64_t out[4] = {0, 0, 0, 0};
                int st = co.submit(in.data(), n * 8, n, tag ^ 0x55, out, sizeof out, exec, 8);
                const uint64_t want_tail = n == 1 ? tag : ~tag;
                if (st || out[0] != tag * 0x9e3779b97f4a7c15ull || out[1] != n || out[2] != (tag ^ 0x55) || out[3] != want_tail) bad++;
            }
        });
    for (auto &th : ts) th.join();
    if (argc > 5) {   // "lone-after-burst": the burst above is followed, after an idle gap, by one caller: it must not pay the burst's gather window
        std::this_thread::sleep_for(std::chrono::milliseconds(atoi(argv[5])));
        std::vector<uint64_t> in(in_row / 8, 7);
        const auto l0 = std::chrono::steady_clock::now();
        for (int c = 0; c < 20; c++) { uint64_t out[4]; in[0] = 1000 + c; if (co.submit(in.data(), 64, 8, 0, out, sizeof out, exec, 8) || out[0] != (1000ull + c) * 0x9e3779b97f4a7c15ull) bad++; }
        const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - l0).count() / 20;
        printf("lone caller after the burst: %.0f us per call (device model: %.0f us)\n", per, a_us + b_us);
        g_rows -= 20;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t total = 0;
    for (unsigned t = 0; t < T; t++) total += calls - (t % 3 == 2 ? calls / 2 : 0);
    printf("threads %u: %llu calls in %.3f s = %.0f calls/s, %llu batches (avg %.1f rows, max %llu), wrong results: %llu\n", T, (unsigned long long)total, secs, total / secs,
           (unsigned long long)g_batches.load(), (double)g_rows.load() / (double)(g_batches.load() ? g_batches.load() : 1), (unsigned long long)g_maxbatch.load(), (unsigned long long)bad.load());
    return bad.load() || g_rows.load() != total ? 1 : 0;
}
