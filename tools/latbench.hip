// latbench.hip -- what a LIMB-PARALLEL F_p product would buy the latency paths (a lone LinCombG1's Horner, the finish of a lone commitment: chains of
// dependent products on ONE wavefront).  Two forms of the chain x <- x * y (lazy 13 x 30-bit limbs, mont_core30 of field.hpp), one wavefront per SIMD:
//   lane : one lane per chain (what the library does today)
//   quad : the four lanes of a quad share ONE product: lane q owns the columns 4 q .. 4 q + 3 of the accumulator (16 columns, the upper three of A and p
//          are zero), the multiplier y is replicated, the Montgomery factor m of a round is computed by lane 0 and broadcast (DPP quad_perm), the column
//          shift at the end of a round pulls one 64-bit column from the neighbour lane, the final carry sweep runs lane-local and hands its carry on.
// Prints ns per product for both and checks the quad result against the lane result.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/latbench tools/latbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../go-kzg_amd/csrc/field.hpp"
#include "../go-kzg_amd/csrc/coop_inv.hpp"
using namespace kzg;

template <int CTRL> __device__ __forceinline__ uint32_t dppc(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ uint64_t dppc64(uint64_t v) { return (uint64_t)dppc<CTRL>((uint32_t)v) | ((uint64_t)dppc<CTRL>((uint32_t)(v >> 32)) << 32); }
constexpr int Q_BCAST0 = 0x00, Q_DOWN = 0xF9 /* lane q <- q + 1 */, Q_UP = 0x90 /* lane q <- q - 1 */;

// one lane per chain
__global__ __launch_bounds__(64) void k_chain_lane(uint32_t *io, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t A[13], B[13], r[13];
    for (int i = 0; i < 13; i++) { A[i] = io[t * 32 + i] & 0x3fffffffu; B[i] = io[t * 32 + 16 + i] & 0x3fffffffu; }
    A[12] &= 0xfffffu; B[12] &= 0xfffffu;
    for (int it = 0; it < iters; it++) {
        mont_core30(r, A, B);
#pragma unroll
        for (int i = 0; i < 13; i++) A[i] = r[i];
    }
    for (int i = 0; i < 13; i++) io[t * 32 + i] = A[i];
}
// four lanes per chain: lane q of a quad owns columns 4 q + s, s = 0..3
__global__ __launch_bounds__(64) void k_chain_quad(uint32_t *io, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, q = t & 3u, chain = t >> 2;
    uint32_t Aq[4], Pq[4], B[13];
    for (int s = 0; s < 4; s++) {
        const uint32_t j = 4 * q + s;
        Aq[s] = j < 13 ? (io[chain * 32 + j] & (j == 12 ? 0xfffffu : 0x3fffffffu)) : 0u;
        Pq[s] = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) if (k == (int)j) Pq[s] = FpP::p30(k);
    }
    for (int i = 0; i < 13; i++) B[i] = io[chain * 32 + 16 + i] & (i == 12 ? 0xfffffu : 0x3fffffffu);
    for (int it = 0; it < iters; it++) {
        uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 13; i++) {
#pragma unroll
            for (int s = 0; s < 4; s++) acc[s] += (uint64_t)Aq[s] * B[i];
            const uint32_t m0 = ((uint32_t)acc[0] * FpP::INV30) & 0x3fffffffu;
            const uint32_t m = dppc<Q_BCAST0>(m0);
#pragma unroll
            for (int s = 0; s < 4; s++) acc[s] += (uint64_t)m * Pq[s];
            const uint64_t carry = q == 0 ? (acc[0] >> 30) : 0;      // lane 0: the low 30 bits of column 0 are zero by construction of m
            const uint64_t nb = dppc64<Q_DOWN>(acc[0]);              // the neighbour's bottom column becomes this lane's top column
            acc[0] = acc[1] + carry; acc[1] = acc[2]; acc[2] = acc[3]; acc[3] = q == 3 ? 0 : nb;
            if (i == 6) {                                            // keeps every column below 2^64: lane-local sweep, carry handed to the neighbour (not propagated)
#pragma unroll
                for (int s = 0; s < 3; s++) { acc[s + 1] += acc[s] >> 30; acc[s] &= 0x3fffffffull; }
                const uint64_t up = dppc64<Q_UP>(acc[3] >> 30);
                acc[3] &= 0x3fffffffull;
                if (q != 0) acc[0] += up;
            }
        }
        // final normalisation: local sweep, carry to the neighbour, local sweep again (a second hand-over is needed only if a limb overflows twice: the
        // incoming carry is < 2^35 and the limbs are < 2^30 after the first sweep, so one more pass of hand-overs settles it)
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            uint64_t c = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) { const uint64_t x = acc[s] + c; acc[s] = x & 0x3fffffffull; c = x >> 30; }
            const uint64_t up = dppc64<Q_UP>(c);
            if (q != 0) acc[0] += up;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) Aq[s] = (uint32_t)acc[s];
    }
    for (int s = 0; s < 4; s++) if (4 * q + s < 13) io[chain * 32 + 4 * q + s] = Aq[s];
}

// ---- round 6: the F_p inversion, one lane (inv<FpP>, Pornin's binary GCD) against the wave-cooperative form (coop_inv.hpp) ----
// chains of `iters` dependent inversions x <- inv(x) + 1 (the +1 keeps the chain off the fixed points); one wavefront per workgroup
__global__ __launch_bounds__(64) void k_inv_lane(uint32_t *io, int iters, int all_lanes) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (!all_lanes && threadIdx.x != 0) return;
    fp x;
    for (int i = 0; i < 12; i++) x.l[i] = io[t * 32 + i];
    x.l[11] &= 0x0fffffffu;
    const fp one_ = one<FpP>();
    for (int it = 0; it < iters; it++) x = add(inv<FpP>(x), one_);
    for (int i = 0; i < 12; i++) io[t * 32 + i] = x.l[i];
}
__global__ __launch_bounds__(64) void k_inv_coop(uint32_t *io, int iters) {
    const uint32_t t = blockIdx.x * 64;                 // lane 0's operand
    fp x;
    for (int i = 0; i < 12; i++) x.l[i] = io[t * 32 + i];
    x.l[11] &= 0x0fffffffu;
    const fp one_ = one<FpP>();
    for (int it = 0; it < iters; it++) x = add(wave_inv_fp(x, 0), one_);
    if (threadIdx.x == 0) for (int i = 0; i < 12; i++) io[t * 32 + i] = x.l[i];
}
static void inversion_rows(uint32_t *d, const std::vector<uint32_t> &h, size_t words, int blocks) {
    std::vector<uint32_t> a(words), b(words);
    auto run = [&](int which, int nblocks, int iters, std::vector<uint32_t> *out) {
        hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (which == 0) hipLaunchKernelGGL(k_inv_lane, dim3(nblocks), dim3(64), 0, 0, d, iters, 0);
        else if (which == 1) hipLaunchKernelGGL(k_inv_coop, dim3(nblocks), dim3(64), 0, 0, d, iters);
        else hipLaunchKernelGGL(k_inv_lane, dim3(nblocks), dim3(64), 0, 0, d, iters, 1);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (out) hipMemcpy(out->data(), d, words * 4, hipMemcpyDeviceToHost);
        return (double)ms * 1e-3;
    };
    run(0, blocks, 3, &a); run(1, blocks, 3, &b);
    size_t bad = 0;
    for (int c = 0; c < blocks; c++) for (int i = 0; i < 12; i++) if (a[(size_t)c * 64 * 32 + i] != b[(size_t)c * 64 * 32 + i]) { bad++; break; }
    printf("cooperative inversion == lane inversion on %d chains of 3 inversions: %s (%zu mismatches)\n", blocks, bad ? "NO" : "yes", bad);
    const int it = 64;
    for (int rep = 0; rep < 2; rep++) {
        const double l1 = run(0, 1, it, nullptr), c1 = run(1, 1, it, nullptr), l64 = run(2, 1, it, nullptr);
        printf("a single wavefront on the chip, chains of %d dependent inversions: one lane %.1f us per inversion, all 64 lanes (64 inversions) %.1f us, wave-cooperative %.1f us (x%.2f)\n",
               it, l1 / it * 1e6, l64 / it * 1e6, c1 / it * 1e6, l1 / c1);
        const double lb = run(0, blocks, it, nullptr), cb = run(1, blocks, it, nullptr);
        printf("one wavefront per SIMD (%d wavefronts): one lane %.1f us per inversion, wave-cooperative %.1f us (x%.2f)\n", blocks, lb / it * 1e6, cb / it * 1e6, lb / cb);
    }
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const int blocks = cus * 4;                          // one 64-lane workgroup per SIMD
    const size_t lanes = (size_t)blocks * 64, words = lanes * 32;
    std::vector<uint32_t> h(words), h1(words), h2(words);
    uint64_t st = 88172645463325252ull;
    for (auto &w : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; w = (uint32_t)st; }
    uint32_t *d; hipMalloc(&d, words * 4);
    inversion_rows(d, h, words, blocks);
    auto run = [&](bool quad, int iters, std::vector<uint32_t> *out) {
        hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (quad) hipLaunchKernelGGL(k_chain_quad, dim3(blocks), dim3(64), 0, 0, d, iters); else hipLaunchKernelGGL(k_chain_lane, dim3(blocks), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (out) hipMemcpy(out->data(), d, words * 4, hipMemcpyDeviceToHost);
        return (double)ms * 1e-3;
    };
    // correctness: chain c of the quad kernel == lane c of the lane kernel started from the same (A, B) -- the quad kernel reads chain c's operands from row c
    run(false, 3, &h1); run(true, 3, &h2);
    size_t bad = 0;
    for (size_t c = 0; c < lanes / 4; c++) for (int i = 0; i < 13; i++) if (h1[c * 32 + i] != h2[c * 32 + i]) { bad++; break; }
    printf("quad product == lane product on %zu chains of 3 products: %s (%zu mismatches)\n", lanes / 4, bad ? "NO" : "yes", bad);
    run(false, 64, nullptr); run(true, 64, nullptr);
    for (int rep = 0; rep < 2; rep++) {
        const int it = 4096;
        const double tl = run(false, it, nullptr), tq = run(true, it, nullptr);
        printf("one wavefront per SIMD, chains of %d dependent products: lane %.1f ns per product, quad %.1f ns per product (x%.2f)\n", it, tl / it * 1e9, tq / it * 1e9, tl / tq);
    }
    // one wavefront on the whole chip (an idle machine around a lone call)
    {
        const int it = 4096;
        hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k_chain_lane, dim3(1), dim3(64), 0, 0, d, it); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double tl = ms * 1e-3;
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k_chain_quad, dim3(1), dim3(64), 0, 0, d, it); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("a single wavefront on the chip: lane %.1f ns per product, quad %.1f ns per product (x%.2f)\n", tl / it * 1e9, ms * 1e-3 / it * 1e9, tl / (ms * 1e-3));
    }
    return 0;
}
