// r16_ab.hip -- stand-alone A/B harness for the 256-lane form of the 4096-point F_r transform (fr16.hpp; see its header for why it is not in the library).
//   tools/ab_fr_r16/build.sh && tools/ab_fr_r16/r16_ab [batch] [reps]
// The same random rows go through k_fr_fft4096_r16 (here) and through the library's kzg_hip_fft_fr_batch_dev (k_fr_fft4096_r4); results are compared word for word,
// both are timed with HIP events.  Exit code 0 = bit-exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fr16.hpp"
#include "../../include/kzg_hip.h"

using namespace kzg;
__device__ __forceinline__ uint32_t bitrev32(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

// 256 lanes x 16 register-resident values (fr16.hpp): 72 KiB of LDS, two workgroups per CU.
template <bool SCALE>
__global__ __launch_bounds__(256, 2) void k_fr_fft4096_r16(const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, const uint32_t *__restrict__ tw,
                                                           const fr *scale, uint32_t rows_log) {
    extern __shared__ uint32_t smem[];
    const uint32_t t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t rows = 1u << rows_log, row = blockIdx.x & (rows - 1);
    const fr *src = in + (uint64_t)(blockIdx.x >> rows_log) * in_stride;
    fr *dst = out + (uint64_t)blockIdx.x * fr4::N;
    frl a[16], b[16];
    {
        fr x[16];
        fr16::load(t, src, n_in, rows, rows_log ? bitrev32(row, rows_log) : 0u, x);
        fr16::pass_a(x, b, tw);
    }
    // A -> B: lane (block u, registers k) -> lane (g, j, registers k')
    const uint32_t u = fr16::bitrev8(fr16::lane_a_nat(t));
    uint32_t g, j;
    fr16::lane_b(t, g, j);
    fr16::stage_put<1>(smem, b, 16u * u, 1u, 0u ^ (w & 1u));
    __syncthreads();
    fr16::stage_get<1>(smem, a, 256u * g + j, 16u, 0u ^ (w >> 1));
    __syncthreads();
    fr16::stage_put<1>(smem, b, 16u * u, 1u, 1u ^ (w & 1u));
    __syncthreads();
    fr16::stage_get<1>(smem, a, 256u * g + j, 16u, 1u ^ (w >> 1));
    fr16::pass_b(a, j, tw);
    __syncthreads();                                       // every lane has read its last stage before the area is written again
    // B -> C: lane (g, j, registers k') -> lane t = 16 k' + j, registers g
    fr16::stage_put<2>(smem, a, 256u * g + j, 16u, 0u ^ (w & 1u));
    __syncthreads();
    fr16::stage_get<2>(smem, b, t, 256u, 0u ^ (w >> 1));
    __syncthreads();
    fr16::stage_put<2>(smem, a, 256u * g + j, 16u, 1u ^ (w & 1u));
    __syncthreads();
    fr16::stage_get<2>(smem, b, t, 256u, 1u ^ (w >> 1));
    fr16::pass_c(b, t, tw);
    frl sc = frl_zero();
    if (SCALE) sc = frl_const_from_kilic(*scale);
    fr16::store<SCALE>(t, b, sc, dst);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const uint64_t batch = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    kzg_hip_fft *fs = nullptr;
    if (kzg_hip_fft_settings_new(0, 12, &fs)) { fprintf(stderr, "no gfx950 device: %s\n", kzg_hip_last_error()); return 2; }
    std::vector<fr> roots(4097);
    if (kzg_hip_fft_roots(fs, 0, roots.data())) return 2;
    std::vector<uint32_t> tw(fr4::TW_WORDS);
    fr4::build_twiddles(roots.data(), 4096, tw.data());
    std::vector<fr> h((size_t)batch * 4096);
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (auto &v : h) {          // any 255-bit pattern below r is a valid Montgomery image
        for (int i = 0; i < 8; i++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v.l[i] = (uint32_t)z; }
        v.l[7] &= 0x3fffffffu;
    }
    fr *d_in, *d_a, *d_b; uint32_t *d_tw;
    CK(hipMalloc(&d_in, h.size() * sizeof(fr))); CK(hipMalloc(&d_a, h.size() * sizeof(fr))); CK(hipMalloc(&d_b, h.size() * sizeof(fr))); CK(hipMalloc(&d_tw, tw.size() * 4));
    CK(hipMemcpy(d_in, h.data(), h.size() * sizeof(fr), hipMemcpyHostToDevice)); CK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft4096_r16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, fr16::LDS_BYTES));
    auto r16 = [&] { hipLaunchKernelGGL(k_fr_fft4096_r16<false>, dim3((uint32_t)batch), dim3(fr16::LANES), fr16::LDS_BYTES, s, d_in, (uint64_t)4096, (uint64_t)4096, d_a, d_tw, (const fr *)nullptr, 0u); };
    auto r4 = [&] { return kzg_hip_fft_fr_batch_dev(fs, d_in, 4096, batch, 0, d_b, s); };
    r16(); if (r4()) { fprintf(stderr, "library: %s\n", kzg_hip_last_error()); return 2; }
    CK(hipStreamSynchronize(s));
    std::vector<fr> a(h.size()), b(h.size());
    CK(hipMemcpy(a.data(), d_a, a.size() * sizeof(fr), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, b.size() * sizeof(fr), hipMemcpyDeviceToHost));
    const bool same = !memcmp(a.data(), b.data(), a.size() * sizeof(fr));
    float ms16 = 0, ms4 = 0;
    CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) r16(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms16, e0, e1));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) r4(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms4, e0, e1));
    printf("batch %llu x %d: r16 (256 lanes x 16 values) %.3f ms per launch = %.2f M FFT/s;  r4 (library) %.3f ms = %.2f M FFT/s;  r16 / r4 time %.2f;  bit-exact: %s\n",
           (unsigned long long)batch, reps, ms16 / reps, batch * reps / (ms16 * 1e-3) * 1e-6, ms4 / reps, batch * reps / (ms4 * 1e-3) * 1e-6, ms16 / ms4, same ? "yes" : "NO");
    kzg_hip_fft_settings_free(fs);
    return same ? 0 : 1;
}
