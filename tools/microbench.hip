// microbench.hip -- measures the gfx950 issue rates that bound the F_p / F_r multiplier (no figure for the
// integer-multiply rate is in the local guides: SURVEY.md 8d says calibrate).  Prints Gops/s per instruction
// and the derived F_p products/s of the library's multiplier.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../go-kzg_amd/csrc/field.hpp"
#include "../go-kzg_amd/csrc/g1.hpp"
using namespace kzg;

#define ITERS 4096
#define CHAINS 8

#define DEF_KERNEL(NAME, DECL, BODY)                                              \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {       \
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;                           \
    DECL                                                                          \
    for (int i = 0; i < ITERS; i++) { BODY }                                      \
    uint32_t acc = 0;                                                             \
    for (int c = 0; c < CHAINS; c++) acc ^= (uint32_t)x[c] ^ (uint32_t)(x[c] >> 16); \
    out[t] = acc;                                                                 \
}
#define DECL64 uint64_t x[CHAINS]; uint32_t a = seed + t, b = seed * 3 + t; for (int c = 0; c < CHAINS; c++) x[c] = seed + c + t;
#define DECL32 uint32_t x[CHAINS]; uint32_t a = seed + t, b = seed * 3 + t; for (int c = 0; c < CHAINS; c++) x[c] = seed + c + t;

DEF_KERNEL(k_mad_u64_u32, DECL64, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc");)
DEF_KERNEL(k_mul_lo_u32, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));)
DEF_KERNEL(k_mul_hi_u32, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));)
DEF_KERNEL(k_mad_u32_u24, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));)
DEF_KERNEL(k_add_u32, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));)
DEF_KERNEL(k_addc_co, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc");)
DEF_KERNEL(k_lshl_add_u64, DECL64, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[c]) : "v"((uint64_t)a));)
DEF_KERNEL(k_mov_b32, DECL32, _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(a));)

__global__ __launch_bounds__(256) void k_fma_f64(uint32_t *out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    double x[CHAINS], a = 1.0000001 + seed * 1e-9, b = 1e-9 * t;
    for (int c = 0; c < CHAINS; c++) x[c] = c + t;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    double s = 0; for (int c = 0; c < CHAINS; c++) s += x[c];
    out[t] = (uint32_t)s;
}
__global__ __launch_bounds__(256) void k_fp_mul(fp *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp x = io[t], y = io[t ^ 1];
    for (int i = 0; i < iters; i++) { x = mul(x, y); y = mul(y, x); }
    io[t] = add(x, y);
}
__global__ __launch_bounds__(256) void k_fr_mul(fr *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fr x = io[t], y = io[t ^ 1];
    for (int i = 0; i < iters; i++) { x = mul(x, y); y = mul(y, x); }
    io[t] = add(x, y);
}
__global__ __launch_bounds__(128) void k_g1_add_chain(g1j *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    g1j x = io[t], y = io[t ^ 1];
#pragma nounroll
    for (int i = 0; i < iters; i++) { x = g1_add(x, y); }
    io[t] = x;
}
__global__ __launch_bounds__(128) void k_g1_dbl_chain(g1j *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    g1j x = io[t];
#pragma nounroll
    for (int i = 0; i < iters; i++) { x = g1_dbl(x); }
    io[t] = x;
}

template <class K, class... A> double time_kernel(K k, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("device %s  CUs %d  clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    const int blocks = pr.multiProcessorCount * 8, threads = 256;
    uint32_t *d; hipMalloc(&d, (size_t)blocks * threads * 4);
    double nops = (double)blocks * threads * ITERS * CHAINS;
#define RUN(K) { double s = time_kernel(K, dim3(blocks), dim3(threads), d, 12345u); printf("%-16s %8.1f Gop/s (lane-ops)   %6.2f cyc/wave-instr/SIMD @2.4GHz\n", #K, nops / s * 1e-9, 2.4e9 * (pr.multiProcessorCount * 4.0) * 64.0 / (nops / s)); }
    RUN(k_mad_u64_u32) RUN(k_mul_lo_u32) RUN(k_mul_hi_u32) RUN(k_mad_u32_u24) RUN(k_add_u32) RUN(k_addc_co) RUN(k_lshl_add_u64) RUN(k_mov_b32) RUN(k_fma_f64)
    printf("(k_addc_co issues 2 instructions per op)\n");
    // field / group throughput of the library's own arithmetic
    {
        size_t n = (size_t)blocks * threads; fp *io; hipMalloc(&io, n * sizeof(fp)); hipMemset(io, 0x11, n * sizeof(fp));
        int it = 256; double s = time_kernel(k_fp_mul, dim3(blocks), dim3(threads), io, it);
        printf("fp_mul   %8.2f G mul/s\n", (double)n * it * 2 / s * 1e-9);
        fr *io2 = (fr *)io; s = time_kernel(k_fr_mul, dim3(blocks), dim3(threads), io2, it);
        printf("fr_mul   %8.2f G mul/s\n", (double)n * it * 2 / s * 1e-9);
        hipFree(io);
    }
    {
        size_t n = (size_t)pr.multiProcessorCount * 4 * 128; g1j *io; hipMalloc(&io, n * sizeof(g1j));
        // points: use the generator with slightly different z to avoid the equal-input fast paths: fill x,y,z with junk field elements
        hipMemset(io, 0x07, n * sizeof(g1j));
        int it = 64; double s = time_kernel(k_g1_add_chain, dim3(n / 128), dim3(128), io, it);
        printf("g1_add   %8.2f M add/s  (junk coordinates: timing only)\n", (double)n * it / s * 1e-6);
        s = time_kernel(k_g1_dbl_chain, dim3(n / 128), dim3(128), io, it);
        printf("g1_dbl   %8.2f M dbl/s\n", (double)n * it / s * 1e-6);
        hipFree(io);
    }
    return 0;
}
