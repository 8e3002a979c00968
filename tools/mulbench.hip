// mulbench.hip -- F_p multiplier variants vs occupancy (waves/SIMD forced through dynamic LDS per workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../go-kzg_amd/csrc/field.hpp"
using namespace kzg;

template <int CH, int V> __global__ __launch_bounds__(256) void k_mul(fp *io, int iters) {
    extern __shared__ uint32_t dummy[];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp x[CH], y = io[t ^ 1];
    for (int c = 0; c < CH; c++) { x[c] = io[t]; x[c].l[0] += c; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = V ? mont_mul_fp30(x[c], y) : mont_mul_inl<FpP>(x[c], y);
    }
    fp s = x[0];
    for (int c = 1; c < CH; c++) s = add(s, x[c]);
    io[t] = s;
    if (iters < 0) dummy[threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_mad_dep(uint32_t *out, uint32_t seed, int iters) {
    extern __shared__ uint32_t dummy[];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = seed + t; uint32_t a = seed * 7 + t, b = seed * 13 + t;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    }
    out[t] = (uint32_t)x ^ (uint32_t)(x >> 32);
    if (iters < 0) dummy[threadIdx.x] = t;
}
template <class K, class... A> double time_kernel(K k, dim3 g, dim3 b, size_t sh, A... args) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, sh, 0, args...); hipDeviceSynchronize();
    hipEventRecord(e0, 0); hipLaunchKernelGGL(k, g, b, sh, 0, args...); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e-3;
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    int cus = pr.multiProcessorCount;
    size_t n = (size_t)cus * 8 * 256; fp *io; hipMalloc(&io, n * sizeof(fp)); hipMemset(io, 0x11, n * sizeof(fp));
    uint32_t *o32; hipMalloc(&o32, n * 4);
    int wps[4] = {1, 2, 4, 8};
    for (int w = 0; w < 4; w++) {
        int blocks_per_cu = wps[w];                 // 256 threads = 4 waves = 1 wave per SIMD per block
        size_t sh = blocks_per_cu == 8 ? 0 : (160 * 1024) / blocks_per_cu - 1024;
        int blocks = cus * blocks_per_cu; int it = 128;
        double s1 = time_kernel(k_mul<1, 0>, dim3(blocks), dim3(256), sh, io, it);
        double s2 = time_kernel(k_mul<1, 1>, dim3(blocks), dim3(256), sh, io, it);
        double s4 = time_kernel(k_mul<2, 1>, dim3(blocks), dim3(256), sh, io, it);
        double sd = time_kernel(k_mad_dep, dim3(blocks), dim3(256), sh, o32, 1u, 256);
        double lanes = (double)blocks * 256;
        printf("waves/SIMD %d: CIOS32 %6.2f G/s | fp30 1-chain %6.2f G/s  2-chain %6.2f G/s | dependent mad chain: %5.2f cycles/mad/wave\n", wps[w],
               lanes * it / s1 * 1e-9, lanes * it / s2 * 1e-9, lanes * it * 2 / s4 * 1e-9, sd * 2.4e9 / (256.0 * 16) );
    }
    return 0;
}
