// the direct pass of a lone G1 transform with four / two lanes per (output, term) (g1_coop_kernels.hpp)
#define KZG_MULQ_NOINLINE 1
#include "g1_coop_kernels.hpp"
namespace kzg {
void launch_g1_direct_coop(hipStream_t s, int lanes, uint32_t wgs, size_t pad_lds, const g1j *src, uint64_t src_stride, uint64_t src_valid, g1j *dst, uint32_t logn, uint32_t logR,
                           uint64_t Ns, const fr *roots, uint64_t W, const fr *sc, uint64_t total, uint32_t logT, uint32_t logU) {
    if (lanes == 4) hipLaunchKernelGGL(k_g1_fft_direct_coop<4>, dim3(wgs), dim3(G1_DIRECT_BLOCK), pad_lds, s, src, src_stride, src_valid, dst, logn, logR, Ns, roots, W, sc, total, logT, logU);
    else hipLaunchKernelGGL(k_g1_fft_direct_coop<2>, dim3(wgs), dim3(G1_DIRECT_BLOCK), pad_lds, s, src, src_stride, src_valid, dst, logn, logR, Ns, roots, W, sc, total, logT, logU);
}
}  // namespace kzg
