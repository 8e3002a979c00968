// the decimation-in-frequency stage with four / two lanes per butterfly (g1_coop_kernels.hpp)
#define KZG_MULQ_NOINLINE 1
#include "g1_coop_kernels.hpp"
namespace kzg {
void launch_g1_stage_coop_dif(hipStream_t s, int lanes, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W, uint64_t total) {
    launch_stage_coop<true>(s, lanes, data, n, batch, m, roots, wnaf, W, total);
}
}  // namespace kzg
