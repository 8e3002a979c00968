// k_fr.hip -- F_r kernels: batched radix-2 (I)FFT, DAS FFT extension, Toeplitz coefficient gather,
// quotient by (X - x).  Replaces fft_fr.go:30-105, das_extension.go:7-84, fk20_single.go:89-119 and
// poly.go:14-40 of the reference.  Whole transforms of <= 4096 points live in LDS (4096 x 32 B = 128 KiB of
// the CU's 160 KiB), limb-major so that consecutive lanes hit consecutive banks.
#include "internal.hpp"
#include "fr_fft4096.hpp"
#include "fr_das2048.hpp"
#include "coop_inv.hpp"
#include <stdlib.h>
#include <string.h>

namespace kzg {

static constexpr uint32_t FR_TILE_LOG = 12;          // 4096 points per LDS tile
static constexpr uint32_t FR_TILE = 1u << FR_TILE_LOG;

__device__ __forceinline__ uint32_t bitrev32(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

// limb-major LDS view: limb k of element i at s[k * n + i]
struct lds_view {
    uint32_t *s; uint32_t n;
    __device__ __forceinline__ fr get(uint32_t i) const {
        fr v;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = s[k * n + i];
        return v;
    }
    __device__ __forceinline__ void put(uint32_t i, const fr &v) const {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k * n + i] = v.l[k];
    }
};
struct glob_view {
    fr *p;
    __device__ __forceinline__ fr get(uint64_t i) const { return p[i]; }
    __device__ __forceinline__ void put(uint64_t i, const fr &v) const { p[i] = v; }
};

// one DIT butterfly group stage over `cnt` butterflies with half-size m (data already bit-reversed)
template <class V>
__device__ __forceinline__ void fft_butterfly(const V &v, uint64_t bf, uint64_t m, const fr *roots, uint64_t rstride) {
    uint64_t j = bf & (m - 1);
    uint64_t i0 = ((bf - j) << 1) + j, i1 = i0 + m;
    fr x = v.get(i0), y = v.get(i1);
    if (j) y = mul(y, roots[j * rstride]);   // w^0 = 1: the reference multiplies anyway (fft_fr.go:49), same value
    v.put(i0, add(x, y));
    v.put(i1, sub(x, y));
}

// Tile kernel: log2(tn) stages in LDS.  BITREV: tile == whole transform, input read in natural order from `in`
// (zero-padded beyond n_in) and scattered bit-reversed; otherwise the tile is read in place from `data`.
template <bool BITREV>
__global__ __launch_bounds__(1024) void k_fr_fft_tile(const fr *in, uint64_t in_stride, uint64_t n_in, fr *data, uint32_t log_tn,
                                                      uint64_t tiles_per_row, const fr *roots, uint64_t W, const fr *scale) {
    extern __shared__ uint32_t smem[];
    const uint32_t tn = 1u << log_tn, T = blockDim.x, tid = threadIdx.x;
    lds_view v{smem, tn};
    const uint64_t tile = blockIdx.x;
    fr *dst = data + tile * tn;
    if (BITREV) {
        const fr *src = in + tile * in_stride;   // tiles_per_row == 1: tile index == batch index
        for (uint32_t i = tid; i < tn; i += T) {
            fr x = (i < n_in) ? src[i] : zero<FrP>();
            v.put(bitrev32(i, log_tn), x);
        }
    } else {
        for (uint32_t i = tid; i < tn; i += T) v.put(i, dst[i]);
    }
    __syncthreads();
    for (uint32_t m = 1; m < tn; m <<= 1) {
        uint64_t rstride = W / (2ull * m);
        for (uint32_t bf = tid; bf < tn / 2; bf += T) fft_butterfly(v, bf, m, roots, rstride);
        __syncthreads();
    }
    const bool last = (tiles_per_row == 1);
    fr sc;
    if (last && scale) sc = *scale;
    for (uint32_t i = tid; i < tn; i += T) {
        fr x = v.get(i);
        if (last && scale) x = mul(x, sc);
        dst[i] = x;
    }
}

// 4096-point transform in six radix-4 passes on lazy 29-bit limbs (fr_fft4096.hpp): the hot size (scale-12 FFT, the Toeplitz
// coefficient transforms of FK20 at scale 12 and of FK20Multi at scale 16 / chunk 16).  One workgroup per transform; 146 KiB of LDS.
template <bool SCALE>
__global__ __launch_bounds__(1024) void k_fr_fft4096_r4(const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, const uint32_t *__restrict__ tw,
                                                        const fr *scale, uint32_t rows_log) {
    extern __shared__ uint32_t smem[];
    const uint32_t t = threadIdx.x, a = __builtin_amdgcn_readfirstlane(t >> 6), b = t & 63u;
    // rows_log = 0: workgroup w transforms row w.  rows_log = k > 0: the workgroups are the 2^k rows of transforms of 2^k * 4096 points --
    // row a of transform w >> k is its subsequence bitrev_k(a) + 2^k i (what a bit-reversal copy would have put in block a), left in natural
    // order in block a of the output for the upper stages (k_fr_fft_upper)
    const uint32_t rows = 1u << rows_log, row = blockIdx.x & (rows - 1);
    const fr *src = in + (uint64_t)(blockIdx.x >> rows_log) * in_stride;
    fr *dst = out + (uint64_t)blockIdx.x * fr4::N;
    fr4::pass_first(t, src, n_in, smem, tw, rows, rows_log ? bitrev32(row, rows_log) : 0u);
    __syncthreads();
    fr4::pass_lo<4>(a, b, smem, tw);
    __syncthreads();
    fr4::pass_lo<16>(a, b, smem, tw);
    __syncthreads();
    fr4::pass_hi<64>(t, smem, tw);
    __syncthreads();
    fr4::pass_hi<256>(t, smem, tw);
    __syncthreads();
    frl sc = frl_zero();
    if (SCALE) sc = frl_const_from_kilic(*scale);
    fr4::pass_last<SCALE>(t, smem, tw, sc, dst);
}
// 4 .. 2048 points: 4096 / m transforms per workgroup through the first passes of the 4096-point network (fr_fft4096.hpp)
template <int LOGM, bool SCALE>
__global__ __launch_bounds__(1024) void k_fr_fft_small(const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, uint64_t batch, const uint32_t *__restrict__ tw,
                                                       const fr *scale) {
    extern __shared__ uint32_t smem[];
    constexpr uint32_t m = 1u << LOGM, per = fr4::N / m;
    constexpr int A = LOGM / 2;                                            // radix-4 passes (strides 1, 4, ..., 4^(A-1))
    const uint32_t t = threadIdx.x, a = __builtin_amdgcn_readfirstlane(t >> 6), b = t & 63u;
    const uint64_t first = (uint64_t)blockIdx.x * per;
    fr4::pass_first_small<LOGM>(t, in, in_stride, n_in, first, batch, smem, tw);
    __syncthreads();
    if constexpr (A >= 2) { fr4::pass_lo<4>(a, b, smem, tw); __syncthreads(); }
    if constexpr (A >= 3) { fr4::pass_lo<16>(a, b, smem, tw); __syncthreads(); }
    if constexpr (A >= 4) { fr4::pass_hi<64>(t, smem, tw); __syncthreads(); }
    if constexpr (A >= 5) { fr4::pass_hi<256>(t, smem, tw); __syncthreads(); }
    if constexpr (LOGM & 1) { fr4::pass_r2<(1u << (2 * A))>(t, a, b, smem, tw); __syncthreads(); }
    frl sc = frl_zero();
    if (SCALE) sc = frl_const_from_kilic(*scale);
    const uint64_t left = batch - first;                                   // transforms of this workgroup that exist
    fr4::pass_store<SCALE>(t, smem, sc, out + first * m, left >= per ? fr4::N : (uint64_t)left * m);
}
template <int LOGM> static void launch_fr_fft_small(hipStream_t s, const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, uint64_t batch, const uint32_t *tw,
                                                    const fr *scale) {
    const uint32_t per = fr4::N >> LOGM, blocks = (uint32_t)((batch + per - 1) / per);
    if (scale) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft_small<LOGM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fr4::LDS_BYTES);
        hipLaunchKernelGGL((k_fr_fft_small<LOGM, true>), dim3(blocks), dim3(1024), fr4::LDS_BYTES, s, in, in_stride, n_in, out, batch, tw, scale);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft_small<LOGM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, fr4::LDS_BYTES);
        hipLaunchKernelGGL((k_fr_fft_small<LOGM, false>), dim3(blocks), dim3(1024), fr4::LDS_BYTES, s, in, in_stride, n_in, out, batch, tw, scale);
    }
}

// The stages above 4096 of a transform of R * 4096 points (fr4::upper_lane in fr_fft4096.hpp), a lane per k2
template <int LOGR, bool SCALE>
__global__ __launch_bounds__(256) void k_fr_fft_upper(fr *data, const fr *__restrict__ roots_l, uint64_t W, const fr *scale) {
    const uint32_t k2 = blockIdx.x * 256u + threadIdx.x;
    fr4::upper_lane<LOGR, SCALE>(data + (uint64_t)blockIdx.y * ((uint64_t)fr4::N << LOGR), k2, roots_l, W, scale);
}
template <int LOGR> static void launch_fr_fft_upper(hipStream_t s, fr *data, uint64_t batch, const fr *roots_l, uint64_t W, const fr *scale) {
    if (scale) hipLaunchKernelGGL((k_fr_fft_upper<LOGR, true>), dim3(fr4::N / 256, (uint32_t)batch), dim3(256), 0, s, data, roots_l, W, scale);
    else hipLaunchKernelGGL((k_fr_fft_upper<LOGR, false>), dim3(fr4::N / 256, (uint32_t)batch), dim3(256), 0, s, data, roots_l, W, scale);
}

__global__ void k_fr_bitrev_copy(const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, uint32_t logn, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t n = 1ull << logn, b = t >> logn, i = t & (n - 1);
    fr x = (i < n_in) ? in[b * in_stride + i] : zero<FrP>();
    out[b * n + bitrev32((uint32_t)i, logn)] = x;
}
__global__ void k_fr_fft_stage_glob(fr *data, uint32_t logn, uint64_t m, const fr *roots, uint64_t W, uint64_t total, const fr *scale) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t half = 1ull << (logn - 1), b = t / half, bf = t % half;
    glob_view v{data + (b << logn)};
    uint64_t j = bf & (m - 1);
    uint64_t i0 = ((bf - j) << 1) + j, i1 = i0 + m;
    fr x = v.get(i0), y = v.get(i1);
    if (j) y = mul(y, roots[j * (W / (2 * m))]);
    fr o0 = add(x, y), o1 = sub(x, y);
    if (scale) { fr sc = *scale; o0 = mul(o0, sc); o1 = mul(o1, sc); }
    v.put(i0, o0); v.put(i1, o1);
}

static uint32_t ilog2(uint64_t v) { uint32_t r = 0; while ((1ull << r) < v) r++; return r; }
void launch_fr_fft(hipStream_t s, const fr *in, uint64_t in_stride, uint64_t n_in, fr *out, uint64_t n, uint64_t batch, const fr *roots,
                   uint64_t W, const fr *scale, const uint32_t *tw4096, const fr *roots_l) {
    if (n == 0 || batch == 0) return;
    uint32_t logn = ilog2(n);
    static const bool radix2_forced = [] { const char *e = getenv("KZG_HIP_FR_FFT"); return e && !strcmp(e, "radix2"); }();   // A/B and test hook
    if (n == fr4::N && tw4096 && !radix2_forced) {
        prof_begin(s, "fr_fft4096");
        if (scale) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft4096_r4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, fr4::LDS_BYTES);
            hipLaunchKernelGGL(k_fr_fft4096_r4<true>, dim3((uint32_t)batch), dim3(1024), fr4::LDS_BYTES, s, in, in_stride, n_in, out, tw4096, scale, 0u);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft4096_r4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, fr4::LDS_BYTES);
            hipLaunchKernelGGL(k_fr_fft4096_r4<false>, dim3((uint32_t)batch), dim3(1024), fr4::LDS_BYTES, s, in, in_stride, n_in, out, tw4096, scale, 0u);
        }
        prof_end(s, "fr_fft4096");
        return;
    }
    // (from one workgroup per CU on: fewer values are quicker on the one-workgroup-per-transform kernel below, whose small workgroups spread over
    // the whole chip -- 2048 transforms of 32 points: 16 workgroups here, 2048 there)
    static const bool shared_forced = [] { const char *e = getenv("KZG_HIP_FR_FFT"); return e && !strcmp(e, "shared"); }();   // test hook: at every batch size
    if (n >= 4 && n < fr4::N && (shared_forced || batch * n >= 256ull * fr4::N) && tw4096 && !radix2_forced && (batch + (fr4::N / n) - 1) / (fr4::N / n) <= 0x7fffffffull) {
        switch (logn) {
        case 2: launch_fr_fft_small<2>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 3: launch_fr_fft_small<3>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 4: launch_fr_fft_small<4>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 5: launch_fr_fft_small<5>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 6: launch_fr_fft_small<6>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 7: launch_fr_fft_small<7>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 8: launch_fr_fft_small<8>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 9: launch_fr_fft_small<9>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        case 10: launch_fr_fft_small<10>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        default: launch_fr_fft_small<11>(s, in, in_stride, n_in, out, batch, tw4096, scale); break;
        }
        return;
    }
    if (n <= FR_TILE) {
        uint32_t T = (uint32_t)(n / 2 < 64 ? 64 : (n / 2 > 1024 ? 1024 : n / 2));
        size_t sh = (size_t)n * 32;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft_tile<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FR_TILE * 32);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FR_TILE * 32);
        hipLaunchKernelGGL(k_fr_fft_tile<true>, dim3((uint32_t)batch), dim3(T), sh, s, in, in_stride, n_in, out, logn, (uint64_t)1, roots, W, scale);
        return;
    }
    if (n <= 16 * (uint64_t)fr4::N && tw4096 && roots_l && !radix2_forced && batch * (n / fr4::N) <= 0x7fffffffull && batch <= 65535) {
        // 8192 .. 65 536 points: the rows (every R-th element, R = n / 4096) through the LDS-resident 4096-point kernel, then all upper
        // stages in one pass with the row values in registers: two launches, every value read and written twice
        const uint32_t rl = logn - 12;
        prof_begin(s, "fr_fft4096");
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft4096_r4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, fr4::LDS_BYTES);
        hipLaunchKernelGGL(k_fr_fft4096_r4<false>, dim3((uint32_t)(batch << rl)), dim3(1024), fr4::LDS_BYTES, s, in, in_stride, n_in, out, tw4096, (const fr *)nullptr, rl);
        prof_end(s, "fr_fft4096");
        switch (rl) {
        case 1: launch_fr_fft_upper<1>(s, out, batch, roots_l, W, scale); break;
        case 2: launch_fr_fft_upper<2>(s, out, batch, roots_l, W, scale); break;
        case 3: launch_fr_fft_upper<3>(s, out, batch, roots_l, W, scale); break;
        default: launch_fr_fft_upper<4>(s, out, batch, roots_l, W, scale); break;
        }
        return;
    }
    // longer transforms (and KZG_HIP_FR_FFT=radix2): bit-reversal copy, 12 stages per 4096-tile in LDS, remaining stages through global memory
    uint64_t total = n * batch;
    hipLaunchKernelGGL(k_fr_bitrev_copy, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, in, in_stride, n_in, out, logn, total);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_fft_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FR_TILE * 32);
    uint64_t tiles_per_row = n / FR_TILE;
    hipLaunchKernelGGL(k_fr_fft_tile<false>, dim3((uint32_t)(tiles_per_row * batch)), dim3(1024), (size_t)FR_TILE * 32, s, (const fr *)nullptr,
                       (uint64_t)0, (uint64_t)0, out, FR_TILE_LOG, tiles_per_row, roots, W, (const fr *)nullptr);
    uint64_t bfs = total / 2;
    for (uint64_t m = FR_TILE; m < n; m <<= 1) {
        const fr *sc = (m * 2 == n) ? scale : nullptr;
        hipLaunchKernelGGL(k_fr_fft_stage_glob, dim3((uint32_t)((bfs + 255) / 256)), dim3(256), 0, s, out, logn, m, roots, W, bfs, sc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// DAS FFT extension (das_extension.go:7-84).  Unrolled recursion: "down" stages for block lengths n .. 2
// (a0 = a + b, a1 = (a - b) * rev[2 i s]) followed by "up" stages for block lengths 2 .. n
// (x +- y * ex[(1 + 2 i) s]), s = n / block length; then a scale by 1/n.  Table indices are NOT rescaled to
// the input length -- exactly as the reference, which always walks the full-width domain (:38, :59).
// ---------------------------------------------------------------------------------------------------------
template <class V>
__device__ __forceinline__ void das_down(const V &v, uint64_t bf, uint64_t h, uint64_t sp, const fr *reversed) {
    uint64_t i = bf & (h - 1), base = (bf - i) << 1;
    fr a = v.get(base + i), b = v.get(base + h + i);
    v.put(base + i, add(a, b));
    fr d = sub(a, b);
    if (i) d = mul(d, reversed[2 * i * sp]);
    v.put(base + h + i, d);
}
template <class V>
__device__ __forceinline__ void das_up(const V &v, uint64_t bf, uint64_t h, uint64_t sp, const fr *expanded) {
    uint64_t i = bf & (h - 1), base = (bf - i) << 1;
    fr x = v.get(base + i), y = v.get(base + h + i);
    fr yr = mul(y, expanded[(1 + 2 * i) * sp]);
    v.put(base + i, add(x, yr));
    v.put(base + h + i, sub(x, yr));
}
__global__ __launch_bounds__(1024) void k_das_ext_lds(fr *vals, uint32_t logn, const fr *expanded, const fr *reversed, const fr *inv_n) {
    extern __shared__ uint32_t smem[];
    const uint32_t n = 1u << logn, T = blockDim.x, tid = threadIdx.x;
    lds_view v{smem, n};
    fr *row = vals + (uint64_t)blockIdx.x * n;
    for (uint32_t i = tid; i < n; i += T) v.put(i, row[i]);
    __syncthreads();
    for (uint32_t h = n / 2; h >= 1; h >>= 1) {
        uint64_t sp = n / (2 * h);
        for (uint32_t bf = tid; bf < n / 2; bf += T) das_down(v, bf, h, sp, reversed);
        __syncthreads();
    }
    for (uint32_t h = 1; h < n; h <<= 1) {
        uint64_t sp = n / (2 * h);
        for (uint32_t bf = tid; bf < n / 2; bf += T) das_up(v, bf, h, sp, expanded);
        __syncthreads();
    }
    fr sc = *inv_n;
    for (uint32_t i = tid; i < n; i += T) row[i] = mul(v.get(i), sc);
}

// DASFFTExtension of 2048 values in eleven passes on lazy 29-bit limbs (fr_das2048.hpp); one workgroup of 512 lanes per row, two rows per CU
__global__ __launch_bounds__(512, 4) void k_das_ext2048_r4(fr *vals, const uint32_t *__restrict__ tw, const fr *inv_n) {
    extern __shared__ uint32_t smem[];
    const uint32_t t = threadIdx.x, a = __builtin_amdgcn_readfirstlane(t >> 6), b = t & 63u;
    fr *row = vals + (uint64_t)blockIdx.x * das2k::N;
    das2k::pass_down_first(t, row, smem, tw);
    __syncthreads();
    das2k::pass_down_wide<128>(t, smem, tw);
    __syncthreads();
    das2k::pass_down_wide<32>(t, smem, tw);
    __syncthreads();
    das2k::pass_down_narrow<8>(a, b, smem, tw);
    __syncthreads();
    das2k::pass_down_narrow<2>(a, b, smem, tw);
    __syncthreads();
    das2k::pass_middle(a, b, smem, tw);
    __syncthreads();
    das2k::pass_up_narrow<2>(a, b, smem, tw);
    __syncthreads();
    das2k::pass_up_narrow<8>(a, b, smem, tw);
    __syncthreads();
    das2k::pass_up_wide<32>(t, smem, tw);
    __syncthreads();
    das2k::pass_up_wide<128>(t, smem, tw);
    __syncthreads();
    das2k::pass_up_last(t, smem, tw, frl_const_from_kilic(*inv_n), row);
}
__global__ void k_das_stage_glob(fr *vals, uint32_t logn, uint64_t h, int up, const fr *tbl, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t half = 1ull << (logn - 1), b = t / half, bf = t % half, n = 1ull << logn;
    glob_view v{vals + (b << logn)};
    if (up) das_up(v, bf, h, n / (2 * h), tbl); else das_down(v, bf, h, n / (2 * h), tbl);
}
__global__ void k_fr_scale(fr *vals, const fr *sc, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    vals[t] = mul(vals[t], *sc);
}
void launch_das_ext(hipStream_t s, fr *vals, uint64_t n, uint64_t batch, const fr *expanded, const fr *reversed, uint64_t W, const fr *inv_n,
                    const uint32_t *tw2048) {
    (void)W;
    if (!n || !batch) return;
    static const bool radix2_forced = [] { const char *e = getenv("KZG_HIP_FR_FFT"); return e && !strcmp(e, "radix2"); }();
    if (n == das2k::N && tw2048 && !radix2_forced) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_das_ext2048_r4), hipFuncAttributeMaxDynamicSharedMemorySize, das2k::LDS_BYTES);
        prof_begin(s, "das_ext2048");
        hipLaunchKernelGGL(k_das_ext2048_r4, dim3((uint32_t)batch), dim3(das2k::THREADS), das2k::LDS_BYTES, s, vals, tw2048, inv_n);
        prof_end(s, "das_ext2048");
        return;
    }
    uint32_t logn = ilog2(n);
    if (n <= FR_TILE) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_das_ext_lds), hipFuncAttributeMaxDynamicSharedMemorySize, FR_TILE * 32);
        uint32_t T = (uint32_t)(n / 2 < 64 ? 64 : (n / 2 > 1024 ? 1024 : n / 2));
        hipLaunchKernelGGL(k_das_ext_lds, dim3((uint32_t)batch), dim3(T), (size_t)n * 32, s, vals, logn, expanded, reversed, inv_n);
        return;
    }
    uint64_t bfs = n * batch / 2;
    dim3 g((uint32_t)((bfs + 255) / 256)), b(256);
    for (uint64_t h = n / 2; h >= 1; h >>= 1) hipLaunchKernelGGL(k_das_stage_glob, g, b, 0, s, vals, logn, h, 0, reversed, bfs);
    for (uint64_t h = 1; h < n; h <<= 1) hipLaunchKernelGGL(k_das_stage_glob, g, b, 0, s, vals, logn, h, 1, expanded, bfs);
    hipLaunchKernelGGL(k_fr_scale, dim3((uint32_t)((n * batch + 255) / 256)), b, 0, s, vals, inv_n, n * batch);
}

// ---------------------------------------------------------------------------------------------------------
// toeplitzCoeffsStepStrided (fk20_single.go:89-103), all `l` offsets of all `batch` polynomials at once:
// out[b][f][0] = p[n-1-f]; out[b][f][1..k+1] = 0; out[b][f][k+2+t] = p[2l - f - 1 + t l]
// ---------------------------------------------------------------------------------------------------------
__global__ void k_toeplitz_coeffs(const fr *poly, uint64_t poly_stride, uint64_t n, uint64_t l, uint64_t total, fr *out, const fr *scale) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t k = n / l, k2 = 2 * k;
    uint64_t i = t % k2, f = (t / k2) % l, b = t / (k2 * l);
    const fr *p = poly + b * poly_stride;
    fr v = zero<FrP>();
    if (i == 0) v = p[n - 1 - f];
    else if (i >= k + 2) v = p[2 * l - f - 1 + (i - (k + 2)) * l];
    if (scale && (i == 0 || i >= k + 2)) v = mul(v, *scale);
    out[t] = v;
}
void launch_toeplitz_coeffs(hipStream_t s, const fr *poly, uint64_t poly_stride, uint64_t n, uint64_t l, uint64_t batch, fr *out, const fr *scale) {
    uint64_t total = batch * l * 2 * (n / l);
    if (!total) return;
    hipLaunchKernelGGL(k_toeplitz_coeffs, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, poly, poly_stride, n, l, total, out, scale);
}

// ---------------------------------------------------------------------------------------------------------
// q = p / (X - x): q[nq-1] = p[n-1], q[i] = p[i+1] + x q[i+1]  (what polyLongDiv computes for the monic linear
// divisor, poly.go:14-40; the reference spends one InvModFr per step on the constant 1).  Blocked Horner:
// 256 lanes each run a contiguous segment with carry-in 0, a suffix scan composes the 256 segment maps, then every
// lane adds x^(distance) * carry.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quotient_linear(const fr *poly_all, uint64_t poly_stride, uint64_t n, const fr *xp, fr *q_all, uint64_t q_stride) {
    __shared__ fr head[256], xpow[256], cin[256];
    const uint32_t t = threadIdx.x;
    const fr *poly = poly_all + (uint64_t)blockIdx.x * poly_stride;      // one workgroup per polynomial of the batch
    fr *q = q_all + (uint64_t)blockIdx.x * q_stride;
    const uint64_t nq = n - 1, m = (nq + 255) / 256;
    const uint64_t lo = (uint64_t)t * m, hi = (lo + m < nq) ? lo + m : nq;
    const fr x = xp[blockIdx.x];
    fr acc = zero<FrP>(), pw = one<FrP>();
    if (lo < nq) {
        for (uint64_t i = hi; i-- > lo;) { acc = add(poly[i + 1], mul(x, acc)); q[i] = acc; pw = mul(pw, x); }
    }
    // segment s maps its carry-in c to head[s] + xpow[s] c; the carry into segment s is the composition of the maps of segments
    // s + 1 .. 255 applied to 0.  Suffix scan of affine maps (Kogge-Stone, 8 steps of two products) instead of a 256-step chain:
    // (H, P)[s] <- (H[s] + P[s] H[s + d], P[s] P[s + d]).
    head[t] = acc; xpow[t] = pw;
    __syncthreads();
#pragma nounroll
    for (uint32_t d = 1; d < 256; d <<= 1) {
        fr h = head[t], pq = xpow[t];
        const bool has = t + d < 256;
        fr h2, p2;
        if (has) { h2 = head[t + d]; p2 = xpow[t + d]; }
        __syncthreads();
        if (has) { head[t] = add(h, mul(pq, h2)); xpow[t] = mul(pq, p2); }
        __syncthreads();
    }
    cin[t] = t + 1 < 256 ? head[t + 1] : zero<FrP>();
    __syncthreads();
    if (lo < nq) {
        fr c = cin[t];
        if (!is_zero<FrP>(c)) {
            fr p = one<FrP>();
            for (uint64_t i = hi; i-- > lo;) { p = mul(p, x); q[i] = add(q[i], mul(p, c)); }
        }
    }
}
void launch_quotient_linear(hipStream_t s, const fr *poly, uint64_t poly_stride, uint64_t n, uint64_t batch, const fr *x, fr *q, uint64_t q_stride) {
    if (!batch) return;
    hipLaunchKernelGGL(k_quotient_linear, dim3((uint32_t)batch), dim3(256), 0, s, poly, poly_stride, n, x, q, q_stride);
}
// bls.AsFr over a slice (bls/bignum_kilic.go:61-65): out[i] = Montgomery image of the uint64 in[i * in_stride]
__global__ void k_fr_from_u64(const uint64_t *in, uint64_t in_stride, fr *out, uint64_t n) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) out[t] = fr_from_u64(in[t * in_stride]);
}
void launch_fr_from_u64(hipStream_t s, const uint64_t *in, uint64_t in_stride, fr *out, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_from_u64, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, in_stride, out, n);
}
// coalesced one-polynomial calls of different lengths share a batch: row b holds lens[b * lens_stride] coefficients, the rest of
// the n_max-wide row is zero-filled here (zero coefficients add nothing to a commitment or a quotient)
__global__ void k_fr_zero_tails(fr *rows, uint64_t n_max, const uint64_t *lens, uint64_t lens_stride, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t b = t / n_max, i = t % n_max;
    if (i >= lens[b * lens_stride]) rows[t] = zero<FrP>();
}
void launch_fr_zero_tails(hipStream_t s, fr *rows, uint64_t n_max, uint64_t batch, const uint64_t *lens, uint64_t lens_stride) {
    uint64_t total = n_max * batch;
    if (!total) return;
    hipLaunchKernelGGL(k_fr_zero_tails, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, rows, n_max, lens, lens_stride, total);
}

__global__ void k_fr_any_nonzero(const fr *vals, uint64_t n, uint32_t *flag) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n && !is_zero<FrP>(vals[t])) atomicOr(flag, 1u);
}
void launch_fr_any_nonzero(hipStream_t s, const fr *vals, uint64_t n, uint32_t *flag) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_any_nonzero, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, vals, n, flag);
}

__global__ void k_fr_powers(const fr *base, uint64_t n, fr *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    fr acc = one<FrP>(), pw = *base;
    for (uint64_t e = t; e; e >>= 1) { if (e & 1) acc = mul(acc, pw); pw = mul(pw, pw); }
    out[t] = acc;
}
void launch_fr_powers(hipStream_t s, const fr *base, uint64_t n, fr *out) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_powers, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, base, n, out);
}

// ---------------------------------------------------------------------------------------------------------
// eth/ byte-level path.
// bytes_to_bls_field / bls.FrFrom32 (eth/helpers.go:105-109, bls/bignum_kilic.go:33-44, bls.ValidFr bls/bignum_all.go:12-35):
// 32 little-endian bytes -> Montgomery image, rejected (flag of the blob set) unless the value is < r.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_fr_from_le32(const uint8_t *in, fr *out, uint64_t per_blob, uint64_t total, uint32_t *bad) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    fr v = *reinterpret_cast<const fr *>(in + 32 * t);      // little-endian host == little-endian limbs
    bool lt = false;
    for (int i = 7; i >= 0; i--) { uint32_t m = FrP::mod(i); if (v.l[i] < m) { lt = true; break; } if (v.l[i] > m) break; }
    if (!lt) { atomicOr(&bad[t / per_blob], 1u); out[t] = zero<FrP>(); return; }
    out[t] = to_mont<FrP>(v);
}
void launch_fr_from_le32(hipStream_t s, const uint8_t *in, fr *out, uint64_t per_blob, uint64_t batch, uint32_t *bad) {
    uint64_t total = per_blob * batch;
    if (!total) return;
    hipLaunchKernelGGL(k_fr_from_le32, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, in, out, per_blob, total, bad);
}
__global__ void k_fr_bitrev_gather(const fr *in, fr *out, uint32_t logn, uint64_t n) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[t] = in[bitrev32((uint32_t)t, logn)];
}
void launch_fr_bitrev_gather(hipStream_t s, const fr *in, fr *out, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_bitrev_gather, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, out, ilog2(n), n);
}
// ComputeKZGProof's field part (eth/helpers.go:179-198) with bls.EvaluatePolyInEvaluationForm (bls/globals.go:106-153), one workgroup per
// (polynomial, z) row of a batch:
//   y = (z^n - 1) / n * sum_i p_i w_i / (z - w_i);   q_i = (p_i - y) / (w_i - z).
// Every lane inverts the PRODUCT of its (up to four) denominators once (binary GCD) and unwinds it (Montgomery's trick: the reference's
// BatchInvModFr, per lane); LDS tree for the sum.  A row whose z is in the domain ("invalid z challenge", :190-192) sets flag[row] and gets
// a zero quotient (its proof is never looked at) -- no host round trip between this kernel and the commitment of the quotients.
__global__ __launch_bounds__(1024) void k_eth_quotient(const fr *poly_all, uint64_t poly_stride, const fr *domain, uint64_t dom_stride, uint64_t n, const fr *z_all,
                                                       uint64_t z_stride, const fr *inv_n, fr *q_all, fr *y_all, uint32_t *flag_all) {
    // dynamic LDS: the product tree of the workgroup's batch inversion (2048 elements, coop_inv.hpp); its first 1024 elements double as the reduction area below
    extern __shared__ uint32_t eth_q_smem[];
    fr *tree = reinterpret_cast<fr *>(eth_q_smem);
    fr *red = tree;
    __shared__ uint32_t bad;
    __shared__ fr zn_sh, y_sh;
    const uint32_t tid = threadIdx.x;
    const uint64_t row = blockIdx.x;
    const fr *poly = poly_all + row * poly_stride;
    fr *q = q_all + row * n;
    const fr z = z_all[row * z_stride];
    if (tid == 0) bad = 0;
    __syncthreads();
#ifdef KZG_ETHQ_TIMING                                          // A/B builds: phase times of row 0 on stdout (100 MHz wall clock)
    uint64_t tq[8]; int tqi = 0;
#define ETHQ_T() do { __syncthreads(); tq[tqi++] = wall_clock64(); } while (0)
#else
#define ETHQ_T() do { } while (0)
#endif
    ETHQ_T();
    fr part = zero<FrP>();
    for (uint64_t base = 0; base < n; base += 4096) {
        fr d[4], pv[4], pre[4];
        uint32_t cnt = 0, hit = 0;
        fr acc = one<FrP>();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t i = base + tid + 1024u * k;
            if (i < n) {
                d[k] = sub(z, domain[i * dom_stride]);
                pv[k] = poly[i];
                if (is_zero<FrP>(d[k])) { hit = 1; d[k] = one<FrP>(); }
                pre[k] = acc;
                acc = mul(acc, d[k]);
                cnt = k + 1;
            }
        }
        if (hit) atomicOr(&bad, 1u);
        ETHQ_T();
        // 1 / acc for all 1024 lanes with ONE inversion (round 6): a product tree in LDS, its root inverted by the first wavefront cooperatively, inverses pushed
        // back down.  Through round 5 every lane ran its own binary GCD here: 16 wavefronts x ~26 k instructions on one CU, ~170 us of a lone ComputeKZGProof's 0.62 ms.
        // (acc is never zero: a zero denominator was replaced by one above, and lanes beyond n hold one.)
#ifdef KZG_ETH_QUOTIENT_LANE_INV
        fr ia = inv<FrP>(acc);
#else
        // (z^n -- 12 dependent squarings that every one of the 1024 lanes used to run after the reduction: 64 us of a 184 us kernel with 16 wavefronts sharing the CU's
        // issue slots -- is computed ONCE, by the second wavefront, while the first one inverts the root)
        fr ia = block_batch_inverse<FrP, 10>(acc, tree, tid, [&]() {
            if (base != 0) return;
            fr t = z;
            for (uint64_t m = 1; m < n; m <<= 1) t = sqr(t);
            if (tid == 64) zn_sh = t;
        });
#endif
        ETHQ_T();
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if ((uint32_t)k < cnt) {
                const uint64_t i = base + tid + 1024u * k;
                const fr di = mul(ia, pre[k]);                     // 1 / (z - w_i)
                ia = mul(ia, d[k]);
                q[i] = di;                                         // stash it; second use below, after y is known
                part = add(part, mul(mul(pv[k], domain[i * dom_stride]), di));
            }
        }
    }
    ETHQ_T();
    __syncthreads();                                            // (every lane has read its inverse out of the tree before the area is reused)
    red[tid] = part;
    __syncthreads();
    for (uint32_t off = 512; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = add(red[tid], red[tid + off]);
        __syncthreads();
    }
    ETHQ_T();
#ifdef KZG_ETH_QUOTIENT_LANE_INV
    fr zn = z;                                                  // z^n, n a power of two
    for (uint64_t m = 1; m < n; m <<= 1) zn = sqr(zn);
    const fr y = mul(mul(red[0], sub(zn, one<FrP>())), *inv_n);
#else
    if (tid == 0) y_sh = mul(mul(red[0], sub(zn_sh, one<FrP>())), *inv_n);    // one lane: y = (z^n - 1) / n * sum (eth/helpers.go:199-201)
    __syncthreads();
    const fr y = y_sh;
#endif
    ETHQ_T();
    const bool invalid = bad != 0;
    if (tid == 0) { y_all[row] = invalid ? zero<FrP>() : y; if (invalid) flag_all[row] = 1u; }
    for (uint64_t i = tid; i < n; i += 1024) {
        const fr di = q[i];
        q[i] = invalid ? zero<FrP>() : neg<FrP>(mul(sub(poly[i], y), di));   // (p_i - y) / (w_i - z)
    }
    ETHQ_T();
#ifdef KZG_ETHQ_TIMING
    if (tid == 0 && row == 0 && n <= 4096) printf("ethq phases (us): load+prefix %.1f | inverse %.1f | unwind %.1f | reduce %.1f | z^n, y %.1f | final %.1f\n", (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01,
                                    (tq[3] - tq[2]) * 0.01, (tq[4] - tq[3]) * 0.01, (tq[5] - tq[4]) * 0.01, (tq[6] - tq[5]) * 0.01);
#endif
#undef ETHQ_T
}
// The same quotient with a row spread over S = n / 1024 workgroups (round 6; small batches, n >= 2048): ONE element per lane, so a lone ComputeKZGProof uses four CUs
// instead of one and a lane has no local prefix products to build and unwind -- the single-workgroup kernel above spends 65 of its 130 us on them with 16 wavefronts
// sharing one CU's issue slots.  Two launches, no cross-workgroup synchronisation: (1) each workgroup inverts its 1024 denominators (product tree + one cooperative
// inversion), stashes 1 / (z - w_i) in q and writes its share of sum p_i w_i / (z - w_i) and its z^n; (2) every workgroup adds the row's S shares, forms y and finishes its
// own 1024 elements.  shares: S + 1 elements per row (the last one z^n).
__global__ __launch_bounds__(1024) void k_eth_quotient_parts(const fr *poly_all, uint64_t poly_stride, const fr *domain, uint64_t dom_stride, uint64_t n, uint32_t S, const fr *z_all,
                                                             uint64_t z_stride, fr *q_all, uint32_t *flag_all, fr *shares) {
    extern __shared__ uint32_t eth_q_smem[];
    fr *tree = reinterpret_cast<fr *>(eth_q_smem);
    fr *red = tree;
    __shared__ uint32_t bad;
    __shared__ fr zn_sh;
    const uint32_t tid = threadIdx.x;
    const uint64_t row = blockIdx.x / S; const uint32_t part = blockIdx.x % S;
    const fr *poly = poly_all + row * poly_stride;
    fr *q = q_all + row * n;
    const fr z = z_all[row * z_stride];
    if (tid == 0) bad = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)part * 1024u + tid;            // < n: n = 1024 S
    const fr w = domain[i * dom_stride];
    fr d = sub(z, w);
    if (is_zero<FrP>(d)) { atomicOr(&bad, 1u); d = one<FrP>(); }
    const fr pw = mul(poly[i], w);                              // (independent of the inverse: issued before the tree's barriers)
    const fr di = block_batch_inverse<FrP, 10>(d, tree, tid, [&]() {
        fr t = z;                                               // z^n on the second wavefront while the first one inverts
        for (uint64_t m = 1; m < n; m <<= 1) t = sqr(t);
        if (tid == 64) zn_sh = t;
    });
    q[i] = di;                                                  // stash 1 / (z - w_i); second use in the finishing launch
    const fr term = mul(pw, di);
    __syncthreads();                                            // (every lane has read its inverse out of the tree before the area is reused)
    red[tid] = term;
    __syncthreads();
    for (uint32_t off = 512; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = add(red[tid], red[tid + off]);
        __syncthreads();
    }
    if (tid == 0) {
        shares[row * (S + 1) + part] = red[0];
        if (part == 0) shares[row * (S + 1) + S] = zn_sh;
        if (bad) atomicOr(&flag_all[row], 1u);
    }
}
__global__ __launch_bounds__(1024) void k_eth_quotient_finish(const fr *poly_all, uint64_t poly_stride, uint64_t n, uint32_t S, const fr *inv_n, fr *q_all, fr *y_all,
                                                              const uint32_t *flag_all, const fr *shares) {
    const uint32_t tid = threadIdx.x;
    const uint64_t row = blockIdx.x / S; const uint32_t part = blockIdx.x % S;
    fr sum = shares[row * (S + 1)];                             // (wave-uniform loads: every lane forms the same y)
    for (uint32_t p = 1; p < S; p++) sum = add(sum, shares[row * (S + 1) + p]);
    const fr y = mul(mul(sum, sub(shares[row * (S + 1) + S], one<FrP>())), *inv_n);   // y = (z^n - 1) / n * sum (eth/helpers.go:199-201)
    const bool invalid = flag_all[row] != 0;
    if (tid == 0 && part == 0) y_all[row] = invalid ? zero<FrP>() : y;
    const uint64_t i = (uint64_t)part * 1024u + tid;
    fr *q = q_all + row * n;
    q[i] = invalid ? zero<FrP>() : neg<FrP>(mul(sub(poly_all[row * poly_stride + i], y), q[i]));   // (p_i - y) / (w_i - z)
}
// test hook: element i inverted by wavefront i cooperatively (out_coop), by lane 0 of that wavefront alone (out_lane), and -- workgroups of 1024 consecutive elements, the
// tail padded with ones -- by the workgroup batch inversion the quotient kernel uses (out_block)
__global__ __launch_bounds__(64) void k_fr_inv_both(const fr *in, fr *out_coop, fr *out_lane) {
    const fr x = in[blockIdx.x];
    const fr y = wave_inv<FrP>(x, 0);
    if (threadIdx.x == 0) { out_coop[blockIdx.x] = y; out_lane[blockIdx.x] = inv<FrP>(x); }
}
__global__ __launch_bounds__(1024) void k_fr_inv_block(const fr *in, uint64_t n, fr *out_block) {
    extern __shared__ uint32_t inv_smem[];
    const uint64_t i = blockIdx.x * 1024ull + threadIdx.x;
    fr v = i < n ? in[i] : one<FrP>();
    if (is_zero<FrP>(v)) v = one<FrP>();                           // (the batch form needs non-zero values, like its caller guarantees)
    const fr y = block_batch_inverse<FrP, 10>(v, reinterpret_cast<fr *>(inv_smem), threadIdx.x);
    if (i < n) out_block[i] = y;
}
void launch_fr_inv_test(hipStream_t s, const fr *in, uint64_t n, fr *out_coop, fr *out_lane, fr *out_block) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_inv_both, dim3((uint32_t)n), dim3(64), 0, s, in, out_coop, out_lane);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_inv_block), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2048 * sizeof(fr)));
    hipLaunchKernelGGL(k_fr_inv_block, dim3((uint32_t)((n + 1023) / 1024)), dim3(1024), 2048 * sizeof(fr), s, in, n, out_block);
}
uint64_t eth_quotient_scratch_elems(uint64_t n, uint64_t batch) {   // elements the caller appends to its quotient buffer for the row-split form (0: not used at this shape)
    static const bool one_wg = [] { const char *e = getenv("KZG_HIP_ETH_QUOTIENT"); return e && !strcmp(e, "one"); }();   // A/B and test hook: one workgroup per row at every size
    const uint64_t S = n / 1024;
    // small batches only: from ~64 rows on the one-workgroup rows fill the chip and their four elements per lane amortise the tree (512 rows: 87 k against 66 k proofs/s)
    if (one_wg || n < 2048 || (n & (n - 1)) != 0 || S > 64 || batch * S > 256) return 0;
    return batch * (S + 1);
}
void launch_eth_quotient(hipStream_t s, const fr *poly, uint64_t poly_stride, const fr *domain, uint64_t n, uint64_t batch, const fr *z, uint64_t z_stride,
                         const fr *inv_n, fr *q, fr *y_out, uint32_t *flag, uint64_t dom_stride, fr *scratch) {
    if (!batch) return;
    constexpr size_t lds = 2048 * sizeof(fr);                    // 64 KiB: the batch inversion's product tree
    if (scratch && eth_quotient_scratch_elems(n, batch)) {       // a row over S workgroups, two launches
        const uint32_t S = (uint32_t)(n / 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eth_quotient_parts), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_eth_quotient_parts, dim3((uint32_t)(batch * S)), dim3(1024), lds, s, poly, poly_stride, domain, dom_stride, n, S, z, z_stride, q, flag, scratch);
        hipLaunchKernelGGL(k_eth_quotient_finish, dim3((uint32_t)(batch * S)), dim3(1024), 0, s, poly, poly_stride, n, S, inv_n, q, y_out, flag, scratch);
        return;
    }
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eth_quotient), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // (per device: set on every launch, like the F_r transforms)
    hipLaunchKernelGGL(k_eth_quotient, dim3((uint32_t)batch), dim3(1024), lds, s, poly, poly_stride, domain, dom_stride, n, z, z_stride, inv_n, q, y_out, flag);
}

// bls.PolyLinComb (bls/globals.go:155-178) over resident rows: out[i] = sum_j scalars[j] * vectors[j][i]; a lane per coefficient, the scalars
// wave-uniform.  No vector -> zeros (:157-159).
__global__ void k_poly_lincomb(const fr *vectors, uint64_t stride, const fr *scalars, uint64_t count, uint64_t n, fr *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    fr acc = zero<FrP>();
    for (uint64_t j = 0; j < count; j++) acc = add(acc, mul(scalars[j], vectors[j * stride + t]));
    out[t] = acc;
}
void launch_poly_lincomb(hipStream_t s, const fr *vectors, uint64_t stride, const fr *scalars, uint64_t count, uint64_t n, fr *out) {
    if (!n) return;
    hipLaunchKernelGGL(k_poly_lincomb, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, vectors, stride, scalars, count, n, out);
}

// CheckProofMulti's coefficient scaling (kzg_multi_proofs.go:55-66): c_i <- c_i / x^i.  Lane i computes x^-i by
// square-and-multiply on the once-inverted x (the reference inverts x^i afresh for every i).
__global__ void k_fr_scale_by_inv_powers(fr *c, const fr *x, uint64_t n, fr *xpow_n) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    fr xi = inv<FrP>(*x), acc = one<FrP>(), pw = xi;
    for (uint64_t e = t; e; e >>= 1) { if (e & 1) acc = mul(acc, pw); pw = sqr(pw); }
    c[t] = mul(c[t], acc);
    if (t == 0 && xpow_n) {
        fr a = one<FrP>(), p2 = *x;
        for (uint64_t e = n; e; e >>= 1) { if (e & 1) a = mul(a, p2); p2 = sqr(p2); }
        *xpow_n = a;
    }
}
void launch_fr_scale_by_inv_powers(hipStream_t s, fr *c, const fr *x, uint64_t n, fr *xpow_n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_scale_by_inv_powers, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, c, x, n, xpow_n);
}

// ---------------------------------------------------------------------------------------------------------
// Erasure recovery: zero_poly.go:116-217, recover_from_samples.go:9-109.
// The vanishing polynomial Z(X) = prod_{i missing} (X - w^i) is unique, so instead of the reference's leaf / FFT-product
// tree the device evaluates it directly on the domain -- zero_eval[k] = prod_i (w^k - w^{m_i}), one lane per k, the
// missing-root loads are wave-uniform -- and gets the coefficients with one inverse FFT.  O(n m) Fr products: 5e8 at
// scale 15 with half the samples missing, a few ms on the chip against 172 ms for the reference's tree (BENCH.md).
// ---------------------------------------------------------------------------------------------------------
// The product chain runs on lazy 29-bit limbs (fr_lazy.hpp): frl_mul divides by 2^261 while both operands are Kilic images (2^256), so every step
// leaves a stray 2^-5; `corr` = the Kilic image of 2^(5 n_missing) (computed on the host) takes all of them out in one last product.  A step is one
// lazy subtraction + 153 multiply-adds instead of a canonical subtraction + the 170-multiply-add product with its packing: 1.5x.
// A point's chain is cut into `segs` pieces, one wavefront of the workgroup each (the missing root of a step stays wave-uniform), multiplied
// together through LDS at the end: 65 536 points alone are one wavefront per SIMD, and a lone wavefront waits on its own dependent products.
__global__ void __launch_bounds__(1024) k_zero_eval_direct(const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t length,
                                                           fr *zero_eval, fr corr) {
    extern __shared__ uint32_t zsh[];                                     // [segs - 1][9][64]
    const uint32_t lane = threadIdx.x & 63, seg = threadIdx.x >> 6, segs = blockDim.x >> 6;
    uint64_t k = blockIdx.x * 64ull + lane;
    if (k >= length) k = length - 1;                                      // idle lanes repeat the last point: every lane reaches the barrier
    const uint64_t per = (n_missing + segs - 1) / segs, lo = seg * per, hi = lo + per < n_missing ? lo + per : n_missing;
    const frl x = frl_unpack(expanded[k * stride]);
    frl acc = frl_unpack(one<FrP>());
#pragma nounroll
    for (uint64_t i = lo; i < hi; i++) {
        const frl m = frl_unpack(expanded[missing[i] * stride]);          // wave-uniform
        acc = frl_mul(frl_sub<2>(x, m), acc);                               // raw difference (bound 3) x normalised running product (bound < 2)
    }
    if (seg) {
#pragma unroll
        for (int j = 0; j < 9; j++) zsh[((seg - 1) * 9 + j) * 64 + lane] = acc.l[j];
    }
    __syncthreads();
    if (seg) return;
    for (uint32_t sgm = 1; sgm < segs; sgm++) {
        frl o;
#pragma unroll
        for (int j = 0; j < 9; j++) o.l[j] = zsh[((sgm - 1) * 9 + j) * 64 + lane];
        acc = frl_mul(o, acc);
    }
    if (blockIdx.x * 64ull + lane < length) zero_eval[k] = frl_canon_lt2r(frl_mul(acc, frl_const_from_kilic(corr)));
}
void launch_zero_eval_direct(hipStream_t s, const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t length, fr *zero_eval) {
    if (!length) return;
    uint32_t segs = 1;                                                    // aim at 4 wavefronts per SIMD (4096 on the chip)
    while (segs < 16 && length * segs < 262144 && 2ull * segs <= n_missing) segs *= 2;
    // every product of two Kilic images on the lazy limbs leaves 2^-5: n_missing chain steps (a segment with fewer steps than `per` simply has
    // fewer) + segs - 1 to join the segments
    fr corr = one<FrP>(), pw = fr_from_u64(32);                           // 2^(5 (n_missing + segs - 1)), by square-and-multiply on the host
    for (uint64_t e = n_missing + segs - 1; e; e >>= 1) { if (e & 1) corr = mul(corr, pw); pw = mul(pw, pw); }
    hipLaunchKernelGGL(k_zero_eval_direct, dim3((uint32_t)((length + 63) / 64)), dim3(64 * segs), (segs - 1) * 9 * 64 * 4, s, expanded, stride, missing,
                       n_missing, length, zero_eval, corr);
}
// data[b][i] *= table[i * stride] over rows of n values (the coefficient shift of the transform-based DAS extension: x -> w_2n x)
__global__ void k_fr_mul_table_rows(fr *data, const fr *table, uint64_t stride, uint64_t n, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    data[t] = mul(data[t], table[(t % n) * stride]);
}
void launch_fr_mul_table_rows(hipStream_t s, fr *data, const fr *table, uint64_t stride, uint64_t n, uint64_t batch) {
    const uint64_t total = n * batch;
    if (!total) return;
    hipLaunchKernelGGL(k_fr_mul_table_rows, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, data, table, stride, n, total);
}

// ---- the vanishing polynomial as a product tree (large erasure sets): monic factors kept as their non-leading part ----
// A node of degree d is x^d + a(x), deg a < d, stored as the d coefficients of a.  Leaves hold 16 roots; a leaf with fewer (the tail, and the
// padding up to a power of two of leaves) is filled with roots at 0, i.e. multiplied by x: every node stays monic of its level's degree, and the
// root is Z(x) x^pad.  Joining two nodes: (x^d + a)(x^d + b) = x^2d + x^d (a + b) + a b, and a b (degree <= 2d - 2) is one cyclic product of
// size 2d -- two forward transforms, a pointwise product, an inverse transform, all batched over the level.
constexpr int ZLEAF = 16;
__global__ void __launch_bounds__(64) k_zero_leaves(const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t leaves, fr *a) {
    __shared__ fr c[ZLEAF + 1][64];                                       // running product, coefficient-major: lane-contiguous rows
    const uint32_t lane = threadIdx.x;
    const uint64_t t = blockIdx.x * 64ull + lane;
    if (t >= leaves) return;
    const uint64_t lo = t * ZLEAF;
    c[0][lane] = one<FrP>();
    for (int i = 0; i < ZLEAF; i++) {                                     // c <- c (x - r_i); beyond the erasure set r = 0: c <- c x
        const bool real = lo + i < n_missing;
        const fr r = real ? expanded[missing[lo + i] * stride] : zero<FrP>();
        c[i + 1][lane] = c[i][lane];                                      // the leading coefficient (1) moves up
        for (int j = i; j >= 1; j--) c[j][lane] = real ? sub(c[j - 1][lane], mul(r, c[j][lane])) : c[j - 1][lane];
        c[0][lane] = real ? neg<FrP>(mul(r, c[0][lane])) : zero<FrP>();
    }
    for (int j = 0; j < ZLEAF; j++) a[t * ZLEAF + j] = c[j][lane];
}
// out[p][k] = f[2p][k] * f[2p + 1][k], rows of m values
__global__ void k_zero_pair_products(const fr *f, uint64_t m, uint64_t total, fr *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t p = t / m, k = t - p * m;
    out[t] = mul(f[2 * p * m + k], f[(2 * p + 1) * m + k]);
}
// c[p][k] += a[2p][k - d] + a[2p + 1][k - d] for k >= d (rows of 2d values in c, of d values in a)
__global__ void k_zero_join(fr *c, const fr *a, uint64_t d, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t p = t / (2 * d), k = t - p * 2 * d;
    if (k >= d) c[t] = add(c[t], add(a[2 * p * d + k - d], a[(2 * p + 1) * d + k - d]));
}
// Z = root / x^pad: poly[k] = root[k + pad] below the degree, 1 at the degree, 0 above
__global__ void k_zero_unpad(const fr *root, uint64_t pad, uint64_t n_missing, uint64_t length, fr *poly) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= length) return;
    poly[t] = t < n_missing ? root[t + pad] : (t == n_missing ? one<FrP>() : zero<FrP>());
}
void launch_zero_leaves(hipStream_t s, const fr *expanded, uint64_t stride, const uint64_t *missing, uint64_t n_missing, uint64_t leaves, fr *a) {
    hipLaunchKernelGGL(k_zero_leaves, dim3((uint32_t)((leaves + 63) / 64)), dim3(64), 0, s, expanded, stride, missing, n_missing, leaves, a);
}
void launch_zero_pair_products(hipStream_t s, const fr *f, uint64_t m, uint64_t pairs, fr *out) {
    const uint64_t total = m * pairs;
    hipLaunchKernelGGL(k_zero_pair_products, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, f, m, total, out);
}
void launch_zero_join(hipStream_t s, fr *c, const fr *a, uint64_t d, uint64_t pairs) {
    const uint64_t total = 2 * d * pairs;
    hipLaunchKernelGGL(k_zero_join, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, c, a, d, total);
}
void launch_zero_unpad(hipStream_t s, const fr *root, uint64_t pad, uint64_t n_missing, uint64_t length, fr *poly) {
    hipLaunchKernelGGL(k_zero_unpad, dim3((uint32_t)((length + 255) / 256)), dim3(256), 0, s, root, pad, n_missing, length, poly);
}

// poly[i] *= base^i  (ShiftPoly / UnshiftPoly, recover_from_samples.go:9-40, with base = 5^-1 / 5)
__global__ void k_fr_scale_by_powers(fr *poly, const fr *base, uint64_t n) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    fr acc = one<FrP>(), pw = *base;
    for (uint64_t e = t; e; e >>= 1) { if (e & 1) acc = mul(acc, pw); pw = sqr(pw); }
    poly[t] = mul(poly[t], acc);
}
void launch_fr_scale_by_powers(hipStream_t s, fr *poly, const fr *base, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_scale_by_powers, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, poly, base, n);
}
// mode 0: out[i] = present[i] ? a[i] * b[i] : 0      (recover_from_samples.go:66-73)
// mode 1: out[i] = a[i] / b[i]                        (DivModFr loop, :93-96)
// mode 2: flag |= present[i] && a[i] != b[i]          (final consistency check, :103-107)
__global__ void k_fr_pointwise(const fr *a, const fr *b, const uint8_t *present, fr *out, uint64_t n, int mode, uint32_t *flag) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (mode == 0) out[t] = present[t] ? mul(a[t], b[t]) : zero<FrP>();
    else if (mode == 1) out[t] = mul(a[t], inv<FrP>(b[t]));
    else if (present[t] && !equal<FrP>(a[t], b[t])) atomicOr(flag, 1u);
}
void launch_fr_pointwise(hipStream_t s, const fr *a, const fr *b, const uint8_t *present, fr *out, uint64_t n, int mode, uint32_t *flag) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_pointwise, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a, b, present, out, n, mode, flag);
}

__global__ void k_fr_to_le32(const fr *in, uint8_t *out, uint64_t n) {   // bls.FrTo32 (bls/bignum_kilic.go:46-55)
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    *reinterpret_cast<fr *>(out + 32 * t) = from_mont<FrP>(in[t]);
}
void launch_fr_to_le32(hipStream_t s, const fr *in, uint8_t *out, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_to_le32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

}  // namespace kzg
