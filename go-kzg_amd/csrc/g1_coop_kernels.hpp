// g1_coop_kernels.hpp -- the G1 transform kernels with several lanes per multiplication (g1_quad.hpp): the stage kernel with four / two lanes per
// butterfly and the direct pass with four / two lanes per (output, term).  Each instantiation inlines a whole scalar multiplication, so they are
// compiled in translation units of their own (k_g1_coop_dif.hip, k_g1_coop_dit.hip, k_g1_coop_direct.hip) beside k_g1.hip.
#pragma once
#include "internal.hpp"
#include "g1_quad.hpp"
#include <stdlib.h>
#include <string.h>

namespace kzg {

#ifndef G1_BLOCK
#define G1_BLOCK 256
#endif
struct g1jq_slot { uint32_t w[39]; uint32_t inf; uint32_t pad; };            // 41 words: odd stride, no LDS bank conflicts
#define G1_DIRECT_BLOCK 64              // one wavefront per workgroup: the dispatcher spreads 1024 of them over the 1024 SIMDs
static inline uint32_t ilog2_coop(uint64_t v) { uint32_t r = 0; while ((1ull << r) < v) r++; return r; }

// The same stages with FOUR lanes per butterfly (g1_quad.hpp): for launches that would leave most SIMDs empty (at most 16 384 butterflies: up to 8
// polynomials of 4096 points), where the stage time is the latency of one scalar multiplication on one lane.  The quad shares the digit loop of
// the multiplication (648 dependency levels of one product each instead of ~1 370 sequential products); loads, the table, the shared
// (x + y, x - y) formulas and the packing are computed redundantly by the four lanes, lane 0 of the quad stores.  Regular odd-digit schedule
// (lanes of a wavefront may hold different twiddles).  DIF = false: (x, y) -> (x + w y, x - w y); DIF = true: (x, y) -> (x + y, (x - y) w).
// WNAF: the width-5 NAF digit rows of the twiddles (wave-uniform launches) instead of the regular odd-digit schedule.
// L = 2: the same with PAIRS of lanes (two rounds for the levels of three or four products): for launches that no longer fit four lanes per butterfly
// into one wavefront per SIMD but still leave half of the SIMDs empty (9-16 polynomials of 4096 points).
template <bool DIF, bool WNAF, int L> __global__ __launch_bounds__(G1_BLOCK, 2) void k_g1_fft_stage_quad(g1j *data, uint32_t logn, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W, uint64_t total, uint64_t batch) {
    const uint64_t t4 = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint32_t role = (uint32_t)(t4 & (uint64_t)(L - 1));
    const uint64_t t = t4 / L;
    if (t >= total) return;
    const uint64_t half = 1ull << (logn - 1), groups = half / m;
    const uint64_t j = t / (groups * batch), rem = t % (groups * batch), b = rem / groups, g = rem % groups;   // twiddle-major, as in k_g1_fft_stage
    g1j *row = data + (b << logn);
    const uint64_t i0 = g * 2 * m + j, i1 = i0 + m;
    g1j y = row[i1];
    g1j x = row[i0];
    const uint64_t ti = j * (W / (2 * m));
    const fr kk = roots[ti];                                // (k1, k2) GLV pair, both halves non-negative
    glv_halves h;
#pragma unroll
    for (int i = 0; i < 4; i++) { h.k1[i] = kk.l[i]; h.k2[i] = kk.l[4 + i]; }
    h.neg1 = h.neg2 = 0;
    g1aq tbl[8]; fq dz[7]; g1j packed;
#define QMUL(pq, res) (WNAF ? g1_mul_glv_wnaf_quad<L>((pq), kk, tbl, dz, wnaf + ti * KZG_WNAF_ROW, (res), packed, role) : g1_mul_glv_regular_quad<L>((pq), h, tbl, dz, (res), packed, role))
    if (!DIF) {
        g1jq yq; int st = is_inf(y) ? 0 : 1;
        if (st == 1) {
            if (j) {
                st = QMUL(g1jq_unpack(y), yq);
                if (st == 2) y = packed; else if (st == 0) y = g1_inf();
            } else yq = g1jq_unpack(y);
        }
        if (st == 1 && !is_inf(x)) {
            g1jq sum, dif;
            if (KZG_LIKELY(g1jq_addsub(g1jq_unpack(x), yq, sum, dif))) {
                if (role == 0) {
                    fp z3 = packq(sum.z);
                    g1j o0, o1;
                    o0.x = packq(sum.x); o0.y = packq(sum.y); o0.z = z3;
                    o1.x = packq(dif.x); o1.y = packq(dif.y); o1.z = z3;
                    row[i0] = o0; row[i1] = o1;
                }
                return;
            }
        }
        if (st == 1) y = g1jq_pack(yq);
        if (role == 0) { row[i0] = g1_add(x, y); row[i1] = g1_add(x, g1_neg(y)); }
    } else {
        if (!is_inf(x) && !is_inf(y)) {
            g1jq sum, dif;
            if (KZG_LIKELY(g1jq_addsub(g1jq_unpack(x), g1jq_unpack(y), sum, dif))) {
                g1j o0; o0.x = packq(sum.x); o0.y = packq(sum.y); o0.z = packq(sum.z);
                g1j o1;
                if (j) {
                    g1jq dq;
                    const int st = QMUL(dif, dq);
                    o1 = st == 1 ? g1jq_pack(dq) : st == 2 ? packed : g1_inf();
                } else { o1.x = packq(dif.x); o1.y = packq(dif.y); o1.z = o0.z; }
                if (role == 0) { row[i0] = o0; row[i1] = o1; }
                return;
            }
        }
        g1j s_ = g1_add(x, y), d_ = g1_add(x, g1_neg(y));   // an infinite operand or x == +-y: generic complete formulas
        if (j && !is_inf(d_)) {
            g1jq dq;
            const int st = QMUL(g1jq_unpack(d_), dq);
            d_ = st == 1 ? g1jq_pack(dq) : st == 2 ? packed : g1_inf();
        }
        if (role == 0) { row[i0] = s_; row[i1] = d_; }
    }
}
#undef QMUL
// the irregular width-5 NAF schedule where every wavefront holds one twiddle (L lanes x (n / 2 / m) batch butterflies per twiddle), else the regular one
static inline bool g1_quad_wnaf(uint64_t n, uint64_t batch, uint64_t m, int lanes) {
    static const bool off = [] { const char *e = getenv("KZG_HIP_G1_MUL"); return e && e[0] == 'r'; }();
    return !off && ((n / 2 / m) * batch * lanes) % 64 == 0;
}
// these launches are at most one 256-lane workgroup per CU: 96 KiB of unused dynamic LDS keeps the dispatcher from putting two on one CU (two
// wavefronts on a SIMD take 1.76x as long as one) while another CU stays empty
template <bool DIF> static inline void launch_stage_coop(hipStream_t s, int lanes, g1j *data, uint64_t n, uint64_t batch, uint64_t m, const fr *roots, const int8_t *wnaf, uint64_t W,
                                                  uint64_t total) {
    static const bool once = [] {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_g1_fft_stage_quad<DIF, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_g1_fft_stage_quad<DIF, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_g1_fft_stage_quad<DIF, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_g1_fft_stage_quad<DIF, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        return true; }();
    (void)once;
    const size_t lds = total * lanes <= device_simd_lanes() ? 96 * 1024 : 0;
    const dim3 qg((uint32_t)((lanes * total + G1_BLOCK - 1) / G1_BLOCK));
    const uint32_t logn = ilog2_coop(n);
    const bool wn = g1_quad_wnaf(n, batch, m, lanes);
    if (lanes == 4) {
        if (wn) hipLaunchKernelGGL((k_g1_fft_stage_quad<DIF, true, 4>), qg, dim3(G1_BLOCK), lds, s, data, logn, m, roots, wnaf, W, total, batch);
        else hipLaunchKernelGGL((k_g1_fft_stage_quad<DIF, false, 4>), qg, dim3(G1_BLOCK), lds, s, data, logn, m, roots, wnaf, W, total, batch);
    } else {
        if (wn) hipLaunchKernelGGL((k_g1_fft_stage_quad<DIF, true, 2>), qg, dim3(G1_BLOCK), lds, s, data, logn, m, roots, wnaf, W, total, batch);
        else hipLaunchKernelGGL((k_g1_fft_stage_quad<DIF, false, 2>), qg, dim3(G1_BLOCK), lds, s, data, logn, m, roots, wnaf, W, total, batch);
    }
}
// The same pass with L = 2 or 4 lanes per (output, term): the multiplication runs on the cooperative group operations of g1_quad.hpp (regular odd-digit
// schedule: the lanes of a wavefront hold different twiddles), the group's lane 0 joins the sum.  For passes that leave SIMDs empty at one lane per
// term: a lone transform of up to 1024 points runs radix 16 on quads (1.3 ms per pass instead of 2.0-2.4), 2048 points on pairs (1.8 ms).
template <int L> __global__ __launch_bounds__(G1_DIRECT_BLOCK, 2) void k_g1_fft_direct_coop(const g1j *in, uint64_t in_stride, uint64_t n_valid, g1j *out, uint32_t logn,
                                                                                          uint32_t logR, uint64_t Ns, const fr *roots, uint64_t W, const fr *scale, uint64_t total,
                                                                                          uint32_t logT, uint32_t logU) {
    __shared__ g1jq_slot buf[G1_DIRECT_BLOCK / L];
    const uint32_t tid = threadIdx.x, role = tid & (uint32_t)(L - 1), item = tid / L;
    const uint64_t t = (blockIdx.x * (uint64_t)blockDim.x + tid) / L;
    const uint64_t n = 1ull << logn, R = 1ull << logR, cols = n >> logR, T = 1ull << logT;
    const uint32_t tt = (uint32_t)(t & (T - 1));
    const uint64_t u = (t >> logT) & ((1ull << logU) - 1), jb = t >> (logT + logU), j = jb % cols, b = jb / cols;
    const bool live = t < total;
    g1jq_acc acc; acc.inf = true;
    uint64_t oidx = 0;
    if (live) {
        const uint64_t k = j & (Ns - 1), idx = j + (uint64_t)tt * cols;
        oidx = b * n + (j - k) * R + k + u * Ns;
        g1j x = idx < n_valid ? in[b * in_stride + idx] : g1_inf();
        if (!is_inf(x)) {
            const uint64_t e = ((uint64_t)tt * (cols / Ns) * (Ns * u + k)) & (n - 1);
            if (e == 0 && !scale) { acc.v = g1jq_unpack(x); acc.inf = false; }
            else {
                fr sc = roots[e * (W >> logn)];
                if (scale) sc = mul(sc, *scale);
                g1aq tbl[8]; fq dz[7]; g1j packed;
                const int st = g1_mul_glv_regular_quad<L>(g1jq_unpack(x), glv_split_signed(from_mont<FrP>(sc)), tbl, dz, acc.v, packed, role);
                if (st == 2) { acc.inf = is_inf(packed); if (!acc.inf) acc.v = g1jq_unpack(packed); } else acc.inf = st == 0;
            }
        }
    }
    // sum of the R terms of an output: items tt = 0 .. R - 1 are adjacent; one lane per item takes part
#define COOP_STORE(i) do { _Pragma("unroll") for (int q = 0; q < 13; q++) { buf[i].w[q] = acc.v.x.l[q]; buf[i].w[13 + q] = acc.v.y.l[q]; buf[i].w[26 + q] = acc.v.z.l[q]; } \
                           buf[i].inf = acc.inf ? 1u : 0u; } while (0)
    if (role == 0) COOP_STORE(item);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = (uint32_t)T / 2; off >= 1; off >>= 1) {
        if (role == 0 && tt < off && !buf[item + off].inf) {
            g1jq q;
#pragma unroll
            for (int i = 0; i < 13; i++) { q.x.l[i] = buf[item + off].w[i]; q.y.l[i] = buf[item + off].w[13 + i]; q.z.l[i] = buf[item + off].w[26 + i]; }
            acc.add(q);
            COOP_STORE(item);
        }
        __syncthreads();
    }
#undef COOP_STORE
    if (live && role == 0 && tt == 0) out[oidx] = acc.inf ? g1_inf() : g1jq_pack(acc.v);
}

}  // namespace kzg
