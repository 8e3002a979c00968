// k_msm.hip -- Pippenger multi-scalar multiplication, batched over polynomials ("blobs") that share one point
// table.  Replaces bls.LinCombG1 -> Kilic G1.MultiExp (bls/bls_kilic.go:132-150) behind CommitToPoly /
// ComputeProofSingle (kzg_single_proofs.go:17-19,36-54).
//
// Pipeline per blob (grid dimension = blob, so a batch fills the 256 CUs):
//   sort       : scalars leave Montgomery form (Kilic FromRed, bls_kilic.go:141-147), are cut into signed c-bit
//                digits, and (digit, point) pairs are counting-sorted by bucket inside one workgroup (LDS histogram,
//                LDS atomics for the scatter cursor).
//   accumulate : one lane per bucket walks its sorted list with mixed additions (affine table in HBM/L2).
//   reduce     : one wavefront per bucket group computes sum_k k * B_k by segment running sums + an LDS tree.
//   combine    : Horner over the window groups (c doublings per window), one lane per blob.
// Two table modes: per-window bucket groups over the plain points (any caller-supplied points), or "fixed base":
// the table also holds 2^(c w) P_i so all windows share ONE bucket group and no doublings remain.
// Bucket contents are summed in a data-dependent order; the group law is commutative and the result is
// normalised afterwards, so the output bytes do not depend on that order.
#include "internal.hpp"

namespace kzg {

#define MSM_SORT_T 1024
#define MSM_ACC_BLOCK 128

struct msm_ws_layout {
    size_t entries_off, offsets_off, buckets_off, gsum_off, per_blob;
    uint64_t K, nent;
};
static msm_ws_layout ws_layout(const msm_plan &p, uint64_t n) {
    msm_ws_layout L;
    L.K = (uint64_t)p.ngroups * p.nb;
    L.nent = n * p.nwin;
    size_t o = 0;
    L.entries_off = o; o += ((L.nent * 4 + 15) / 16) * 16;
    L.offsets_off = o; o += (((L.K + 1) * 4 + 15) / 16) * 16;
    L.buckets_off = o; o += L.K * sizeof(g1j);
    L.gsum_off = o; o += (size_t)p.ngroups * sizeof(g1j);
    L.per_blob = o;
    return L;
}
size_t msm_workspace_bytes(const msm_plan &p, uint64_t n, uint64_t batch) { return ws_layout(p, n).per_blob * batch; }

__device__ __forceinline__ uint32_t scalar_bits(const fr &k, uint32_t off, uint32_t c) {
    uint32_t idx = off >> 5, sh = off & 31;
    if (idx >= 8) return 0;
    uint64_t v = k.l[idx];
    if (idx + 1 < 8) v |= (uint64_t)k.l[idx + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// visits every non-zero signed digit of scalar k: f(window, |digit|, negative)
template <class Fn> __device__ __forceinline__ void for_each_digit(const fr &k, uint32_t c, uint32_t nwin, uint32_t nb, Fn f) {
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t raw = scalar_bits(k, w * c, c) + carry;
        if (raw > nb) { carry = 1; uint32_t mag = (1u << c) - raw; if (mag) f(w, mag, 1u); }   // raw == 2^c: digit 0, carry 1
        else { carry = 0; if (raw) f(w, raw, 0u); }
    }
}

__global__ __launch_bounds__(MSM_SORT_T) void k_msm_sort(msm_plan p, const fr *scalars, uint64_t n, uint8_t *ws, size_t per_blob, size_t entries_off,
                                                         size_t offsets_off, uint32_t K) {
    extern __shared__ uint32_t smem[];
    uint32_t *hist = smem, *part = smem + K;
    const uint32_t tid = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const fr *sc = scalars + b * n;
    uint32_t *entries = (uint32_t *)(ws + b * per_blob + entries_off);
    uint32_t *offsets = (uint32_t *)(ws + b * per_blob + offsets_off);
    for (uint32_t i = tid; i < K; i += MSM_SORT_T) hist[i] = 0;
    __syncthreads();
    for (uint64_t i = tid; i < n; i += MSM_SORT_T) {
        fr k = from_mont<FrP>(sc[i]);
        for_each_digit(k, p.c, p.nwin, p.nb, [&](uint32_t w, uint32_t mag, uint32_t) {
            uint32_t key = (p.fixed ? 0u : w * p.nb) + mag - 1;
            atomicAdd(&hist[key], 1u);
        });
    }
    __syncthreads();
    // exclusive prefix sum over K bins: per-thread chunk sums, block scan of the 1024 partials, write back
    const uint32_t per = (K + MSM_SORT_T - 1) / MSM_SORT_T;
    uint32_t lo = tid * per, hi = lo + per < K ? lo + per : K, sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += hist[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < MSM_SORT_T; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi; i++) { uint32_t cnt = hist[i]; offsets[i] = run; hist[i] = run; run += cnt; }
    if (tid == MSM_SORT_T - 1) offsets[K] = part[tid];
    __syncthreads();
    for (uint64_t i = tid; i < n; i += MSM_SORT_T) {
        fr k = from_mont<FrP>(sc[i]);
        for_each_digit(k, p.c, p.nwin, p.nb, [&](uint32_t w, uint32_t mag, uint32_t neg) {
            uint32_t key = (p.fixed ? 0u : w * p.nb) + mag - 1;
            uint32_t slot = atomicAdd(&hist[key], 1u);
            uint32_t tidx = p.fixed ? (uint32_t)(w * p.table_n + i) : (uint32_t)i;
            entries[slot] = (tidx << 1) | neg;
        });
    }
}

__global__ __launch_bounds__(MSM_ACC_BLOCK, 2) void k_msm_accumulate(const g1a *table, uint8_t *ws, size_t per_blob, size_t entries_off, size_t offsets_off,
                                                                  size_t buckets_off, uint32_t K, uint64_t total) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t b = t / K; uint32_t key = (uint32_t)(t % K);
    const uint32_t *entries = (const uint32_t *)(ws + b * per_blob + entries_off);
    const uint32_t *offsets = (const uint32_t *)(ws + b * per_blob + offsets_off);
    g1j *buckets = (g1j *)(ws + b * per_blob + buckets_off);
    uint32_t s = offsets[key], e = offsets[key + 1];
    g1x_acc acc; acc.init();
#pragma nounroll
    for (uint32_t i = s; i < e; i++) {
        uint32_t en = entries[i];
        g1a q = table[en >> 1];
        if (en & 1u) q.y = neg<FpP>(q.y);
        acc.add(q);
    }
    buckets[key] = acc.to_jac();
}

// sum_{k=1..nb} k * B_k for one (blob, group): 64 lanes x segments of m = nb / 64 buckets
__global__ __launch_bounds__(64) void k_msm_reduce(uint8_t *ws, size_t per_blob, size_t buckets_off, size_t gsum_off, uint32_t nb, uint32_t ngroups) {
    __shared__ g1j buf[64];
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x / ngroups; const uint32_t g = blockIdx.x % ngroups;
    const g1j *buckets = (const g1j *)(ws + b * per_blob + buckets_off) + (uint64_t)g * nb;
    g1j *gsum = (g1j *)(ws + b * per_blob + gsum_off);
    const uint32_t m = nb >= 64 ? nb / 64 : 1;
    const uint32_t lo = lane * m, hi = lo + m;
    g1j s = g1_inf(), w = g1_inf();
    if (lo < nb) {
#pragma nounroll
        for (uint32_t k = hi; k-- > lo;) { s = g1_add(s, buckets[k]); w = g1_add(w, s); }
    }
    // T1 = sum_L w_L
    buf[lane] = w;
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 32; off >= 1; off >>= 1) {
        if (lane < off) buf[lane] = g1_add(buf[lane], buf[lane + off]);
        __syncthreads();
    }
    g1j t1 = buf[0];
    __syncthreads();
    // T2 = sum_L L * s_L
    buf[lane] = g1_mul_small(s, lane);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 32; off >= 1; off >>= 1) {
        if (lane < off) buf[lane] = g1_add(buf[lane], buf[lane + off]);
        __syncthreads();
    }
    if (lane == 0) gsum[g] = g1_add(t1, g1_mul_small(buf[0], m));
}

__global__ __launch_bounds__(64) void k_msm_combine(uint8_t *ws, size_t per_blob, size_t gsum_off, uint32_t c, uint32_t ngroups, uint64_t batch, g1j *out) {
    uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const g1j *gsum = (const g1j *)(ws + b * per_blob + gsum_off);
    g1j acc = g1_inf();
#pragma nounroll
    for (uint32_t g = ngroups; g-- > 0;) {
#pragma nounroll
        for (uint32_t j = 0; j < c; j++) acc = g1_dbl(acc);
        acc = g1_add(acc, gsum[g]);
    }
    out[b] = acc;
}

void launch_msm(hipStream_t s, const msm_plan &p, const g1a *table, const fr *scalars, uint64_t n, uint64_t batch, void *workspace, g1j *out) {
    if (!batch) return;
    msm_ws_layout L = ws_layout(p, n);
    uint8_t *ws = (uint8_t *)workspace;
    uint32_t K = (uint32_t)L.K;
    size_t sh = (size_t)(K + MSM_SORT_T) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_msm_sort), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_msm_sort, dim3((uint32_t)batch), dim3(MSM_SORT_T), sh, s, p, scalars, n, ws, L.per_blob, L.entries_off, L.offsets_off, K);
    uint64_t total = batch * L.K;
    prof_begin(s, "msm_accumulate");
    hipLaunchKernelGGL(k_msm_accumulate, dim3((uint32_t)((total + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, table, ws, L.per_blob,
                       L.entries_off, L.offsets_off, L.buckets_off, K, total);
    prof_end(s, "msm_accumulate");
    hipLaunchKernelGGL(k_msm_reduce, dim3((uint32_t)(batch * p.ngroups)), dim3(64), 0, s, ws, L.per_blob, L.buckets_off, L.gsum_off, p.nb, p.ngroups);
    hipLaunchKernelGGL(k_msm_combine, dim3((uint32_t)((batch + 63) / 64)), dim3(64), 0, s, ws, L.per_blob, L.gsum_off, p.c, p.ngroups, batch, out);
}

// ---------------------------------------------------------------------------------------------------------
// Fixed-base MSM for the device-resident setup (CommitToPoly / ComputeProofSingle on KZGSettings.SecretG1).
// HBM is 288 GB, so the table holds EVERY signed-digit multiple:  T[(w n + i) D + d - 1] = d * 2^(c w) * P_i,
// d = 1..D = 2^(c-1), affine.  A commitment is then n * nwin mixed additions with no doublings, no buckets, no
// sort, and exactly the same work on every lane (the bucket method's Poisson imbalance and reduce step vanish).
// n = 4096, c = 11: 24 x 4096 x 1024 x 96 B = 9.7 GB, 98 304 mixed adds per commitment.
// ---------------------------------------------------------------------------------------------------------
#define FB_BLOCK 128

// pass 1: lane (w, i, seg) walks d = seg S + 1 .. (seg + 1) S with mixed additions from (seg S + 1) b (a short double-and-add);
// X, Y go to the table slot, Z to ztmp.  S = D / segs keeps >= 4 waves per SIMD busy even for the 65 536-row n = 4096 tables.
__global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_build_pass1(const g1a *rows, uint64_t lanes, uint32_t D, uint32_t S, g1a *table, fp *ztmp) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint32_t segs = D / S;
    if (t >= lanes * segs) return;
    const uint64_t row = t / segs; const uint32_t seg = (uint32_t)(t % segs);
    g1a b = rows[row];
    g1j cur = to_jac(b);
    uint32_t k = seg * S + 1;
    if (k > 1) {
        cur = g1_inf();
#pragma nounroll
        for (int bit = 31 - __builtin_clz(k); bit >= 0; bit--) {
            cur = g1_dbl(cur);
            if ((k >> bit) & 1u) cur = g1_madd(cur, b);
        }
    }
    g1a *dst = table + row * D + (uint64_t)seg * S; fp *zd = ztmp + row * D + (uint64_t)seg * S;
#pragma nounroll
    for (uint32_t d = 0; d < S; d++) {
        g1a xy; xy.x = cur.x; xy.y = cur.y;
        dst[d] = xy; zd[d] = cur.z;
        cur = g1_madd(cur, b);
    }
}
// pass 2: Montgomery batch inversion of the lane's S Z-values (prefix products in ptmp), then X/Z^2, Y/Z^3
__global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_build_pass2(uint64_t lanes, uint32_t D, uint32_t S, g1a *table, fp *ztmp, fp *ptmp) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= lanes * (D / S)) return;
    g1a *dst = table + t * S; fp *zd = ztmp + t * S; fp *pd = ptmp + t * S;   // (row, seg) slots are contiguous: row D + seg S = t S
    fp acc = one<FpP>();
#pragma nounroll
    for (uint32_t d = 0; d < S; d++) {
        pd[d] = acc;
        fp z = zd[d];
        if (!is_zero<FpP>(z)) acc = mul(acc, z);
    }
    fp inv_all = inv<FpP>(acc);
#pragma nounroll
    for (uint32_t d = S; d-- > 0;) {
        fp z = zd[d];
        if (is_zero<FpP>(z)) { dst[d] = g1a_inf(); continue; }
        fp zi = mul(inv_all, pd[d]);
        inv_all = mul(inv_all, z);
        fp zi2 = sqr(zi);
        g1a xy = dst[d];
        xy.x = mul(xy.x, zi2); xy.y = mul(xy.y, mul(zi2, zi));
        dst[d] = xy;
    }
}

// main kernel: lane handles points i = lane, lane + L, ... of one blob; block tree-reduces through LDS
__global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_accumulate(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                            uint64_t n, uint32_t blocks_per_blob, g1j *partials) {
    __shared__ g1j buf[FB_BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint64_t blob = blockIdx.x / blocks_per_blob; const uint32_t blk = blockIdx.x % blocks_per_blob;
    const uint64_t L = (uint64_t)blocks_per_blob * FB_BLOCK;
    const fr *sc = scalars + blob * n;
    g1x_acc acc; acc.init();   // XYZZ, unpacked lazy limbs: 10 products per mixed addition, no pack / reduce per product
    for (uint64_t i = (uint64_t)blk * FB_BLOCK + tid; i < n; i += L) {
        fr k = from_mont<FrP>(sc[i]);
        uint32_t carry = 0;
#pragma nounroll
        for (uint32_t w = 0; w < nwin; w++) {
            uint32_t raw = scalar_bits(k, w * c, c) + carry;
            uint32_t mag, ng;
            if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
            if (mag) {
                g1a q = table[((uint64_t)w * table_n + i) * D + (mag - 1)];
                if (ng) q.y = neg<FpP>(q.y);
                acc.add(q);
            }
        }
    }
    buf[tid] = acc.to_jac();
    __syncthreads();
#pragma nounroll
    for (uint32_t off = FB_BLOCK / 2; off >= 1; off >>= 1) {
        if (tid < off) buf[tid] = g1_add(buf[tid], buf[tid + off]);
        __syncthreads();
    }
    if (tid == 0) partials[blockIdx.x] = buf[0];
}
// ---------------------------------------------------------------------------------------------------------
// Batch-affine variant of the walk (KZG_HIP_FB_MODE=ba; the XYZZ walk above is the default).  The 16 table entries of a
// point are first summed pairwise in AFFINE coordinates over two levels (16 -> 8 -> 4 sums per point) with ONE field
// inversion per lane and level (Montgomery's trick over all pairs of the lane: the forward pass keeps a running product of
// the x differences and stores its prefixes, the backward pass turns the inverse of the total into per-pair inverses);
// only the 4 sums per point go through the XYZZ accumulator.  An affine addition is 5 products + 1 squaring + its share of
// the inversion (~31 k instructions over 128 / 64 pairs at 16 points per lane) against 8 + 2 for the mixed addition.
//   * level-1 backward and level-2 forward are fused (the x difference of two consecutive level-1 sums feeds the level-2
//     running product as soon as both exist), so level 2 has no forward pass over memory;
//   * gathers are issued one pair (backward) or four pairs (forward) ahead of their use;
//   * values stay unpacked and lazily reduced (fq); bounds of (x, y): table (1,1) -> level 1 (6,4) -> level 2 (16,7);
//   * infinity (zero digit, point beyond n) is the all-zero (x, y): a lazily reduced coordinate of a real sum is never
//     all-zero because every subq adds a positive multiple of p;
//   * P == +-Q: a pair with dx == 0 (mod p) zeroes the lane's running product, the inversion returns 0, and the lane
//     falls back to the plain XYZZ walk over its table entries.
// Workspace: limb planes, word k of element q of lane T at ((q W + k) LT + T): every access is a coalesced dword stream.
// ---------------------------------------------------------------------------------------------------------
struct ba_pt { fq x, y; };
__device__ __forceinline__ fq ba_ld(const uint32_t *ws, uint64_t q, uint32_t W, uint32_t k0, uint64_t LT, uint64_t T) {
    fq o;
#pragma unroll
    for (int k = 0; k < 13; k++) o.l[k] = ws[(q * W + k0 + k) * LT + T];
    return o;
}
__device__ __forceinline__ void ba_st(uint32_t *ws, uint64_t q, uint32_t W, uint32_t k0, uint64_t LT, uint64_t T, const fq &v) {
#pragma unroll
    for (int k = 0; k < 13; k++) ws[(q * W + k0 + k) * LT + T] = v.l[k];
}
__device__ __forceinline__ bool ba_is_inf(const fq &x) {
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) z |= x.l[k];
    return z == 0;
}
__device__ __forceinline__ fq ba_zero() {
    fq o;
#pragma unroll
    for (int k = 0; k < 13; k++) o.l[k] = 0;
    return o;
}
// A + B for two finite points with distinct x; t = 1 / (xB - xA).  Input bounds (BX, BY) -> output (2 BX + 4, BY + 3).
template <int BX, int BY> __device__ __forceinline__ ba_pt ba_pair(const ba_pt &A, const ba_pt &B, const fq &t) {
    fq dy = subq<BY + 1>(B.y, A.y);                       // 2 BY + 1
    fq lam = mulq(dy, t);
    fq l2 = sqrq(lam);
    ba_pt o;
    o.x = subq<BX + 1>(subq<BX + 1>(l2, A.x), B.x);       // 2 + 2 (BX + 1)
    o.y = subq<BY + 1>(mulq(lam, subq<2 * BX + 5>(A.x, o.x)), A.y);   // (3 BX + 5) * 2 <= 600; result 2 + BY + 1
    return o;
}
// XYZZ accumulator fed with lazily reduced affine coordinates (bounds (16, 7): products 32, 14 <= 600)
__device__ __forceinline__ void ba_acc_add(g1x_acc &acc, const ba_pt &q) {
    if (ba_is_inf(q.x)) return;
    if (acc.inf) {
        const fq one_q = unpackq(one<FpP>());
        acc.v.x = mulq(q.x, one_q); acc.v.y = mulq(q.y, one_q); acc.v.zz = one_q; acc.v.zzz = one_q;   // same values, bound 2
        acc.inf = false;
        return;
    }
    if (g1x_madd_fast(acc.v, q.x, q.y)) return;
    g1a qa; qa.x = packq(q.x); qa.y = packq(q.y);
    g1x sgen = g1x_madd(g1xq_pack(acc.v), qa);            // P == +-Q: generic, complete formulas
    if (is_inf(sgen)) acc.inf = true; else acc.v = g1xq_unpack(sgen);
}
// running product -> its inverse (unpacked); false when the product was 0 mod p
__device__ __forceinline__ bool ba_invert(const fq &run, fq &inv_out) {
    fp ri = inv<FpP>(packq(run));
    inv_out = unpackq(ri);
    return !is_zero<FpP>(ri);
}
// carries of the signed-digit recoding as a bit mask: bit w = carry INTO window w
__device__ __forceinline__ uint64_t ba_carry_mask(const fr &k, uint32_t c, uint32_t nwin, uint32_t D) {
    uint64_t cm = 0; uint32_t carry = 0;
#pragma nounroll
    for (uint32_t w = 0; w + 1 < nwin; w++) {
        carry = (scalar_bits(k, w * c, c) + carry > D) ? 1u : 0u;
        cm |= (uint64_t)carry << (w + 1);
    }
    return cm;
}
__device__ __forceinline__ void ba_digit(const fr &k, uint64_t cm, uint32_t w, uint32_t c, uint32_t nwin, uint32_t D, uint32_t &mag, uint32_t &ng) {
    if (w >= nwin) { mag = 0; ng = 0; return; }
    uint32_t raw = scalar_bits(k, w * c, c) + (uint32_t)((cm >> w) & 1u);
    if (raw > D) { mag = (1u << c) - raw; ng = 1; } else { mag = raw; ng = 0; }
}
// table entry of window w (clamped to the padded range), point i, magnitude mag; mag == 0 reads entry 1 (valid memory, unused)
__device__ __forceinline__ const g1a *ba_entry(const g1a *table, uint64_t table_n, uint32_t D, uint32_t nwin, uint32_t w, uint64_t i, uint32_t mag) {
    const uint32_t ww = w < nwin ? w : nwin - 1;
    return table + ((uint64_t)ww * table_n + i) * D + (mag ? mag - 1 : 0);
}
__global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_accumulate_ba(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                               uint64_t n, uint32_t blocks_per_blob, g1j *partials, uint32_t *ws_pa, uint32_t *ws_pb,
                                                               uint32_t *ws_l1, uint32_t *ws_l2) {
    __shared__ g1j buf[FB_BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint64_t blob = blockIdx.x / blocks_per_blob; const uint32_t blk = blockIdx.x % blocks_per_blob;
    const uint64_t L = (uint64_t)blocks_per_blob * FB_BLOCK;
    const uint64_t T = (uint64_t)blockIdx.x * FB_BLOCK + tid, LT = (uint64_t)gridDim.x * FB_BLOCK;
    const fr *sc = scalars + blob * n;
    const uint32_t P = (uint32_t)((n + L - 1) / L);        // points per lane
    const uint32_t halfw = ((nwin + 7) & ~7u) / 2;         // window pairs per point (windows padded to a multiple of 8 with inf)
    const uint32_t N1 = P * halfw;
    const uint64_t i_base = (uint64_t)blk * FB_BLOCK + tid;
    bool ok = true;
    {   // ---- level 1 forward: running product of the x differences, four pairs of gathers in flight ----
        fq run = unpackq(one<FpP>());
#pragma nounroll
        for (uint32_t s_ = 0; s_ < P; s_++) {
            const uint64_t i = i_base + (uint64_t)s_ * L;
            if (i >= n) continue;
            const fr k = from_mont<FrP>(sc[i]);
            const uint64_t cm = ba_carry_mask(k, c, nwin, D);
#pragma nounroll
            for (uint32_t v0 = 0; v0 < halfw; v0 += 4) {
                uint32_t m0[4], m1[4], sg;
                fp xa[4], xb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    ba_digit(k, cm, 2 * (v0 + q), c, nwin, D, m0[q], sg); ba_digit(k, cm, 2 * (v0 + q) + 1, c, nwin, D, m1[q], sg);
                    xa[q] = ba_entry(table, table_n, D, nwin, 2 * (v0 + q), i, m0[q])->x;
                    xb[q] = ba_entry(table, table_n, D, nwin, 2 * (v0 + q) + 1, i, m1[q])->x;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!m0[q] || !m1[q]) continue;
                    ba_st(ws_pa, s_ * halfw + v0 + q, 13, 0, LT, T, run);
                    run = mulq(run, subq<2>(unpackq(xb[q]), unpackq(xa[q])));
                }
            }
        }
        fq I;
        ok = ba_invert(run, I);
        // ---- level 1 backward (descending) fused with level 2 forward: sums to ws_l1, level-2 prefixes to ws_pb ----
        fq run2 = unpackq(one<FpP>());
        fq prev_x = ba_zero();
#pragma nounroll
        for (uint32_t s_ = P; s_-- > 0;) {
            const uint64_t i = i_base + (uint64_t)s_ * L;
            const bool valid = i < n;
            fr k; uint64_t cm = 0;
            if (valid) { k = from_mont<FrP>(sc[i]); cm = ba_carry_mask(k, c, nwin, D); }
            uint32_t nm0 = 0, nn0 = 0, nm1 = 0, nn1 = 0;
            g1a nA, nB;
            if (valid) {
                ba_digit(k, cm, 2 * (halfw - 1), c, nwin, D, nm0, nn0); ba_digit(k, cm, 2 * (halfw - 1) + 1, c, nwin, D, nm1, nn1);
                nA = *ba_entry(table, table_n, D, nwin, 2 * (halfw - 1), i, nm0); nB = *ba_entry(table, table_n, D, nwin, 2 * (halfw - 1) + 1, i, nm1);
            }
#pragma nounroll
            for (uint32_t v = halfw; v-- > 0;) {
                const uint32_t j = s_ * halfw + v;
                const uint32_t m0 = nm0, n0 = nn0, m1 = nm1, n1 = nn1;
                g1a qa = nA, qb = nB;
                if (valid && v > 0) {                      // gathers of the next pair, one addition ahead
                    ba_digit(k, cm, 2 * (v - 1), c, nwin, D, nm0, nn0); ba_digit(k, cm, 2 * (v - 1) + 1, c, nwin, D, nm1, nn1);
                    nA = *ba_entry(table, table_n, D, nwin, 2 * (v - 1), i, nm0); nB = *ba_entry(table, table_n, D, nwin, 2 * (v - 1) + 1, i, nm1);
                }
                ba_pt A, B, o;
                A.x = ba_zero(); A.y = A.x; B = A;
                if (m0) { if (n0) qa.y = neg<FpP>(qa.y); A.x = unpackq(qa.x); A.y = unpackq(qa.y); }
                if (m1) { if (n1) qb.y = neg<FpP>(qb.y); B.x = unpackq(qb.x); B.y = unpackq(qb.y); }
                if (m0 && m1) {
                    fq t = mulq(I, ba_ld(ws_pa, j, 13, 0, LT, T));
                    I = mulq(I, subq<2>(B.x, A.x));
                    o = ba_pair<1, 1>(A, B, t);
                } else o = m0 ? A : B;                      // B is the all-zero infinity when both digits are zero
                ba_st(ws_l1, j, 26, 0, LT, T, o.x); ba_st(ws_l1, j, 26, 13, LT, T, o.y);
                if (v & 1) prev_x = o.x;                    // level-2 pair (j - 1, j): x difference = x_j - x_(j-1)
                else if (!ba_is_inf(o.x) && !ba_is_inf(prev_x)) {
                    ba_st(ws_pb, j >> 1, 13, 0, LT, T, run2);
                    run2 = mulq(run2, subq<7>(prev_x, o.x));
                }
            }
        }
        // ---- level 2 backward (ascending: the reverse of its forward order) ----
        fq I2;
        const bool ok2 = ba_invert(run2, I2);
        ok = ok && ok2;
        const uint32_t N2 = N1 / 2;
        ba_pt nA2, nB2;
        nA2.x = ba_ld(ws_l1, 0, 26, 0, LT, T); nA2.y = ba_ld(ws_l1, 0, 26, 13, LT, T);
        nB2.x = ba_ld(ws_l1, 1, 26, 0, LT, T); nB2.y = ba_ld(ws_l1, 1, 26, 13, LT, T);
#pragma nounroll
        for (uint32_t j2 = 0; j2 < N2; j2++) {
            ba_pt A = nA2, B = nB2, o;
            if (j2 + 1 < N2) {
                nA2.x = ba_ld(ws_l1, 2 * j2 + 2, 26, 0, LT, T); nA2.y = ba_ld(ws_l1, 2 * j2 + 2, 26, 13, LT, T);
                nB2.x = ba_ld(ws_l1, 2 * j2 + 3, 26, 0, LT, T); nB2.y = ba_ld(ws_l1, 2 * j2 + 3, 26, 13, LT, T);
            }
            const bool ia = ba_is_inf(A.x), ib = ba_is_inf(B.x);
            if (!ia && !ib) {
                fq t = mulq(I2, ba_ld(ws_pb, j2, 13, 0, LT, T));
                I2 = mulq(I2, subq<7>(B.x, A.x));
                o = ba_pair<6, 4>(A, B, t);
            } else o = ia ? B : A;
            ba_st(ws_l2, j2, 26, 0, LT, T, o.x); ba_st(ws_l2, j2, 26, 13, LT, T, o.y);
        }
    }
    g1x_acc acc; acc.init();
    if (ok) {   // ---- the 4 sums per point through the XYZZ accumulator ----
        const uint32_t N2 = N1 / 2;
        ba_pt nx; nx.x = ba_ld(ws_l2, 0, 26, 0, LT, T); nx.y = ba_ld(ws_l2, 0, 26, 13, LT, T);
#pragma nounroll
        for (uint32_t j2 = 0; j2 < N2; j2++) {
            ba_pt q = nx;
            if (j2 + 1 < N2) { nx.x = ba_ld(ws_l2, j2 + 1, 26, 0, LT, T); nx.y = ba_ld(ws_l2, j2 + 1, 26, 13, LT, T); }
            ba_acc_add(acc, q);
        }
    } else {    // ---- some dx was 0 mod p in this lane: plain walk over its table entries ----
#pragma nounroll
        for (uint32_t s_ = 0; s_ < P; s_++) {
            const uint64_t i = i_base + (uint64_t)s_ * L;
            if (i >= n) continue;
            const fr k = from_mont<FrP>(sc[i]);
            const uint64_t cm = ba_carry_mask(k, c, nwin, D);
#pragma nounroll
            for (uint32_t w = 0; w < nwin; w++) {
                uint32_t mag, ng;
                ba_digit(k, cm, w, c, nwin, D, mag, ng);
                if (!mag) continue;
                g1a q = *ba_entry(table, table_n, D, nwin, w, i, mag);
                if (ng) q.y = neg<FpP>(q.y);
                acc.add(q);
            }
        }
    }
    buf[tid] = acc.to_jac();
    __syncthreads();
#pragma nounroll
    for (uint32_t off = FB_BLOCK / 2; off >= 1; off >>= 1) {
        if (tid < off) buf[tid] = g1_add(buf[tid], buf[tid + off]);
        __syncthreads();
    }
    if (tid == 0) partials[blockIdx.x] = buf[0];
}
__global__ __launch_bounds__(64) void k_fb_finish(const g1j *partials, uint32_t blocks_per_blob, uint64_t batch, g1j *out) {
    uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= batch) return;
    g1j acc = g1_inf();
#pragma nounroll
    for (uint32_t j = 0; j < blocks_per_blob; j++) acc = g1_add(acc, partials[b * blocks_per_blob + j]);
    out[b] = acc;
}

// element-wise fixed-base products over the same table layout: out[b][i] = scalars[b][i] * P_i  (the FK20 Toeplitz stage,
// ToeplitzPart2's loop fk20_single.go:72-74, where P = xExtFFT is fixed per settings): nwin mixed adds instead of a
// 255-bit double-and-add per element.
__global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_mul_vec(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                         uint64_t i0, uint64_t cnt, uint64_t row, uint64_t total, g1j *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    // t enumerates (b, f, jj): point index i = f * k2 + j0 + jj is supplied through (row = k2, i0 = j0, cnt) by the caller
    uint64_t jj = t % cnt, f = (t / cnt) % (table_n / row), b = t / (cnt * (table_n / row));
    uint64_t i = f * row + i0 + jj;
    fr k = from_mont<FrP>(scalars[b * table_n + i]);
    g1x_acc acc; acc.init();
    uint32_t carry = 0;
#pragma nounroll
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t raw = scalar_bits(k, w * c, c) + carry;
        uint32_t mag, ng;
        if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
        if (mag) {
            g1a q = table[((uint64_t)w * table_n + i) * D + (mag - 1)];
            if (ng) q.y = neg<FpP>(q.y);
            acc.add(q);
        }
    }
    out[t] = acc.to_jac();
}
void launch_fb_mul_vec(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t row, uint64_t j0,
                       uint64_t cnt, uint64_t batch, g1j *out) {
    uint64_t total = batch * (table_n / row) * cnt;
    if (!total) return;
    prof_begin(s, "fb_mul_vec");
    hipLaunchKernelGGL(k_fb_mul_vec, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                       j0, cnt, row, total, out);
    prof_end(s, "fb_mul_vec");
}

static uint32_t fb_blocks_per_blob(uint64_t n, uint64_t batch) {
    // 131072 lanes = 2048 wavefronts = exactly the resident capacity at 2 waves per SIMD: ONE round.  Measured (512 blobs):
    // 131072 lanes 5.7 ms, 262144 lanes (two rounds) 6.4 ms, 98304 / 65536 lanes 10.4 ms.  KZG_HIP_FB_LANES overrides.
    static uint64_t lanes = 0;
    if (!lanes) { const char *e = getenv("KZG_HIP_FB_LANES"); lanes = e ? strtoull(e, nullptr, 10) : 131072; if (lanes < FB_BLOCK) lanes = FB_BLOCK; }
    uint64_t target = lanes / FB_BLOCK;
    uint64_t bpb = target / (batch ? batch : 1);
    uint64_t maxb = (n + FB_BLOCK - 1) / FB_BLOCK;
    if (bpb > maxb) bpb = maxb;
    if (bpb < 1) bpb = 1;
    return (uint32_t)bpb;
}
static bool fb_batch_affine() {                           // read per call so that tests can switch it
    const char *e = getenv("KZG_HIP_FB_MODE");
    return e && e[0] == 'b';
}
// per lane: level-1 prefixes (N1 fq), level-2 prefixes (N1 / 2 fq), level-1 sums (N1 points), level-2 sums (N1 / 2 points)
static size_t fb_ba_words(uint64_t n, uint64_t batch, uint32_t nwin, size_t w[4]) {
    uint32_t bpb = fb_blocks_per_blob(n, batch);
    uint64_t LT = (uint64_t)bpb * batch * FB_BLOCK, L = (uint64_t)bpb * FB_BLOCK;
    uint64_t P = (n + L - 1) / L, N1 = P * (((nwin + 7) & ~7u) / 2);
    w[0] = N1 * 13 * LT; w[1] = (N1 / 2) * 13 * LT; w[2] = N1 * 26 * LT; w[3] = (N1 / 2) * 26 * LT;
    return w[0] + w[1] + w[2] + w[3];
}
size_t fb_partials_bytes(uint64_t n, uint64_t batch, uint32_t nwin) {
    size_t part = ((size_t)fb_blocks_per_blob(n, batch) * batch * sizeof(g1j) + 255) & ~(size_t)255;
    if (!fb_batch_affine()) return part;
    size_t w[4];
    return part + fb_ba_words(n, batch, nwin, w) * sizeof(uint32_t);
}

void launch_fb_msm(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t n, uint64_t batch,
                   void *partials, g1j *out) {
    if (!batch) return;
    uint32_t bpb = fb_blocks_per_blob(n, batch);
    prof_begin(s, "fb_accumulate");
    if (fb_batch_affine()) {
        size_t part = ((size_t)bpb * batch * sizeof(g1j) + 255) & ~(size_t)255, w[4];
        fb_ba_words(n, batch, nwin, w);
        uint32_t *wp = (uint32_t *)((uint8_t *)partials + part);
        hipLaunchKernelGGL(k_fb_accumulate_ba, dim3((uint32_t)(batch * bpb)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, n, bpb,
                           (g1j *)partials, wp, wp + w[0], wp + w[0] + w[1], wp + w[0] + w[1] + w[2]);
    } else
        hipLaunchKernelGGL(k_fb_accumulate, dim3((uint32_t)(batch * bpb)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, n, bpb,
                           (g1j *)partials);
    prof_end(s, "fb_accumulate");
    hipLaunchKernelGGL(k_fb_finish, dim3((uint32_t)((batch + 63) / 64)), dim3(64), 0, s, (const g1j *)partials, bpb, batch, out);
}
// builds the table for `n` affine points: rows (2^(c w) P_i) first, then all multiples window-slab by window-slab
hipError_t launch_fb_build(hipStream_t s, const g1a *pts, uint64_t n, uint32_t c, uint32_t nwin, g1a *table) {
    uint32_t D = 1u << (c - 1);
    g1j *rows_j = nullptr; g1a *rows = nullptr; fp *ztmp = nullptr, *ptmp = nullptr;
    uint64_t lanes_all = (uint64_t)nwin * n;
    hipError_t e;
    if ((e = hipMallocAsync((void **)&rows_j, lanes_all * sizeof(g1j), s)) != hipSuccess) return e;
    if ((e = hipMallocAsync((void **)&rows, lanes_all * sizeof(g1a), s)) != hipSuccess) { hipFreeAsync(rows_j, s); return e; }
    launch_msm_window_table(s, pts, n, c, nwin, rows_j, rows);
    // slabs of windows bound the temporary Z / prefix storage to ~2 x 1 GiB
    uint64_t per_win = n * D * sizeof(fp);
    uint32_t slab = (uint32_t)((1ull << 30) / (per_win ? per_win : 1));
    if (slab < 1) slab = 1;
    if (slab > nwin) slab = nwin;
    if ((e = hipMallocAsync((void **)&ztmp, (uint64_t)slab * per_win, s)) != hipSuccess) { hipFreeAsync(rows_j, s); hipFreeAsync(rows, s); return e; }
    if ((e = hipMallocAsync((void **)&ptmp, (uint64_t)slab * per_win, s)) != hipSuccess) { hipFreeAsync(rows_j, s); hipFreeAsync(rows, s); hipFreeAsync(ztmp, s); return e; }
    for (uint32_t w0 = 0; w0 < nwin; w0 += slab) {
        uint32_t ws = (w0 + slab <= nwin) ? slab : nwin - w0;
        uint64_t lanes = (uint64_t)ws * n;
        uint32_t S = D;                                      // segment length: split rows until the slab has >= 2^20 lanes, S >= 64
        while (S > 64 && lanes * (D / S) < (1ull << 20)) S >>= 1;
        dim3 g((uint32_t)((lanes * (D / S) + FB_BLOCK - 1) / FB_BLOCK)), b(FB_BLOCK);
        hipLaunchKernelGGL(k_fb_build_pass1, g, b, 0, s, rows + (uint64_t)w0 * n, lanes, D, S, table + (uint64_t)w0 * n * D, ztmp);
        hipLaunchKernelGGL(k_fb_build_pass2, g, b, 0, s, lanes, D, S, table + (uint64_t)w0 * n * D, ztmp, ptmp);
    }
    hipFreeAsync(rows_j, s); hipFreeAsync(rows, s); hipFreeAsync(ztmp, s); hipFreeAsync(ptmp, s);
    return hipGetLastError();
}

// fixed-base table rows: tmp[w * n + i] = 2^(c w) * P_i (Jacobian), then normalised to affine by launch_g1_to_affine
__global__ __launch_bounds__(MSM_ACC_BLOCK, 2) void k_msm_window_rows(const g1a *pts, uint64_t n, uint32_t c, uint32_t nwin, g1j *tmp) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1j q = to_jac(pts[i]);
#pragma nounroll
    for (uint32_t w = 0; w < nwin; w++) {
        tmp[(uint64_t)w * n + i] = q;
#pragma nounroll
        for (uint32_t j = 0; j < c; j++) q = g1_dbl(q);
    }
}
void launch_msm_window_table(hipStream_t s, const g1a *pts, uint64_t n, uint32_t c, uint32_t nwin, g1j *tmp, g1a *out) {
    if (!n) return;
    hipLaunchKernelGGL(k_msm_window_rows, dim3((uint32_t)((n + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, pts, n, c, nwin, tmp);
    launch_g1_to_affine(s, tmp, out, n * nwin);
}

}  // namespace kzg
