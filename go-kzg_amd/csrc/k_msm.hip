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
#include "g1_quad.hpp"
#include "coop_inv.hpp"
#include <atomic>
#include <cstring>

namespace kzg {

#define MSM_SORT_T 1024
#define MSM_ACC_BLOCK 128
#define MSM_NB 128                      // buckets per window group: signed 8-bit digits, |d| in 1..128

// a lane's XYZZ accumulator as it travels through LDS and the workspaces: the 4 x 13 lazy limbs as they are (no pack / unpack /
// reduction between the walks and the trees) + the infinity flag.  53 words: an odd stride, so LDS trees have no bank conflicts.
struct fb_partial { uint32_t w[52]; uint32_t inf; };
__device__ __forceinline__ void fb_partial_store(fb_partial &o, const g1x_acc &a) {
#pragma unroll
    for (int i = 0; i < 13; i++) { o.w[i] = a.v.x.l[i]; o.w[13 + i] = a.v.y.l[i]; o.w[26 + i] = a.v.zz.l[i]; o.w[39 + i] = a.v.zzz.l[i]; }
    o.inf = a.inf ? 1u : 0u;
}
__device__ __forceinline__ void fb_partial_load(const fb_partial &p, g1xq &v) {
#pragma unroll
    for (int i = 0; i < 13; i++) { v.x.l[i] = p.w[i]; v.y.l[i] = p.w[13 + i]; v.zz.l[i] = p.w[26 + i]; v.zzz.l[i] = p.w[39 + i]; }
}

// ---------------------------------------------------------------------------------------------------------
// Wave-level exchange for the reduction trees (the "wavefront shuffles" of the north star): lane i receives what lane i + DELTA of the
// SAME wavefront holds -- DPP row shifts (v_mov_b32 row_shl) inside a row of 16 lanes, ds_bpermute_b32 (the LDS crossbar, no LDS memory)
// across rows; 53 moves per lazy XYZZ point, no LDS round trip, no workgroup barrier.  ROW = true (tree levels: only lanes i < DELTA <= 32
// use the result, so a DPP shift never has to leave its row of 16); ROW = false (scans: every lane i with i + DELTA < 64 uses it):
// ds_bpermute at every distance.  Lanes whose source falls outside receive an unspecified value.
// KZG_NO_WAVE_SHUFFLE restores the LDS + __syncthreads exchange of round 2 (A/B builds, profiles/r03_wave_shuffle_ab.md); the bucket
// scan of k_msm_reduce keeps the LDS form by default (KZG_MSM_REDUCE_SHUFFLE selects the shuffle form: measured 3 % slower).
// ---------------------------------------------------------------------------------------------------------
template <uint32_t DELTA, bool ROW> __device__ __forceinline__ uint32_t lane_down(uint32_t v) {
    if constexpr (ROW && DELTA < 16) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x100 + DELTA, 0xf, 0xf, false);   // row_shl:DELTA
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & 63u) + DELTA) & 63u) << 2), (int)v);
}
template <uint32_t DELTA, bool ROW> __device__ __forceinline__ void point_down(g1xq &o, uint32_t &oinf, const g1xq &v, uint32_t vinf) {
#pragma unroll
    for (int i = 0; i < 13; i++) {
        o.x.l[i] = lane_down<DELTA, ROW>(v.x.l[i]); o.y.l[i] = lane_down<DELTA, ROW>(v.y.l[i]);
        o.zz.l[i] = lane_down<DELTA, ROW>(v.zz.l[i]); o.zzz.l[i] = lane_down<DELTA, ROW>(v.zzz.l[i]);
    }
    oinf = lane_down<DELTA, ROW>(vinf);
}
// off in {1, 2, 4, 8, 16, 32}; wave-uniform
template <bool ROW = true> __device__ __forceinline__ void point_down_any(g1xq &o, uint32_t &oinf, const g1xq &v, uint32_t vinf, uint32_t off) {
    switch (off) {
    case 1: point_down<1, ROW>(o, oinf, v, vinf); break;
    case 2: point_down<2, ROW>(o, oinf, v, vinf); break;
    case 4: point_down<4, ROW>(o, oinf, v, vinf); break;
    case 8: point_down<8, ROW>(o, oinf, v, vinf); break;
    case 16: point_down<16, ROW>(o, oinf, v, vinf); break;
    default: point_down<32, ROW>(o, oinf, v, vinf); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variable-base MSM (bls.LinCombG1 on caller-supplied points, bls/bls_kilic.go:132-150): Pippenger buckets over the GLV halves.
//   k_i P_i = s1 |k1| P_i + s2 |k2| phi(P_i), |k1|, |k2| < 2^126.5 (glv_split_signed): 16 signed 8-bit windows per half, no carry
//   out of the top one.  ngroups = 16: one bucket group per window (120 doublings in the final Horner);  ngroups = 8: the table
//   also holds 2^64 P_i (a cached point set, kzg_hip_points_new), window w >= 8 of a half uses it and the Horner has 56 doublings.
// Pipeline per blob:  sort (workgroup per blob, LDS histogram + scatter; keys (bucket, GLV half), the phi half first)  ->  accumulate (small batches:
// S lanes per bucket walk a share of its list with mixed additions, LDS merge; batches that fill the GPU: a lane per segment of 64 sorted entries,
// k_msm_accumulate_seg + k_msm_merge_segs)  ->  reduce (128 lanes per window group: suffix scan + tree, sum_d d B_d = sum_m
// sum_{d >= m} B_d)  ->  combine (Horner over the groups, then normalise).  Everything between the kernels is lazy XYZZ limbs.
// Sums are taken in a data-dependent order; the group law is commutative and the result is normalised, so the output bytes do
// not depend on it.
// ---------------------------------------------------------------------------------------------------------
#ifndef MSM_SEG
#define MSM_SEG 64                      // entries per lane of the balanced accumulate (k_msm_accumulate_seg)
#endif
struct msm_ws_layout {
    size_t entries_off, offsets_off, buckets_off, gsum_off, segs_off, per_blob;
    uint64_t K, nent, nseg;
};
static msm_ws_layout ws_layout(const msm_plan &p, uint64_t n) {
    msm_ws_layout L;
    L.K = (uint64_t)p.ngroups * MSM_NB;
    L.nent = n * 32;                                      // 2 halves x 16 windows
    size_t o = 0;
    L.entries_off = o; o += ((L.nent * 4 + 15) / 16) * 16;
    L.offsets_off = o; o += (((2 * L.K + 1) * 4 + 15) / 16) * 16;   // one offset per (bucket, GLV half): key 2 b holds the phi half, key 2 b + 1 the plain one
    L.buckets_off = o; o += ((L.K * sizeof(fb_partial) + 15) / 16) * 16;
    L.gsum_off = o; o += (((size_t)p.ngroups * sizeof(fb_partial) + 15) / 16) * 16;
    L.nseg = (L.nent + MSM_SEG - 1) / MSM_SEG;
    L.segs_off = o; o += ((L.nseg * 2 * sizeof(fb_partial) + 15) / 16) * 16;   // two boundary partial sums per segment
    L.per_blob = o;
    return L;
}
size_t msm_workspace_bytes(const msm_plan &p, uint64_t n, uint64_t batch) { return ws_layout(p, n).per_blob * batch; }

// visits every non-zero signed digit of the GLV halves of scalar k (Montgomery form): f(window 0..15, |digit| 1..128, half, negative)
template <class Fn> __device__ __forceinline__ void for_each_glv_digit(const fr &k_mont, Fn f) {
    const glv_halves h = glv_split_signed(from_mont<FrP>(k_mont));   // Kilic FromRed (bls_kilic.go:141-147), then the split
#pragma unroll
    for (uint32_t half = 0; half < 2; half++) {
        const uint32_t *mag = half ? h.k2 : h.k1;
        const uint32_t sgn = half ? h.neg2 : h.neg1;
        uint32_t carry = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) {
            uint32_t raw = ((mag[w >> 2] >> ((w & 3) * 8)) & 255u) + carry;
            if (raw > MSM_NB) { carry = 1; f(w, 256u - raw, half, sgn ^ 1u); }       // raw <= 256: 256 - raw in 0..127; 0 -> digit 0, carry 1
            else { carry = 0; f(w, raw, half, sgn); }
        }
    }
}

__global__ __launch_bounds__(MSM_SORT_T) void k_msm_sort(uint32_t ngroups, uint64_t table_n, const fr *scalars, uint64_t sc_stride, uint64_t n, uint8_t *ws,
                                                         size_t per_blob, size_t entries_off, size_t offsets_off, uint32_t K) {
    extern __shared__ uint32_t smem[];
    uint32_t *hist = smem, *part = smem + K;
    const uint32_t tid = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const fr *sc = scalars + b * sc_stride;
    uint32_t *entries = (uint32_t *)(ws + b * per_blob + entries_off);
    uint32_t *offsets = (uint32_t *)(ws + b * per_blob + offsets_off);
    const uint32_t wmask = ngroups - 1;                     // 15: a group per window; 7: windows 8..15 fold onto 0..7 (2^64 P_i rows)
    // K = 2 x buckets here: within a bucket the entries of the phi half (key 2 b) come before the plain ones (key 2 b + 1), so that a walk adds
    // the phi half un-mapped first and applies phi ONCE to the running sum (phi is an endomorphism: phi(P1) + phi(P2) = phi(P1 + P2))
    for (uint32_t i = tid; i < K; i += MSM_SORT_T) hist[i] = 0;
    __syncthreads();
    for (uint64_t i = tid; i < n; i += MSM_SORT_T)
        for_each_glv_digit(sc[i], [&](uint32_t w, uint32_t mag, uint32_t half, uint32_t) {
            if (mag) atomicAdd(&hist[(((w & wmask) * MSM_NB + mag - 1) << 1) | (half ^ 1u)], 1u);
        });
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
    for (uint64_t i = tid; i < n; i += MSM_SORT_T)
        for_each_glv_digit(sc[i], [&](uint32_t w, uint32_t mag, uint32_t half, uint32_t neg) {
            if (!mag) return;
            uint32_t slot = atomicAdd(&hist[(((w & wmask) * MSM_NB + mag - 1) << 1) | (half ^ 1u)], 1u);
            uint32_t pidx = (uint32_t)i + ((w & ~wmask) ? (uint32_t)table_n : 0u);      // window >= 8 of a folded plan: the 2^64 P_i row
            entries[slot] = (pidx << 2) | (half << 1) | neg;
        });
}

// S lanes per (blob, bucket): each walks its share of the bucket's sorted list, then the S partial sums are merged through LDS
__global__ __launch_bounds__(MSM_ACC_BLOCK, 2) void k_msm_accumulate(const g1a *table, uint8_t *ws, size_t per_blob, size_t entries_off, size_t offsets_off,
                                                                  size_t buckets_off, uint32_t K, uint32_t S, uint64_t total) {
    __shared__ fb_partial buf[MSM_ACC_BLOCK];
    const uint32_t tid = threadIdx.x;
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + tid;
    const bool live = t < total;
    const uint32_t sidx = (uint32_t)(t % S);
    const uint64_t bk = t / S;
    const uint64_t b = bk / K; const uint32_t key = (uint32_t)(bk % K);
    g1x_acc acc; acc.init();
    if (live) {
        const uint32_t *entries = (const uint32_t *)(ws + b * per_blob + entries_off);
        const uint32_t *offsets = (const uint32_t *)(ws + b * per_blob + offsets_off);
        const uint32_t s0 = offsets[2 * key], len = offsets[2 * key + 2] - s0;
        const uint32_t s = s0 + (uint32_t)((uint64_t)len * sidx / S), e = s0 + (uint32_t)((uint64_t)len * (sidx + 1) / S);
#pragma nounroll
        for (uint32_t i = s; i < e; i++) {
            const uint32_t en = entries[i];
            g1a q = table[en >> 2];
            if (en & 2u) q.x = mul(q.x, glv_beta());       // phi(x, y) = (beta x, y)
            if (en & 1u) q.y = neg<FpP>(q.y);
            acc.add(q);
        }
    }
    if (S > 1) {                                           // block-uniform: S divides the block size
        fb_partial_store(buf[tid], acc);
        __syncthreads();
#pragma nounroll
        for (uint32_t off = S / 2; off >= 1; off >>= 1) {
            if (sidx < off) {
                g1xq v; fb_partial_load(buf[tid + off], v);
                g1x_acc_merge(acc, v, buf[tid + off].inf != 0);
                fb_partial_store(buf[tid], acc);
            }
            __syncthreads();
        }
    }
    if (live && sidx == 0) fb_partial_store(((fb_partial *)(ws + b * per_blob + buckets_off))[key], acc);
}

// Balanced form for batches that fill the GPU on their own: a lane per SEGMENT of MSM_SEG consecutive sorted entries instead of a lane per
// bucket.  Bucket lists are Poisson-distributed (mean 64 entries at n = 4096: a wavefront of 64 bucket lanes waits for a list of ~84), segments are
// all equal.  A lane walks its entries bucket by bucket: buckets that lie wholly inside the segment are stored straight to the bucket array; the
// bucket of the first entry and the bucket of the last entry may continue in the neighbouring segments -- their partial sums go to the segment's two
// slots and k_msm_merge_segs adds them up per bucket (on average one addition per bucket).  Inside a bucket the phi half comes first (k_msm_sort):
// it is accumulated un-mapped and phi -- X <- beta X on the XYZZ sum -- is applied once, when the walk crosses into the plain half or leaves the bucket.
__device__ __forceinline__ void acc_apply_phi(g1x_acc &a) { if (!a.inf) a.v.x = mulq(a.v.x, unpackq(glv_beta())); }
__global__ __launch_bounds__(MSM_ACC_BLOCK, 2) void k_msm_accumulate_seg(const g1a *table, uint8_t *ws, size_t per_blob, size_t entries_off, size_t offsets_off,
                                                                      size_t buckets_off, size_t segs_off, uint32_t K2, uint64_t nseg, uint64_t total) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t b = t / nseg; const uint32_t seg = (uint32_t)(t % nseg);
    const uint32_t *entries = (const uint32_t *)(ws + b * per_blob + entries_off);
    const uint32_t *offsets = (const uint32_t *)(ws + b * per_blob + offsets_off);
    fb_partial *buckets = (fb_partial *)(ws + b * per_blob + buckets_off);
    fb_partial *slots = (fb_partial *)(ws + b * per_blob + segs_off) + 2ull * seg;
    const uint32_t E = offsets[K2];
    const uint32_t s = seg * MSM_SEG, e = s + MSM_SEG < E ? s + MSM_SEG : E;
    if (s >= e) return;                                    // beyond the blob's entries (zero digits are not entries): no bucket reaches into this segment
    // key of the first entry: the last key whose offset is <= s (empty keys share an offset with their successor: take the last one)
    uint32_t lo = 0, hi = K2;                              // invariant: offsets[lo] <= s < offsets[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (offsets[mid] <= s) lo = mid; else hi = mid; }
    uint32_t key = lo, kend = offsets[key + 1];
    const uint32_t first_bucket = key >> 1;
    const bool first_open = offsets[key & ~1u] < s;        // the first bucket started in an earlier segment
    g1x_acc acc; acc.init();
    uint32_t en_next = entries[s];
    g1a qn = table[en_next >> 2];
#pragma nounroll
    for (uint32_t i = s; i < e; i++) {
        while (i >= kend) {                                // leave key (possibly across several empty keys)
            if (!(key & 1u)) acc_apply_phi(acc);           // end of a phi half: map the sum
            if (key & 1u) {                                // end of a bucket
                const uint32_t bk = key >> 1;
                if (bk == first_bucket && first_open) fb_partial_store(slots[0], acc); else fb_partial_store(buckets[bk], acc);
                acc.init();
            }
            key++; kend = offsets[key + 1];
        }
        g1a q = qn; const uint32_t en = en_next;
        if (i + 1 < e) { en_next = entries[i + 1]; qn = table[en_next >> 2]; }   // the next gather is in flight during this addition
        if (en & 1u) q.y = neg<FpP>(q.y);
        acc.add(q);
    }
    // the bucket of the last entry: closed here if the segment holds its end, else a boundary partial
    if (!(key & 1u)) acc_apply_phi(acc);
    const uint32_t bk = key >> 1;
    const bool last_closed = offsets[2 * bk + 2] <= e;
    // slot 0: the segment's first bucket unless it lies wholly inside; slot 1: its last bucket (if another one) when it continues beyond
    if (bk == first_bucket) { if (!first_open && last_closed) fb_partial_store(buckets[bk], acc); else fb_partial_store(slots[0], acc); }
    else if (last_closed) fb_partial_store(buckets[bk], acc);
    else fb_partial_store(slots[1], acc);
}
// a lane per (blob, bucket): buckets that straddle segment boundaries are the sum of their segments' boundary partials; empty buckets are infinity;
// buckets inside one segment were stored by the walk
__global__ __launch_bounds__(MSM_ACC_BLOCK, 2) void k_msm_merge_segs(uint8_t *ws, size_t per_blob, size_t offsets_off, size_t buckets_off, size_t segs_off, uint32_t K, uint64_t total) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t b = t / K; const uint32_t bk = (uint32_t)(t % K);
    const uint32_t *offsets = (const uint32_t *)(ws + b * per_blob + offsets_off);
    fb_partial *buckets = (fb_partial *)(ws + b * per_blob + buckets_off);
    const fb_partial *slots = (const fb_partial *)(ws + b * per_blob + segs_off);
    const uint32_t a = offsets[2 * bk], z = offsets[2 * bk + 2];
    g1x_acc acc; acc.init();
    if (a == z) { fb_partial_store(buckets[bk], acc); return; }
    const uint32_t seg_lo = a / MSM_SEG, seg_hi = (z - 1) / MSM_SEG;
    if (seg_lo == seg_hi) return;
#pragma nounroll
    for (uint32_t sg = seg_lo; sg <= seg_hi; sg++) {
        const fb_partial &p = slots[2ull * sg + ((sg == seg_lo && a > seg_lo * MSM_SEG) ? 1 : 0)];
        g1xq v; fb_partial_load(p, v);
        g1x_acc_merge(acc, v, p.inf != 0);
    }
    fb_partial_store(buckets[bk], acc);
}

// sum_{d=1..128} d * B_d for one (blob, window group) = sum_m T_m with the suffix sums T_m = sum_{d >= m} B_d: a Hillis-Steele
// suffix scan over the 128 lanes (7 steps) and a tree sum (7 steps), every step one lazy XYZZ + XYZZ addition
__global__ __launch_bounds__(MSM_NB) void k_msm_reduce(uint8_t *ws, size_t per_blob, size_t buckets_off, size_t gsum_off, uint32_t ngroups) {
    __shared__ fb_partial buf[MSM_NB];
    const uint32_t d = threadIdx.x;
    const uint64_t b = blockIdx.x / ngroups; const uint32_t g = blockIdx.x % ngroups;
    const fb_partial *buckets = (const fb_partial *)(ws + b * per_blob + buckets_off) + (uint64_t)g * MSM_NB;
    g1x_acc acc; acc.inf = buckets[d].inf != 0;
    if (!acc.inf) fb_partial_load(buckets[d], acc.v);
#if defined(KZG_NO_WAVE_SHUFFLE) || !defined(KZG_MSM_REDUCE_SHUFFLE)   // measured: the shuffle form below is 3 % SLOWER for a lone LinCombG1 (profiles/r03_wave_shuffle_ab.md)
    fb_partial_store(buf[d], acc);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 1; off < MSM_NB; off <<= 1) {
        g1xq v; bool vinf = true;
        if (d + off < MSM_NB) { vinf = buf[d + off].inf != 0; if (!vinf) fb_partial_load(buf[d + off], v); }
        __syncthreads();
        if (!vinf) { g1x_acc_merge(acc, v, false); fb_partial_store(buf[d], acc); }
        __syncthreads();
    }
#pragma nounroll
    for (uint32_t off = MSM_NB / 2; off >= 1; off >>= 1) {
        if (d < off) {
            g1xq v; fb_partial_load(buf[d + off], v);
            g1x_acc_merge(acc, v, buf[d + off].inf != 0);
            fb_partial_store(buf[d], acc);
        }
        __syncthreads();
    }
#else
    // The 128 buckets of a group sit on two wavefronts.  Suffix scan INSIDE each wavefront by lane shuffles (6 steps, no LDS, no barrier),
    // then the lower wavefront adds the upper one's total (its lane 0) -- the one exchange through LDS; the tree sum likewise: 6 shuffle
    // levels per wavefront, then lane 0 adds the other wavefront's sum.  16 dependent additions as before, 2 barriers instead of 42.
    const uint32_t lane = d & 63u;
    if (acc.inf) acc.v = g1xq_from_affine(g1a_inf());      // defined limbs for the shuffles
#pragma nounroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        g1xq v; uint32_t vi;
        point_down_any<false>(v, vi, acc.v, acc.inf ? 1u : 0u, off);
        if (lane + off < 64) g1x_acc_merge(acc, v, vi != 0);
    }
    if (d == 64) fb_partial_store(buf[0], acc);            // T_64 = the sum of the upper 64 buckets
    __syncthreads();
    if (d < 64) { g1xq v; fb_partial_load(buf[0], v); g1x_acc_merge(acc, v, buf[0].inf != 0); }
    // acc = T_d (suffix sums); the answer is their sum
#pragma nounroll
    for (uint32_t off = 32; off >= 1; off >>= 1) {
        g1xq v; uint32_t vi;
        point_down_any(v, vi, acc.v, acc.inf ? 1u : 0u, off);
        if (lane < off) g1x_acc_merge(acc, v, vi != 0);
    }
    __syncthreads();                                       // buf[0] has been read by everyone
    if (d == 64) fb_partial_store(buf[0], acc);
    __syncthreads();
    if (d == 0) { g1xq v; fb_partial_load(buf[0], v); g1x_acc_merge(acc, v, buf[0].inf != 0); }
#endif
    if (d == 0) fb_partial_store(((fb_partial *)(ws + b * per_blob + gsum_off))[g], acc);
}

// p <- 2 p on one lane (dbl-2008-s-1, a = 0; the formulas and bounds of coop_xyzz_dbl below: (X, Y, ZZ, ZZZ) <= (11, 5, 2, 2) in and out)
__device__ __forceinline__ void g1xq_dbl_lane(g1xq &p) {
    const fq u = addq(p.y, p.y);
    const fq v = sqrq(u), xx = sqrq(p.x);
    const fq m = addq(addq(xx, xx), xx);
    const fq w = mulq(u, v), s_ = mulq(p.x, v), zz3 = mulq(v, p.zz), mm = sqrq(m);
    const fq x3 = subq<5>(mm, addq(s_, s_));
    const fq t1 = mulq(m, subq<8>(s_, x3)), t2 = mulq(w, p.y), zzz3 = mulq(w, p.zzz);
    p.x = x3; p.y = subq<3>(t1, t2); p.zz = zz3; p.zzz = zzz3;
}
// Throughput form of the reduce (batches that fill the GPU): the scan above spends 14 additions on EVERY bucket; here a lane owns 4 consecutive
// buckets d = 4 c + 1 .. 4 c + 4 and runs the classic double running sum on them (T_c = sum B, S_c = sum j B_{4c+j}: 8 additions), then
//   sum_d d B_d = sum_c S_c + 4 sum_{c >= 1} SUF_c,   SUF_c = sum_{c' >= c} T_c'
// -- a suffix scan of the T's over the group's 32 lanes (5 steps), two doublings, one addition and a tree (5 steps): 21 operations on 32 lanes per
// group instead of 14 on 128 (2.7x less work; the chain is longer, which a lone MSM would pay: it keeps the scan).  Four (blob, group) pairs per workgroup.
__global__ __launch_bounds__(MSM_NB) void k_msm_reduce_chunks(uint8_t *ws, size_t per_blob, size_t buckets_off, size_t gsum_off, uint32_t ngroups, uint64_t total_groups) {
    __shared__ fb_partial buf[MSM_NB];
    const uint32_t tid = threadIdx.x, c = tid & 31u;
    const uint64_t G = blockIdx.x * 4ull + (tid >> 5);
    const bool live = G < total_groups;
    const uint64_t b = live ? G / ngroups : 0; const uint32_t g = live ? (uint32_t)(G % ngroups) : 0;
    const fb_partial *buckets = (const fb_partial *)(ws + b * per_blob + buckets_off) + (uint64_t)g * MSM_NB + 4u * c;
    g1x_acc T, S; T.init(); S.init();
    if (live) {
#pragma nounroll
        for (int j = 3; j >= 0; j--) {
            const bool vinf = buckets[j].inf != 0;
            g1xq v;
            if (!vinf) { fb_partial_load(buckets[j], v); g1x_acc_merge(T, v, false); }
            if (!T.inf) { const g1xq t = T.v; g1x_acc_merge(S, t, false); }
        }
    }
    fb_partial_store(buf[tid], T);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 1; off < 32; off <<= 1) {           // suffix scan of T inside the group's 32 lanes
        g1xq v; bool vinf = true;
        if (c + off < 32) { vinf = buf[tid + off].inf != 0; if (!vinf) fb_partial_load(buf[tid + off], v); }
        __syncthreads();
        if (!vinf) { g1x_acc_merge(T, v, false); fb_partial_store(buf[tid], T); }
        __syncthreads();
    }
    if (c >= 1 && !T.inf) {                                 // V_c = S_c + 4 SUF_c
        g1xq q = T.v;
        g1xq_dbl_lane(q); g1xq_dbl_lane(q);
        g1x_acc_merge(S, q, false);
    }
    fb_partial_store(buf[tid], S);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = 16; off >= 1; off >>= 1) {
        if (c < off) {
            g1xq v; fb_partial_load(buf[tid + off], v);
            g1x_acc_merge(S, v, buf[tid + off].inf != 0);
            fb_partial_store(buf[tid], S);
        }
        __syncthreads();
    }
    if (live && c == 0) fb_partial_store(((fb_partial *)(ws + b * per_blob + gsum_off))[g], S);
}

// ---------------------------------------------------------------------------------------------------------
// Wave-cooperative XYZZ arithmetic for the serial tails.  A chain of dependent group operations on ONE point (the Horner over the
// window groups: 120 doublings) is bound by the latency of one F_p product after the other on one SIMD.  The products INSIDE a
// doubling / addition are mostly independent, so the four wavefronts of a 256-thread workgroup (one per SIMD of the CU) each take
// one product of a dependency level and exchange the results through LDS: a doubling is 3 levels deep instead of 9 products, an
// addition 4 instead of 13.  Lane column c of every wave works on the same point c (64 points per workgroup); all four waves keep
// a full replica of the running point, so control flow stays identical across them.
// LDS: two alternating sets of four 13-limb slots per column, limb-major (conflict-free), one barrier per level.
// ---------------------------------------------------------------------------------------------------------
struct coop_lds { uint32_t x[2][4][13][64]; };
struct coop_ctx {
    coop_lds *L; uint32_t wave, col, set;
    __device__ __forceinline__ void put(const fq &v) {
#pragma unroll
        for (int i = 0; i < 13; i++) L->x[set][wave][i][col] = v.l[i];
    }
    __device__ __forceinline__ fq get(uint32_t slot) const {
        fq v;
#pragma unroll
        for (int i = 0; i < 13; i++) v.l[i] = L->x[set][slot][i][col];
        return v;
    }
    __device__ __forceinline__ void exchange() { __syncthreads(); }          // after put(), before get(): one barrier per level
    __device__ __forceinline__ void next() { set ^= 1u; }                    // the following level writes the other set
};
// p <- 2 p (dbl-2008-s-1, a = 0) on lazy limbs, bounds (X, Y, ZZ, ZZZ) <= (11, 5, 2, 2) in and out:
//   U = 2 Y : 10;  V = U^2, XX = X^2 : 2 (100, 121 <= 600);  M = 3 XX : 6;  W = U V, S = X V, ZZ' = V ZZ, MM = M^2 : 2
//   X' = MM - 2 S (M = 5) : 7;  T1 = M (S - X') (S - X' with M = 8 : 10, 60), T2 = W Y : 2;  Y' = T1 - T2 (M = 3) : 5;  ZZZ' = W ZZZ : 2
__device__ __forceinline__ void coop_xyzz_dbl(g1xq &p, coop_ctx &c) {
    const fq u = addq(p.y, p.y);
    fq mine;
    if (c.wave == 0) mine = sqrq(u); else if (c.wave == 1) mine = sqrq(p.x); else mine = u;
    c.put(mine); c.exchange();
    const fq v = c.get(0), xx = c.get(1);
    c.next();
    const fq m = addq(addq(xx, xx), xx);
    if (c.wave == 0) mine = mulq(u, v); else if (c.wave == 1) mine = mulq(p.x, v); else if (c.wave == 2) mine = mulq(v, p.zz); else mine = sqrq(m);
    c.put(mine); c.exchange();
    const fq w = c.get(0), s_ = c.get(1), zz3 = c.get(2), mm = c.get(3);
    c.next();
    const fq x3 = subq<5>(mm, addq(s_, s_));
    if (c.wave == 0) mine = mulq(m, subq<8>(s_, x3)); else if (c.wave == 1) mine = mulq(w, p.y); else mine = mulq(w, p.zzz);
    c.put(mine); c.exchange();
    const fq t1 = c.get(0), t2 = c.get(1), zzz3 = c.get(2);
    c.next();
    p.x = x3; p.y = subq<3>(t1, t2); p.zz = zz3; p.zzz = zzz3;
}
// a <- a + b (add-2008-s), bounds as in g1xq_add_fast.  Returns false when P == 0 (equal / opposite operands): `a` is then
// untouched and the caller takes the generic path.  All four waves compute the same verdict.
__device__ __forceinline__ bool coop_xyzz_add(g1xq &a, const g1xq &b, coop_ctx &c) {
    fq mine;
    if (c.wave == 0) mine = mulq(a.x, b.zz); else if (c.wave == 1) mine = mulq(b.x, a.zz); else if (c.wave == 2) mine = mulq(a.y, b.zzz); else mine = mulq(b.y, a.zzz);
    c.put(mine); c.exchange();
    const fq u1 = c.get(0), u2 = c.get(1), s1 = c.get(2), s2 = c.get(3);
    c.next();
    const fq pp_ = subq<3>(u2, u1), r = subq<3>(s2, s1);
    if (c.wave == 0) mine = sqrq(pp_); else if (c.wave == 1) mine = sqrq(r); else if (c.wave == 2) mine = mulq(a.zz, b.zz); else mine = mulq(a.zzz, b.zzz);
    c.put(mine); c.exchange();
    const fq pp = c.get(0), rr = c.get(1), zz12 = c.get(2), zzz12 = c.get(3);
    c.next();
    const bool ok = !is_zero_mod_p_q(pp);
    if (c.wave == 0) mine = mulq(pp_, pp); else if (c.wave == 1) mine = mulq(u1, pp); else mine = mulq(zz12, pp);
    c.put(mine); c.exchange();
    const fq ppp = c.get(0), q_ = c.get(1), zz3 = c.get(2);
    c.next();
    const fq x3 = subq<3>(subq<3>(subq<3>(rr, ppp), q_), q_);
    if (c.wave == 0) mine = mulq(r, subq<12>(q_, x3)); else if (c.wave == 1) mine = mulq(s1, ppp); else mine = mulq(zzz12, ppp);
    c.put(mine); c.exchange();
    const fq t1 = c.get(0), t2 = c.get(1), zzz3 = c.get(2);
    c.next();
    if (ok) { a.x = x3; a.y = subq<3>(t1, t2); a.zz = zz3; a.zzz = zzz3; }
    return ok;
}

// acc += w for the replicated accumulators of a cooperating workgroup; called by all 256 threads (barriers inside).  Columns with
// nothing to add (winf) still run the addition, on a private copy of whatever `w` holds, and drop the result.
__device__ __forceinline__ void coop_acc_add(g1x_acc &acc, const g1xq &w, bool winf, coop_ctx &c) {
    g1xq sum = acc.v, addend = w;                          // private copies: the operands must not move while the levels exchange
    const bool ok = coop_xyzz_add(sum, addend, c);
    if (winf) return;
    if (acc.inf) { acc.v = addend; acc.inf = false; }
    else if (ok) acc.v = sum;
    else g1x_acc_merge(acc, addend, false);                // equal / opposite operands: generic complete formulas, identical on the four waves
}

// Horner over the window groups (8 doublings per group), then normalise and convert.  The doublings are the critical path of a
// lone MSM (120 for 16 groups, 56 for 8): workgroup = 4 cooperating waves, lane column = blob (64 blobs per workgroup).
#ifdef KZG_COMBINE_WAVE_COOP                                 // round 2's form: four WAVEFRONTS per blob column, exchange through LDS (A/B builds)
__global__ __launch_bounds__(256) void k_msm_combine(uint8_t *ws, size_t per_blob, size_t gsum_off, uint32_t ngroups, uint64_t batch, g1j *out, int to_kilic) {
    __shared__ coop_lds lds;
    coop_ctx c; c.L = &lds; c.wave = threadIdx.x >> 6; c.col = threadIdx.x & 63u; c.set = 0;
    const uint64_t b = blockIdx.x * 64ull + c.col;
    const bool live = b < batch;
    const fb_partial *gsum = (const fb_partial *)(ws + (live ? b : 0) * per_blob + gsum_off);
    g1x_acc acc; acc.init();
    acc.v = g1xq_from_affine(g1a_inf());                   // defined limbs while the accumulator is still empty (results are discarded)
#pragma nounroll
    for (uint32_t g = ngroups; g-- > 0;) {
#pragma nounroll
        for (uint32_t j = 0; j < 8; j++) {                 // barriers inside: every thread runs the doubling, empty accumulators ignore it
            g1xq d = acc.v;
            coop_xyzz_dbl(d, c);
            if (!acc.inf) acc.v = d;
        }
        const bool winf = !live || gsum[g].inf != 0;
        g1xq w;
        if (winf) w = acc.v; else fb_partial_load(gsum[g], w);
        coop_acc_add(acc, w, winf, c);
    }
    if (live && c.wave == 0) {
        g1j r;
        if (acc.inf) r = g1_inf();
        else {   // x = X / ZZ, y = Y / ZZZ with one inversion
            g1x px = g1xq_pack(acc.v);
            fp i = inv<FpP>(mul(px.zz, px.zzz));
            r.x = mul(px.x, mul(i, px.zzz)); r.y = mul(px.y, mul(i, px.zz)); r.z = one<FpP>();
        }
        out[b] = to_kilic ? g1_to_kilic(r) : r;
    }
}
#else
// Quad form (g1_quad.hpp): the four LANES of a quad hold the replicas and exchange the products of a level by DPP broadcasts -- no LDS, no barrier
// (a level 1.3 us instead of 1.84).  64 blobs per 256-lane workgroup, as before.
__global__ __launch_bounds__(256) void k_msm_combine(uint8_t *ws, size_t per_blob, size_t gsum_off, uint32_t ngroups, uint64_t batch, g1j *out, int to_kilic) {
    const uint32_t role = threadIdx.x & 3u;
    const uint64_t b = blockIdx.x * 64ull + (threadIdx.x >> 2);
    const bool live = b < batch;
    const fb_partial *gsum = (const fb_partial *)(ws + (live ? b : 0) * per_blob + gsum_off);
    g1x_acc acc; acc.init();
    acc.v = g1xq_from_affine(g1a_inf());                   // defined limbs while the accumulator is still empty (results are discarded)
#pragma nounroll
    for (uint32_t g = ngroups; g-- > 0;) {
#pragma nounroll
        for (uint32_t j = 0; j < 8; j++) {                 // every lane runs the doubling, empty accumulators ignore it
            g1xq d = acc.v;
            quad_xyzz_dbl(d, role);
            if (!acc.inf) acc.v = d;
        }
        const bool winf = !live || gsum[g].inf != 0;
        g1xq w;
        if (winf) w = acc.v; else fb_partial_load(gsum[g], w);
        quad_acc_add(acc, w, winf, role);
    }
    {   // x = X / ZZ, y = Y / ZZZ with one inversion per blob; a wavefront with up to three live blobs (a lone LinCombG1: one) inverts them cooperatively over its
        // lanes, a full one (16 blobs) in its lanes side by side (wave_inv_any decides; every lane of the wavefront is here)
        const bool own = live && role == 0;
        const g1x px = g1xq_pack(acc.v);
        const fp i = wave_inv_any(mul(px.zz, px.zzz), own && !acc.inf);
        if (own) {
            g1j r;
            if (acc.inf) r = g1_inf();
            else { r.x = mul(px.x, mul(i, px.zzz)); r.y = mul(px.y, mul(i, px.zz)); r.z = one<FpP>(); }
            out[b] = to_kilic ? g1_to_kilic(r) : r;
        }
    }
}
#endif

void launch_msm(hipStream_t s, const msm_plan &p, const g1a *table, const fr *scalars, uint64_t sc_stride, uint64_t n, uint64_t batch, void *workspace, g1j *out,
                bool to_kilic) {
    if (!batch) return;
    msm_ws_layout L = ws_layout(p, n);
    uint8_t *ws = (uint8_t *)workspace;
    uint32_t K = (uint32_t)L.K;
    const uint32_t K2 = 2 * K;                                // sort keys: (bucket, GLV half)
    size_t sh = (size_t)(K2 + MSM_SORT_T) * 4;
    hipLaunchKernelGGL(k_msm_sort, dim3((uint32_t)batch), dim3(MSM_SORT_T), sh, s, p.ngroups, p.table_n, scalars, sc_stride, n, ws, L.per_blob, L.entries_off,
                       L.offsets_off, K2);
    // lanes per bucket: fill one round of resident wavefronts (131 072 lanes) when the batch alone does not, at most 16
    uint32_t S = 1;
    while (S < 16 && batch * L.K * S * 2 <= 2 * device_simd_lanes()) S *= 2;
    uint64_t total = batch * L.K * S;
    static const int seg_mode = [] { const char *e = getenv("KZG_HIP_MSM_SEG"); return e ? atoi(e) : -1; }();   // 0 / 1: never / always the balanced form (A/B runs, tests)
    prof_begin(s, "msm_accumulate");
    // from one full round of resident lanes on (64 MSMs of 4096 points): measured 2.19 vs 2.98 ms at 64, 12.3 vs 17.3 ms at 512; below, the S-lanes-per-bucket
    // form wins (32: 1.89 vs 2.07 ms, 8: 1.14 vs 1.47 ms: its lanes are shorter and a lone MSM is latency-bound)
    const bool balanced = seg_mode == 1 || (seg_mode < 0 && batch * L.nseg >= 2 * device_simd_lanes());
    if (balanced) {
        const uint64_t tseg = batch * L.nseg, tb = batch * L.K;
        hipLaunchKernelGGL(k_msm_accumulate_seg, dim3((uint32_t)((tseg + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, table, ws, L.per_blob,
                           L.entries_off, L.offsets_off, L.buckets_off, L.segs_off, K2, L.nseg, tseg);
        hipLaunchKernelGGL(k_msm_merge_segs, dim3((uint32_t)((tb + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, ws, L.per_blob, L.offsets_off,
                           L.buckets_off, L.segs_off, K, tb);
    } else
        hipLaunchKernelGGL(k_msm_accumulate, dim3((uint32_t)((total + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, table, ws, L.per_blob,
                           L.entries_off, L.offsets_off, L.buckets_off, K, S, total);
    prof_end(s, "msm_accumulate");
    static const int reduce_mode = [] { const char *e = getenv("KZG_HIP_MSM_REDUCE"); return !e ? -1 : !strcmp(e, "chunks") ? 1 : !strcmp(e, "scan") ? 0 : -1; }();   // tests / A/B runs: force a form at any batch size
    if (reduce_mode == 1 || (reduce_mode < 0 && balanced && batch * p.ngroups * 32 >= device_simd_lanes())) {   // (256 MSMs on a cached set: 37.7 k -> 40.7 k MSM/s, 512: 40.4 k -> 43.4 k; 64 MSMs are faster on the scan: 29.3 k vs 27.6 k)
        const uint64_t tg = batch * p.ngroups;
        hipLaunchKernelGGL(k_msm_reduce_chunks, dim3((uint32_t)((tg + 3) / 4)), dim3(MSM_NB), 0, s, ws, L.per_blob, L.buckets_off, L.gsum_off, p.ngroups, tg);
    } else
        hipLaunchKernelGGL(k_msm_reduce, dim3((uint32_t)(batch * p.ngroups)), dim3(MSM_NB), 0, s, ws, L.per_blob, L.buckets_off, L.gsum_off, p.ngroups);
    hipLaunchKernelGGL(k_msm_combine, dim3((uint32_t)((batch + 63) / 64)), dim3(256), 0, s, ws, L.per_blob, L.gsum_off, p.ngroups, batch, out, to_kilic ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------
// Fixed-base MSM for the device-resident setup (CommitToPoly / ComputeProofSingle on KZGSettings.SecretG1).
// HBM is 288 GB, so the table holds EVERY signed-digit multiple:  T[(w n + i) D + d - 1] = d * 2^(c w) * P_i,
// d = 1..D = 2^(c-1), affine.  A commitment is then n * nwin mixed additions with no doublings, no buckets, no
// sort, and exactly the same work on every lane (the bucket method's Poisson imbalance and reduce step vanish).
// n = 4096, c = 11: 24 x 4096 x 1024 x 96 B = 9.7 GB, 98 304 mixed adds per commitment.
// ---------------------------------------------------------------------------------------------------------
#ifndef FB_BLOCK
#define FB_BLOCK 128
#endif
#ifndef FB_FINISH_LANES_MAX
#define FB_FINISH_LANES_MAX 4          // partial sums per blob up to which one lane per blob finishes (A/B builds: 0 = always the cooperative kernel)
#endif
#ifndef FB_KEEP
#define FB_KEEP 16          // plain halves a lane of the GLV walk keeps between its phi pass and its plain pass (4096 coefficients on one 256-lane workgroup: 16)
#endif
#ifndef FB_ACC_WAVES
#define FB_ACC_WAVES 2
#endif

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

// ---------------------------------------------------------------------------------------------------------
// QUAD-cooperative reduction trees (round 6).  The trees at the end of the walk and in the finish are chains of dependent XYZZ additions: pure latency, and for a lone
// commitment they ARE the call (19 additions on the critical path).  Rounds 2-5 spread the products of an addition over the four WAVEFRONTS of the workgroup
// (coop_xyzz_add: one product per wave and dependency level, results exchanged through LDS, one workgroup barrier per level: 2.2 us per level, 8.8 us per addition).
// The four LANES of a quad do the same with DPP broadcasts -- no LDS, no barrier (g1_quad.hpp, the form k_msm_combine has used since round 3): ~1.5 us per level.
// Column c of the 64 columns of a workgroup is quad c (lanes 4 c .. 4 c + 3, which hold replicas); tree levels fetch column c + off from lane + 4 off of the same
// wavefront (DPP / ds_bpermute) or, for the two levels that cross wavefronts, through 32 + 16 LDS slots with one barrier each.
// ---------------------------------------------------------------------------------------------------------
// (hook for an A/B: the tree sites call the addition through this name)
#ifdef KZG_QUAD_ADD_NOINLINE                                  // A/B builds: one out-of-line copy of the addition for all tree sites (measured: a lone commitment 0.225 -> 0.265 ms,
                                                             // the 4096-blob walk unchanged: the call's traffic through scratch costs more than warm instructions save)
__device__ __noinline__ void quad_acc_add_nl(g1x_acc &acc, const g1xq &w, bool winf, uint32_t role) { quad_acc_add(acc, w, winf, role); }
#else
__device__ __forceinline__ void quad_acc_add_nl(g1x_acc &acc, const g1xq &w, bool winf, uint32_t role) { quad_acc_add(acc, w, winf, role); }
#endif
template <int R> __device__ __forceinline__ void quad_point_bcast(g1xq &o, uint32_t &oinf, const g1xq &v, uint32_t vinf) {
    o.x = quad_bcast<R>(v.x); o.y = quad_bcast<R>(v.y); o.zz = quad_bcast<R>(v.zz); o.zzz = quad_bcast<R>(v.zzz);
    oinf = (uint32_t)__builtin_amdgcn_update_dpp((int)vinf, (int)vinf, R * 0x55, 0xf, 0xf, false);
}
// sum of the first `live` (<= 64) quad columns of a 256-lane workgroup into quad 0 (replicated on its lanes 0..3).  Called by all 256 threads (barriers inside).
__device__ __forceinline__ void quad_tree_reduce(g1x_acc &col, fb_partial *buf, uint32_t tid, uint32_t live) {
    const uint32_t role = tid & 3u, quad = tid >> 2;
    uint32_t off = 1;
    while (off < live) off *= 2;                          // smallest power of two >= live
#pragma nounroll
    for (off >>= 1; off >= 16; off >>= 1) {               // columns off .. 2 off - 1 live in other wavefronts than their receivers: through LDS
        if (quad >= off && quad < 2 * off && role == 0) fb_partial_store(buf[quad], col);
        __syncthreads();
        if (quad < off) {                                 // (wave-uniform: 16 quads per wavefront)
            const bool have = quad + off < live;
            const fb_partial &src = buf[have ? quad + off : off];
            g1xq w; fb_partial_load(src, w);
            quad_acc_add_nl(col, w, !have || src.inf != 0, role);
        }
    }
    if (tid < 64) {                                       // the rest is inside the first wavefront: column c + off is lane + 4 off
#pragma nounroll
        for (; off >= 1; off >>= 1) {
            const bool have = quad < off && quad + off < live;
            g1xq w; uint32_t wi;
            point_down_any(w, wi, col.v, col.inf ? 1u : 0u, 4 * off);
            quad_acc_add_nl(col, w, !have || wi != 0, role);
        }
    }
}
// Block-wide sum of the 256 lane accumulators of a table-walk workgroup into lane 0's: the four accumulators of a quad first (three additions), then the tree
// over the 64 quads: 9 additions like the wave-cooperative form below, each ~6 us instead of 8.8.
__device__ __forceinline__ void fb_block_reduce_quad(g1x_acc &acc, fb_partial *buf, uint32_t tid) {
    const uint32_t role = tid & 3u;
    if (acc.inf) acc.v = g1xq_from_affine(g1a_inf());      // defined limbs in empty accumulators (their additions are computed and dropped)
    const uint32_t ainf = acc.inf ? 1u : 0u;
    g1x_acc col;
    { uint32_t vi; quad_point_bcast<0>(col.v, vi, acc.v, ainf); col.inf = vi != 0; }
    { g1xq v; uint32_t vi; quad_point_bcast<1>(v, vi, acc.v, ainf); quad_acc_add_nl(col, v, vi != 0, role); }
    { g1xq v; uint32_t vi; quad_point_bcast<2>(v, vi, acc.v, ainf); quad_acc_add_nl(col, v, vi != 0, role); }
    { g1xq v; uint32_t vi; quad_point_bcast<3>(v, vi, acc.v, ainf); quad_acc_add_nl(col, v, vi != 0, role); }
    quad_tree_reduce(col, buf, tid, 64);
    acc = col;                                             // quad 0 (lanes 0..3) holds the block's sum
}

// Block-wide sum of the 256 lane accumulators of a table-walk workgroup into lane 0's, WAVE-COOPERATIVELY: while the tree runs, three of
// the four wavefronts would idle, so each takes one product of a dependency level of the XYZZ addition instead (coop_xyzz_add: 4 levels
// deep instead of 13 products).  Lane column c of every wave keeps a replica of column c's running sum: the waves publish their 64
// accumulators one after the other (3 cooperative additions), then a 6-level tree over the columns: 9 additions of depth 4 instead of
// 7 of depth 13 (and 8 with 256 lanes).  The tree is what a small batch pays in full: a walk over 64 polynomials 0.86 -> 0.8x ms.
// buf: 64 slots.
#define FB_ACC_BLOCK 256
__device__ __forceinline__ void fb_block_reduce_coop(g1x_acc &acc, fb_partial *buf, coop_lds *lds, uint32_t tid) {
    coop_ctx c; c.L = lds; c.wave = tid >> 6; c.col = tid & 63u; c.set = 0;
    g1x_acc col; col.init();
#pragma nounroll
    for (uint32_t w = 0; w < FB_ACC_BLOCK / 64; w++) {
        if (c.wave == w) fb_partial_store(buf[c.col], acc);
        __syncthreads();
        g1xq v; fb_partial_load(buf[c.col], v);
        const bool vinf = buf[c.col].inf != 0;
        __syncthreads();                                   // everyone has read before the next wave publishes
        if (w == 0) { col.v = v; col.inf = vinf; }
        else coop_acc_add(col, v, vinf, c);
    }
#pragma nounroll
    for (uint32_t off = 32; off >= 1; off >>= 1) {
        const bool have = c.col < off;
#ifdef KZG_NO_WAVE_SHUFFLE
        if (c.wave == 0) fb_partial_store(buf[c.col], col);   // the replicas are identical: one wave publishes the columns
        __syncthreads();
        const uint32_t src = have ? c.col + off : c.col;
        g1xq w; fb_partial_load(buf[src], w);
        const bool winf = !have || buf[src].inf != 0;
        __syncthreads();
#else
        // every wave holds all 64 columns (identical replicas): column c + off comes from lane c + off of the SAME wave
        g1xq w; uint32_t wi;
        point_down_any(w, wi, col.v, col.inf ? 1u : 0u, off);
        const bool winf = !have || wi != 0;
#endif
        coop_acc_add(col, w, winf, c);
    }
    acc = col;                                             // column 0 (of every wave) holds the block's sum
}
__device__ __forceinline__ uint32_t scalar_bits(const fr &k, uint32_t off, uint32_t c) {
    uint32_t idx = off >> 5, sh = off & 31;
    if (idx >= 8) return 0;
    uint64_t v = k.l[idx];
    if (idx + 1 < 8) v |= (uint64_t)k.l[idx + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// main kernel: lane handles points i = lane, lane + L, ... of one blob; the block reduces cooperatively through LDS.
// SPLIT (small batches: fewer than 32 polynomials would leave most of the 131 072 resident lanes empty): the windows of a point are
// divided among `wsplit` lanes (virtual point v = q n + i walks windows [q wpg, (q + 1) wpg) of point i; the carry into its first window
// comes from a scan of the lower digits, bit operations only), so a lone commitment is 4096 x 8 lanes with 2 additions each instead
// of 4096 lanes with 16: the walk of one polynomial 0.2 -> 0.1 ms.  The unsplit instantiation is the code it was before.
template <bool SPLIT> __global__ __launch_bounds__(FB_ACC_BLOCK, FB_ACC_WAVES) void k_fb_accumulate(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                            uint64_t sc_stride, uint64_t n, uint32_t blocks_per_blob, uint32_t wsplit, fb_partial *partials) {
    __shared__ fb_partial buf[64];
#ifdef KZG_REDUCE_WAVE_COOP
    __shared__ coop_lds lds;
#endif
    const uint32_t tid = threadIdx.x;
    const uint64_t blob = blockIdx.x / blocks_per_blob; const uint32_t blk = blockIdx.x % blocks_per_blob;
    const uint64_t L = (uint64_t)blocks_per_blob * FB_ACC_BLOCK;
    const fr *sc = scalars + blob * sc_stride;   // rows may be wider than n (pinned staging rows read in place over PCIe)
    g1x_acc acc; acc.init();   // XYZZ, unpacked lazy limbs: 10 products per mixed addition, no pack / reduce per product
    const uint32_t wpg = SPLIT ? (nwin + wsplit - 1) / wsplit : nwin;
    const uint64_t nv = SPLIT ? n * wsplit : n;
    for (uint64_t v = (uint64_t)blk * FB_ACC_BLOCK + tid; v < nv; v += L) {
        const uint64_t i = SPLIT ? v % n : v;
        const uint32_t w0 = SPLIT ? (uint32_t)(v / n) * wpg : 0u;
        const uint32_t w1 = SPLIT ? (w0 + wpg < nwin ? w0 + wpg : nwin) : nwin;
        if (SPLIT && w0 >= w1) continue;
        fr k = from_mont<FrP>(sc[i]);
        // software pipeline: the gather of window w + 1 is issued before the addition of window w (+1.3 % measured)
        uint32_t raw, carry = 0, mag, ng;
        if (SPLIT) {
#pragma nounroll
            for (uint32_t w = 0; w < w0; w++) carry = (scalar_bits(k, w * c, c) + carry > D) ? 1u : 0u;
        }
        raw = scalar_bits(k, w0 * c, c) + carry;
        if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
#ifdef KZG_WALK_NO_PREFETCH                                   // A/B builds: gather at the point of use (no register-held next entry)
#pragma nounroll
        for (uint32_t w = w0; w < w1; w++) {
            const uint32_t cmag = mag, cng = ng;
            g1a q = table[((uint64_t)w * table_n + i) * D + (cmag ? cmag - 1 : 0)];
            if (w + 1 < w1) {
                raw = scalar_bits(k, (w + 1) * c, c) + carry;
                if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
            }
            if (cmag) {
#else
        g1a qn = table[((uint64_t)w0 * table_n + i) * D + (mag ? mag - 1 : 0)];
#pragma nounroll
        for (uint32_t w = w0; w < w1; w++) {
            g1a q = qn;
            const uint32_t cmag = mag, cng = ng;
            if (w + 1 < w1) {
                raw = scalar_bits(k, (w + 1) * c, c) + carry;
                if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
                qn = table[((uint64_t)(w + 1) * table_n + i) * D + (mag ? mag - 1 : 0)];
            }
            if (cmag) {
#endif
                if (cng) q.y = neg<FpP>(q.y);
                acc.add(q);
            }
        }
    }
#ifdef KZG_REDUCE_WAVE_COOP                                   // A/B builds: the four-wavefront form of rounds 2-5
    fb_block_reduce_coop(acc, buf, &lds, tid);
#else
    fb_block_reduce_quad(acc, buf, tid);
#endif
    if (tid == 0) fb_partial_store(partials[blockIdx.x], acc);
}
// The same walk over a table HALF as large (round 5): every scalar is split on the device, k = s1 |k1| + s2 |k2| lambda with both magnitudes below
// 2^126.5 (glv_split_signed), and BOTH halves walk the same rows -- phi(d 2^(c w) P) = (beta x, y) of the entry d 2^(c w) P -- so a table of
// nwin = ceil(128 / c) windows serves 2 nwin additions per point: c = 16 is 8 windows = 103 GB for the 16 additions per point that took 16 windows =
// 206 GB.  phi costs nothing per entry: a lane first sums the phi halves of all its points UN-mapped, maps the sum once (phi is an endomorphism:
// X <- beta X on the XYZZ accumulator, one product per lane and blob), then continues with the plain halves.  Virtual points v in [0, nvh) are
// phi halves, v in [nvh, 2 nvh) plain halves (nvh = n, or n * wsplit when the windows of a half are divided among wsplit lanes: small batches).
__device__ __forceinline__ uint32_t mag_bits(const uint32_t (&m)[4], uint32_t off, uint32_t c) {
    const uint32_t idx = off >> 5, sh = off & 31;
    if (idx >= 4) return 0;
    uint64_t v = m[idx];
    if (idx + 1 < 4) v |= (uint64_t)m[idx + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}
template <bool SPLIT> __global__ __launch_bounds__(FB_ACC_BLOCK, FB_ACC_WAVES) void k_fb_accumulate_glv(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                            uint64_t sc_stride, uint64_t n, uint32_t blocks_per_blob, uint32_t wsplit, fb_partial *partials) {
    __shared__ fb_partial buf[64];
#ifdef KZG_REDUCE_WAVE_COOP
    __shared__ coop_lds lds;
#endif
    const uint32_t tid = threadIdx.x;
    const uint64_t blob = blockIdx.x / blocks_per_blob; const uint32_t blk = blockIdx.x % blocks_per_blob;
    const uint64_t L = (uint64_t)blocks_per_blob * FB_ACC_BLOCK;
    const fr *sc = scalars + blob * sc_stride;
    g1x_acc acc; acc.init();
    const uint32_t wpg = SPLIT ? (nwin + wsplit - 1) / wsplit : nwin;
    const uint64_t nvh = SPLIT ? n * wsplit : n;
    bool in_phi = false;                                  // acc holds an un-mapped sum of phi-half entries
    // Each scalar is split ONCE (round 6).  When the lanes of the blob divide its points evenly, the phi half and the plain half of point i land on the SAME lane,
    // nvh / L iterations apart: the lane keeps the plain half (|k1|, its sign: 5 words) of each of its <= FB_KEEP points in a private array while it walks the phi
    // half, and reads it back instead of repeating the Montgomery reduction + Barrett split (two of those per scalar were 1.6 % of the walk).
    const bool keep_ok = !SPLIT && nvh % L == 0 && nvh / L <= FB_KEEP;
    uint32_t kept[FB_KEEP][5];
    uint32_t slot = 0;
    for (uint64_t v = (uint64_t)blk * FB_ACC_BLOCK + tid; v < 2 * nvh; v += L) {
        const bool phi = v < nvh;
        const uint64_t vh = phi ? v : v - nvh;
        const uint64_t i = SPLIT ? vh % n : vh;
        const uint32_t w0 = SPLIT ? (uint32_t)(vh / n) * wpg : 0u;
        const uint32_t w1 = SPLIT ? (w0 + wpg < nwin ? w0 + wpg : nwin) : nwin;
        if (SPLIT && w0 >= w1) continue;
        if (in_phi && !phi) { if (!acc.inf) acc.v.x = mulq(acc.v.x, unpackq(glv_beta())); in_phi = false; }   // the lane crosses into its plain halves: map the phi sum
        uint32_t m[4], sgn;
        if (keep_ok && !phi) {
            const uint32_t at = slot < FB_KEEP ? slot : 0; slot++;
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = kept[at][j];
            sgn = kept[at][4];
            if (slot == (uint32_t)(nvh / L)) slot = 0;
        } else {
            const glv_halves h = glv_split_signed(from_mont<FrP>(sc[i]));
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = phi ? h.k2[j] : h.k1[j];
            sgn = phi ? h.neg2 : h.neg1;
            if (keep_ok) {                                // (phi pass) remember the plain half for this lane's iteration nvh / L from now
                const uint32_t at = slot < FB_KEEP ? slot : 0; slot++;
#pragma unroll
                for (int j = 0; j < 4; j++) kept[at][j] = h.k1[j];
                kept[at][4] = h.neg1;
                if (slot == (uint32_t)(nvh / L)) slot = 0;
            }
        }
        uint32_t raw, carry = 0, mag, ng;
        if (SPLIT) {
#pragma nounroll
            for (uint32_t w = 0; w < w0; w++) carry = (mag_bits(m, w * c, c) + carry > D) ? 1u : 0u;
        }
        raw = mag_bits(m, w0 * c, c) + carry;
        if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
        g1a qn = table[((uint64_t)w0 * table_n + i) * D + (mag ? mag - 1 : 0)];
#pragma nounroll
        for (uint32_t w = w0; w < w1; w++) {
            g1a q = qn;
            const uint32_t cmag = mag, cng = ng;
            if (w + 1 < w1) {
                raw = mag_bits(m, (w + 1) * c, c) + carry;
                if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
                qn = table[((uint64_t)(w + 1) * table_n + i) * D + (mag ? mag - 1 : 0)];
            }
            if (cmag) {
                if (cng ^ sgn) q.y = neg<FpP>(q.y);
                acc.add(q);
                in_phi = phi;
            }
        }
    }
    if (in_phi && !acc.inf) acc.v.x = mulq(acc.v.x, unpackq(glv_beta()));   // a lane that only held phi halves
#ifdef KZG_REDUCE_WAVE_COOP                                   // A/B builds: the four-wavefront form of rounds 2-5
    fb_block_reduce_coop(acc, buf, &lds, tid);
#else
    fb_block_reduce_quad(acc, buf, tid);
#endif
    if (tid == 0) fb_partial_store(partials[blockIdx.x], acc);
}
// one workgroup of four cooperating wavefronts per blob (lane column j = partial sum j): the blob's partial sums are added by a
// tree of wave-cooperative XYZZ additions (4 products deep instead of 13; a lone commitment has 32 partials: 5 levels), then
// column 0 normalises (one inversion: 1 / (ZZ ZZZ)) and converts.  This kernel is pure latency: ~100 us instead of ~230.
__global__ __launch_bounds__(256) void k_fb_finish(const fb_partial *partials, uint32_t blocks_per_blob, uint64_t batch, g1j *out, int to_kilic) {
    __shared__ fb_partial buf[64];
    const uint64_t b = blockIdx.x;
#ifdef KZG_REDUCE_WAVE_COOP                                   // A/B builds: the four-wavefront form of rounds 2-5 (lane column = partial sum, replicas across the wavefronts)
    __shared__ coop_lds lds;
    coop_ctx c; c.L = &lds; c.wave = threadIdx.x >> 6; c.col = threadIdx.x & 63u; c.set = 0;
    const uint32_t col = c.col;
    g1x_acc acc; acc.init();
    acc.v = g1xq_from_affine(g1a_inf());                   // defined limbs while empty
    const uint32_t rounds = (blocks_per_blob + 63) / 64;
#pragma nounroll
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t j = r * 64 + col;
        const bool have = j < blocks_per_blob;
        const fb_partial &pj = partials[b * blocks_per_blob + (have ? j : 0)];
        g1xq w; fb_partial_load(pj, w);
        coop_acc_add(acc, w, !have || pj.inf != 0, c);
    }
    const uint32_t live = blocks_per_blob < 64 ? blocks_per_blob : 64;
    uint32_t off = 1;
    while (off < live) off *= 2;                          // smallest power of two >= live
#pragma nounroll
    for (off >>= 1; off >= 1; off >>= 1) {
        const bool have = col < off && col + off < live;
        g1xq w; uint32_t wi;                               // column col + off lives in lane col + off of the same wave (replicas)
        point_down_any(w, wi, acc.v, acc.inf ? 1u : 0u, off);
        coop_acc_add(acc, w, !have || wi != 0, c);
    }
    const bool first_wave = c.wave == 0;
#else
    // quad column c = lanes 4 c .. 4 c + 3 (replicas) takes the partial sums c, c + 64, ...; then the quad tree; quad 0 ends with the blob's sum (k_msm.hip, above)
    const uint32_t tid = threadIdx.x, role = tid & 3u, quad = tid >> 2, col = tid & 63u;
#ifdef KZG_FINISH_TIMING
    const uint64_t tf0 = wall_clock64();
#endif
    g1x_acc acc; acc.init();
    acc.v = g1xq_from_affine(g1a_inf());                   // defined limbs while empty
    const uint32_t rounds = (blocks_per_blob + 63) / 64;
    // (the next round's partial sum is loaded before this round's addition: its HBM latency hides behind the addition)
    g1xq wn; bool wn_inf;
    { const bool have = quad < blocks_per_blob; const fb_partial &pj = partials[b * blocks_per_blob + (have ? quad : 0)]; fb_partial_load(pj, wn); wn_inf = !have || pj.inf != 0; }
#pragma nounroll
    for (uint32_t r = 0; r < rounds; r++) {
        const g1xq w = wn; const bool winf = wn_inf;
        if (r + 1 < rounds) {
            const uint32_t j = (r + 1) * 64 + quad;
            const bool have = j < blocks_per_blob;
            const fb_partial &pj = partials[b * blocks_per_blob + (have ? j : 0)];
            fb_partial_load(pj, wn); wn_inf = !have || pj.inf != 0;
        }
        quad_acc_add_nl(acc, w, winf, role);
    }
#ifdef KZG_FINISH_TIMING
    __syncthreads(); const uint64_t tf1 = wall_clock64();
#endif
    quad_tree_reduce(acc, buf, tid, blocks_per_blob < 64 ? blocks_per_blob : 64);
#ifdef KZG_FINISH_TIMING
    __syncthreads(); const uint64_t tf2 = wall_clock64();
#endif
    const bool first_wave = tid < 64;
#endif
    if (first_wave) {                                       // wave-uniform: the whole first wavefront takes part in the inversion, its lane 0 owns the result
        const bool own = col == 0;
        const g1x px = g1xq_pack(acc.v);                    // (defined limbs in every column: empty accumulators hold the affine image of infinity)
        // x = X / ZZ, y = Y / ZZZ with ONE inversion: i = 1 / (ZZ ZZZ), 1 / ZZ = i ZZZ, 1 / ZZZ = i ZZ -- spread over the wavefront's lanes (coop_inv.hpp)
        const bool need = own && !acc.inf && !(to_kilic & 2);
#ifdef KZG_FINISH_NOINV                                      // timing experiment only (wrong results): what the inversion costs
        const fp i = mul(px.zz, px.zzz);
#elif defined(KZG_NO_COOP_INV)
        fp i = zero<FpP>();
        if (need) i = inv<FpP>(mul(px.zz, px.zzz));
#else
        const fp i = wave_inv_any(mul(px.zz, px.zzz), need);
#endif
#if !defined(KZG_REDUCE_WAVE_COOP) && !defined(KZG_NO_COOP_INV) && !defined(KZG_FINISH_NOINV)
        // the affine image: x = X i ZZZ on lane 0, y = Y i ZZ on lane 1 (quad 0 holds replicas of the sum; the inverse is the owner's: handed to lane 1 by a DPP move), each
        // followed by its own conversion to Kilic's image -- three dependent products per lane instead of six on one
        const uint32_t aff = __builtin_amdgcn_readfirstlane((uint32_t)(!acc.inf && !(to_kilic & 2)));
        if (aff) {
            fp i1;
#pragma unroll
            for (int k = 0; k < 12; k++) i1.l[k] = (uint32_t)__builtin_amdgcn_readlane((int)i.l[k], 0);
            if (tid < 2) {
                fp c = mul(tid ? px.y : px.x, mul(i1, tid ? px.zz : px.zzz));
                if (to_kilic & 1) c = fp_to_kilic(c);
                fp *dst = tid ? &out[b].y : &out[b].x;
                *dst = c;
                if (tid == 0) out[b].z = (to_kilic & 1) ? fp_kilic_one() : one<FpP>();
            }
        } else if (own) {
            g1j r;
            if (acc.inf) r = g1_inf(); else r = g1x_to_jac(px);  // infinity, or the projective output (kzg_hip_kzg_set_projective_outputs): (X ZZ, Y ZZZ, ZZ), no inversion
            out[b] = (to_kilic & 1) ? g1_to_kilic(r) : r;
        }
#else
        if (own) {
            g1j r;
            if (acc.inf) r = g1_inf();
            else if (to_kilic & 2) r = g1x_to_jac(px);       // projective output (kzg_hip_kzg_set_projective_outputs): (X ZZ, Y ZZZ, ZZ), no inversion
            else { r.x = mul(px.x, mul(i, px.zzz)); r.y = mul(px.y, mul(i, px.zz)); r.z = one<FpP>(); }
            out[b] = (to_kilic & 1) ? g1_to_kilic(r) : r;
        }
#endif
    }
#if defined(KZG_FINISH_TIMING) && !defined(KZG_REDUCE_WAVE_COOP)
    if (threadIdx.x == 0 && b == 0) printf("finish phases (us): %u rounds of load + add %.1f | tree %.1f | inversion + output %.1f\n", (blocks_per_blob + 63) / 64, (tf1 - tf0) * 0.01, (tf2 - tf1) * 0.01, (wall_clock64() - tf2) * 0.01);
#endif
}

// The same for LARGE batches (at most 4 partial sums per blob: 128 blobs and more): one LANE per blob adds its few partials and
// normalises.  The cooperative kernel above spends a 256-thread workgroup -- and with it a quarter of a CU's registers -- on every blob
// although there is nothing left to reduce, so 512 blobs ran as two rounds of 110 us inversions; 512 lanes are 8 wavefronts and one
// round (measured per 512-blob step: 6.02 -> 5.7x ms).
__global__ __launch_bounds__(64) void k_fb_finish_lanes(const fb_partial *partials, uint32_t blocks_per_blob, uint64_t batch, g1j *out, int to_kilic) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= batch) return;
    g1x_acc acc; acc.init();
#pragma nounroll
    for (uint32_t j = 0; j < blocks_per_blob; j++) {
        const fb_partial &pj = partials[b * blocks_per_blob + j];
        g1xq w; fb_partial_load(pj, w);
        g1x_acc_merge(acc, w, pj.inf != 0);
    }
    g1j r;
    if (acc.inf) r = g1_inf();
    else if (to_kilic & 2) r = g1x_to_jac(g1xq_pack(acc.v));       // projective output: no inversion
    else {
        g1x px = g1xq_pack(acc.v);
        fp i = inv<FpP>(mul(px.zz, px.zzz));
        r.x = mul(px.x, mul(i, px.zzz)); r.y = mul(px.y, mul(i, px.zz)); r.z = one<FpP>();
    }
    out[b] = (to_kilic & 1) ? g1_to_kilic(r) : r;
}

// One fixed-base product added into an accumulator: the walk of point i's rows with the signed c-bit digits of a scalar (standard form), sign sg folded in.
//   GLV = false: the plain layout, nwin windows over the whole scalar (phi is ignored, one call per term);
//   GLV = true : the table holds ceil(128 / c) windows; the call walks ONE half of the split scalar -- phi ? |k2| : |k1| -- so a caller sums the phi halves of all
//                its terms first, maps the sum once (acc_apply_phi: X <- beta X), then adds the plain halves (the order of k_fb_accumulate_glv).
template <bool GLV> __device__ __forceinline__ void fb_walk_term(g1x_acc &acc, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, uint64_t i,
                                                                const fr &k, uint32_t sg, bool phi) {
    uint32_t m[8];
    uint32_t sgn = sg;
    if (GLV) {
        const glv_halves h = glv_split_signed(k);
#pragma unroll
        for (int j = 0; j < 4; j++) { m[j] = phi ? h.k2[j] : h.k1[j]; m[4 + j] = 0; }
        sgn ^= phi ? h.neg2 : h.neg1;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = k.l[j];
    }
    auto bits = [&](uint32_t off) -> uint32_t {
        const uint32_t idx = off >> 5, sh = off & 31;
        if (idx >= (GLV ? 4u : 8u)) return 0u;
        uint64_t v = m[idx];
        if (idx + 1 < (GLV ? 4u : 8u)) v |= (uint64_t)m[idx + 1] << 32;
        return (uint32_t)(v >> sh) & ((1u << c) - 1u);
    };
    // software pipeline as in k_fb_accumulate: the gather of window w + 1 is issued before the addition of window w
    uint32_t raw = bits(0), carry, mag, ng;
    if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
    g1a qn = table[((uint64_t)0 * table_n + i) * D + (mag ? mag - 1 : 0)];
#pragma nounroll
    for (uint32_t w = 0; w < nwin; w++) {
        g1a q = qn;
        const uint32_t cmag = mag, cng = ng ^ sgn;
        if (w + 1 < nwin) {
            raw = bits((w + 1) * c) + carry;
            if (raw > D) { carry = 1; mag = (1u << c) - raw; ng = 1; } else { carry = 0; mag = raw; ng = 0; }
            qn = table[((uint64_t)(w + 1) * table_n + i) * D + (mag ? mag - 1 : 0)];
        }
        if (cmag) {
            if (cng) q.y = neg<FpP>(q.y);
            acc.add(q);
        }
    }
}
// element-wise fixed-base products over the same table layout: out[b][i] = scalars[b][i] * P_i  (the FK20 Toeplitz stage,
// ToeplitzPart2's loop fk20_single.go:72-74, where P = xExtFFT is fixed per settings): nwin mixed adds instead of a
// 255-bit double-and-add per element.
template <bool GLV> __global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_mul_vec(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                         uint64_t i0, uint64_t cnt, uint64_t row, uint64_t total, g1j *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    // t enumerates (b, f, jj): point index i = f * k2 + j0 + jj is supplied through (row = k2, i0 = j0, cnt) by the caller
    uint64_t jj = t % cnt, f = (t / cnt) % (table_n / row), b = t / (cnt * (table_n / row));
    uint64_t i = f * row + i0 + jj;
    const fr k = from_mont<FrP>(scalars[b * table_n + i]);
    g1x_acc acc; acc.init();
    if (GLV) { fb_walk_term<true>(acc, table, table_n, c, nwin, D, i, k, 0u, true); acc_apply_phi(acc); }
    fb_walk_term<GLV>(acc, table, table_n, c, nwin, D, i, k, 0u, false);
    out[t] = acc.to_jac();
}
// The FK20 Toeplitz stage fused with the FIRST direct pass of the inverse G1 transform of a LONE polynomial (fk20_single.go:72-74 + the first
// radix-R pass of k_g1_fft_direct over v_i = C[i] X[i]):  out[j R + u] = sum_{t < R} w^(t cols u) C[i] X[i],  i = j + t cols, cols = N / R.
// Every term is a fixed-base product: the scalar C[i] w^e (one F_r product) walks X[i]'s resident table -- nwin mixed additions instead of the
// 128 doublings + 66 additions of a variable-base multiplication -- then the R terms of an output are summed as in the direct pass.  One lane per
// (output, term); one-wavefront workgroups.  roots = ReverseRootsOfUnity (Montgomery), W = its width.
struct fbp_slot { uint32_t w[39]; uint32_t inf; uint32_t pad; };
template <bool GLV> __global__ __launch_bounds__(64, 2) void k_fb_direct_pass1(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars, const fr *roots,
                                                           uint64_t W, uint32_t logn, uint32_t logR, uint64_t total, g1j *out) {
    __shared__ fbp_slot buf[64];
    const uint32_t tid = threadIdx.x;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + tid;
    const uint64_t n = 1ull << logn, R = 1ull << logR, cols = n >> logR;
    const uint32_t tt = (uint32_t)(t & (R - 1));
    const uint64_t u = (t >> logR) & (R - 1), jb = t >> (2 * logR), j = jb % cols, b = jb / cols;
    const bool live = t < total;
    g1jq_acc acc; acc.inf = true;
    if (live) {
        const uint64_t i = j + (uint64_t)tt * cols, e = ((uint64_t)tt * cols * u) & (n - 1);
        fr kmont = scalars[b * n + i];
        if (e) kmont = mul(kmont, roots[e * (W >> logn)]);
        const fr k = from_mont<FrP>(kmont);
        g1x_acc xa; xa.init();
        if (GLV) { fb_walk_term<true>(xa, table, table_n, c, nwin, D, i, k, 0u, true); acc_apply_phi(xa); }
        fb_walk_term<GLV>(xa, table, table_n, c, nwin, D, i, k, 0u, false);
        if (!xa.inf) { acc.v = g1jq_unpack(xa.to_jac()); acc.inf = false; }
    }
#define FBP_STORE(i_) do { _Pragma("unroll") for (int q_ = 0; q_ < 13; q_++) { buf[i_].w[q_] = acc.v.x.l[q_]; buf[i_].w[13 + q_] = acc.v.y.l[q_]; buf[i_].w[26 + q_] = acc.v.z.l[q_]; } \
                           buf[i_].inf = acc.inf ? 1u : 0u; } while (0)
    FBP_STORE(tid);
    __syncthreads();
#pragma nounroll
    for (uint32_t off = (uint32_t)R / 2; off >= 1; off >>= 1) {
        if (tt < off && !buf[tid + off].inf) {
            g1jq q;
#pragma unroll
            for (int i = 0; i < 13; i++) { q.x.l[i] = buf[tid + off].w[i]; q.y.l[i] = buf[tid + off].w[13 + i]; q.z.l[i] = buf[tid + off].w[26 + i]; }
            acc.add(q);
            FBP_STORE(tid);
        }
        __syncthreads();
    }
#undef FBP_STORE
    if (live && tt == 0) out[b * n + j * R + u] = acc.inf ? g1_inf() : g1jq_pack(acc.v);
}
void launch_fb_direct_pass1(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, const fr *roots, uint64_t W, uint64_t batch,
                            uint32_t logR, g1j *out, bool glv) {
    uint32_t logn = 0;
    while ((1ull << logn) < table_n) logn++;
    const uint64_t total = (batch * table_n) << logR, wgs = (total + 63) / 64;
    prof_begin(s, "fb_mul_vec");
    if (glv) hipLaunchKernelGGL(k_fb_direct_pass1<true>, dim3((uint32_t)wgs), dim3(64), wgs <= device_simds() ? 24 * 1024 : 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, roots, W, logn, logR,
                                total, out);
    else hipLaunchKernelGGL(k_fb_direct_pass1<false>, dim3((uint32_t)wgs), dim3(64), wgs <= device_simds() ? 24 * 1024 : 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, roots, W, logn, logR,
                            total, out);
    prof_end(s, "fb_mul_vec");
}

// The FK20 Toeplitz stage fused with the FIRST TWO decimation-in-frequency stages of the inverse G1 transform that follows it
// (fk20_single.go:72-74 + the first two levels of ToeplitzPart3's FFTG1, :80-87).  With v_j = C[j] X[j], q = N / 4, w = the inverse
// root of order N and i < q, two DIF stages give
//     y[i]         =            v_i +           v_{i+q} +          v_{i+2q} +           v_{i+3q}
//     y[q + i]     = w^{2i}   ( v_i -           v_{i+q} +          v_{i+2q} -           v_{i+3q} )
//     y[2q + i]    = w^{i}      v_i + w^{i+q}   v_{i+q} - w^{i}    v_{i+2q} - w^{i+q}   v_{i+3q}
//     y[3q + i]    = w^{3i}     v_i - w^{3i+q}  v_{i+q} - w^{3i}   v_{i+2q} + w^{3i+q}  v_{i+3q}
// i.e. every output is a FOUR-term fixed-base MSM over the resident X table with scalars C[.] * (a root of unity): 4 x nwin mixed
// additions (~720 product-equivalents) replace one table walk + two butterfly levels (~1640), the twiddle multiplications of the two
// widest stages of the transform (a fifth of all of them) disappear.  Lane = (polynomial, output); scalars in Montgomery form;
// roots = ReverseRootsOfUnity (Montgomery), W = its width.
template <bool GLV> __global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_mul_vec_dif2(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                              const fr *roots, uint64_t W, uint64_t total, g1j *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t N = table_n, q = N >> 2;
    const uint64_t i = t % q, o = (t / q) & 3u, b = t / N;
    const uint64_t rs = W / N;                              // stride of the order-N roots inside the width-W table
    g1x_acc acc; acc.init();
    // GLV: the phi halves of the four terms first, the map once, then the plain halves (the scalar of a term is formed in both passes: one F_r product)
#pragma nounroll
    for (uint32_t pass = GLV ? 0u : 1u; pass < 2; pass++) {
#pragma nounroll
        for (uint32_t term = 0; term < 4; term++) {
            const uint64_t pi = i + term * q;
            // exponent and sign of the root that multiplies C[pi] in output o (table above)
            uint64_t e; uint32_t sg;
            if (o == 0) { e = 0; sg = 0; }
            else if (o == 1) { e = 2 * i; sg = term & 1u; }
            else if (o == 2) { e = i + ((term & 1u) ? q : 0); sg = term >> 1; }
            else { e = 3 * i + ((term & 1u) ? q : 0); sg = (term == 1 || term == 2) ? 1u : 0u; }
            fr kmont = scalars[b * N + pi];
            if (e) kmont = mul(kmont, roots[(e & (N - 1)) * rs]);
            const fr k = from_mont<FrP>(kmont);
            fb_walk_term<GLV>(acc, table, table_n, c, nwin, D, pi, k, sg, pass == 0);   // the table walk of k_fb_mul_vec for point pi, sign folded into the digits
        }
        if (GLV && pass == 0) acc_apply_phi(acc);
    }
    out[b * N + o * q + i] = acc.to_jac();
}
void launch_fb_mul_vec_dif2(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, const fr *roots, uint64_t W,
                            uint64_t batch, g1j *out, bool glv) {
    uint64_t total = batch * table_n;
    if (!total) return;
    prof_begin(s, "fb_mul_vec");
    if (glv) hipLaunchKernelGGL(k_fb_mul_vec_dif2<true>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                                roots, W, total, out);
    else hipLaunchKernelGGL(k_fb_mul_vec_dif2<false>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                            roots, W, total, out);
    prof_end(s, "fb_mul_vec");
}
// FK20Multi's Toeplitz stage (fk20_multi.go:79-91): out[b][jj] = sum over the files f of scalars[b][f * row + j0 + jj] * X_f[j0 + jj],
// all files of an output position in ONE lane's accumulator (nfiles x nwin mixed additions): no per-file temporaries and no
// summation pass with generic additions afterwards.
template <bool GLV> __global__ __launch_bounds__(FB_BLOCK, 2) void k_fb_mul_vec_files(const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, uint32_t D, const fr *scalars,
                                                               uint64_t i0, uint64_t cnt, uint64_t row, uint64_t total, g1j *out) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t jj = t % cnt, b = t / cnt, nfiles = table_n / row;
    g1x_acc acc; acc.init();
#pragma nounroll
    for (uint32_t pass = GLV ? 0u : 1u; pass < 2; pass++) {       // GLV: phi halves of every file, the map once, then the plain halves
#pragma nounroll
        for (uint64_t f = 0; f < nfiles; f++) {
            const uint64_t i = f * row + i0 + jj;
            const fr k = from_mont<FrP>(scalars[b * table_n + i]);
            fb_walk_term<GLV>(acc, table, table_n, c, nwin, D, i, k, 0u, pass == 0);
        }
        if (GLV && pass == 0) acc_apply_phi(acc);
    }
    out[t] = acc.to_jac();
}
void launch_fb_mul_vec_files(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t row, uint64_t j0,
                             uint64_t cnt, uint64_t batch, g1j *out, bool glv) {
    uint64_t total = batch * cnt;
    if (!total) return;
    prof_begin(s, "fb_mul_vec");
    if (glv) hipLaunchKernelGGL(k_fb_mul_vec_files<true>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                                j0, cnt, row, total, out);
    else hipLaunchKernelGGL(k_fb_mul_vec_files<false>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                            j0, cnt, row, total, out);
    prof_end(s, "fb_mul_vec");
}
void launch_fb_mul_vec(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t row, uint64_t j0,
                       uint64_t cnt, uint64_t batch, g1j *out, bool glv) {
    uint64_t total = batch * (table_n / row) * cnt;
    if (!total) return;
    prof_begin(s, "fb_mul_vec");
    if (glv) hipLaunchKernelGGL(k_fb_mul_vec<true>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                                j0, cnt, row, total, out);
    else hipLaunchKernelGGL(k_fb_mul_vec<false>, dim3((uint32_t)((total + FB_BLOCK - 1) / FB_BLOCK)), dim3(FB_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars,
                            j0, cnt, row, total, out);
    prof_end(s, "fb_mul_vec");
}

uint64_t device_simd_lanes() {
    static std::atomic<uint64_t> lanes_of_device[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<uint64_t> &slot = lanes_of_device[dev >= 0 && dev < 64 ? dev : 0];
    uint64_t lanes = slot.load(std::memory_order_relaxed);
    if (!lanes) {
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
        lanes = (uint64_t)(cus > 0 ? cus : 256) * 4 * 64;
        slot.store(lanes, std::memory_order_relaxed);        // (a racing thread computes the same value)
    }
    return lanes;
}
// n = virtual points of a blob (the points themselves, or 2 x points for the GLV walk: a phi half and a plain half each)
static uint32_t fb_blocks_per_blob(uint64_t n, uint64_t batch, uint32_t *wsplit = nullptr) {
    // 131072 lanes = 2048 wavefronts = exactly the resident capacity at 2 waves per SIMD: ONE round.  Measured (512 blobs):
    // 131072 lanes 5.7 ms, 262144 lanes (two rounds) 6.4 ms, 98304 / 65536 lanes 10.4 ms.  KZG_HIP_FB_LANES overrides.
    // (thread-safe: coalescer leaders on several host threads get here at once; one value per device)
    static const uint64_t forced_lanes = [] { const char *e = getenv("KZG_HIP_FB_LANES"); return e ? strtoull(e, nullptr, 10) : 0ull; }();
    uint64_t lanes = forced_lanes ? forced_lanes : 2 * device_simd_lanes();   // CUs x 4 SIMDs x 2 resident waves x 64 lanes (256 CUs: 131072)
    if (lanes < FB_ACC_BLOCK) lanes = FB_ACC_BLOCK;
    uint64_t target = lanes / FB_ACC_BLOCK;
    uint64_t bpb = target / (batch ? batch : 1);
    // small batches: up to 8 lanes per point (each takes a share of the windows) while the launch stays within one round of lanes
    uint32_t split = 1;
    while (split < 8 && (batch ? batch : 1) * n * split * 2 <= lanes) split *= 2;
    if (wsplit) *wsplit = split;
    uint64_t maxb = (n * split + FB_ACC_BLOCK - 1) / FB_ACC_BLOCK;
    if (bpb > maxb) bpb = maxb;
    if (bpb < 1) bpb = 1;
    return (uint32_t)bpb;
}
// (the GLV walk has 2 n virtual points and half the windows per virtual point: its lanes per blob, and so its partial sums, equal the plain walk's)
size_t fb_partials_bytes(uint64_t n, uint64_t batch) { return (((size_t)std::max(fb_blocks_per_blob(n, batch), fb_blocks_per_blob(2 * n, batch)) * batch * sizeof(fb_partial)) + 15) / 16 * 16; }

// out[b] = the NORMALISED sum (Z = one), as Kilic images when to_kilic: the per-blob partial sums are added and inverted in one
// latency-bound kernel instead of two.  glv: the table holds ceil(128 / c) windows and both GLV halves of every scalar walk them (k_fb_accumulate_glv).
void launch_fb_msm(hipStream_t s, const g1a *table, uint64_t table_n, uint32_t c, uint32_t nwin, const fr *scalars, uint64_t sc_stride, uint64_t n,
                   uint64_t batch, void *partials, g1j *out, bool to_kilic, bool glv, bool projective) {
    if (!batch) return;
    const int fin = (to_kilic ? 1 : 0) | (projective ? 2 : 0);
    uint32_t split = 1;
    uint32_t bpb = fb_blocks_per_blob(glv ? 2 * n : n, batch, &split);   // (glv: `split` lanes per HALF, so up to 16 lanes per point with one window each)
    if (split > nwin) split = nwin;
    prof_begin(s, "fb_accumulate");
    if (glv) {
        if (split > 1) hipLaunchKernelGGL(k_fb_accumulate_glv<true>, dim3((uint32_t)(batch * bpb)), dim3(FB_ACC_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, sc_stride, n,
                                          bpb, split, (fb_partial *)partials);
        else hipLaunchKernelGGL(k_fb_accumulate_glv<false>, dim3((uint32_t)(batch * bpb)), dim3(FB_ACC_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, sc_stride, n, bpb,
                                1u, (fb_partial *)partials);
    } else if (split > 1) hipLaunchKernelGGL(k_fb_accumulate<true>, dim3((uint32_t)(batch * bpb)), dim3(FB_ACC_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, sc_stride, n,
                                      bpb, split, (fb_partial *)partials);
    else hipLaunchKernelGGL(k_fb_accumulate<false>, dim3((uint32_t)(batch * bpb)), dim3(FB_ACC_BLOCK), 0, s, table, table_n, c, nwin, 1u << (c - 1), scalars, sc_stride, n, bpb,
                            1u, (fb_partial *)partials);
    prof_end(s, "fb_accumulate");
    if (bpb <= FB_FINISH_LANES_MAX) hipLaunchKernelGGL(k_fb_finish_lanes, dim3((uint32_t)((batch + 63) / 64)), dim3(64), 0, s, (const fb_partial *)partials, bpb, batch, out, fin);
    else hipLaunchKernelGGL(k_fb_finish, dim3((uint32_t)batch), dim3(256), 0, s, (const fb_partial *)partials, bpb, batch, out, fin);
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
