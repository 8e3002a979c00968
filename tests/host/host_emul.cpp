// Host build of the device arithmetic headers (go-kzg_amd/csrc/field.hpp, g1.hpp) so that the exact
// source the HIP kernels inline is checked against the oracle on a machine without a GPU.
// TEST INFRASTRUCTURE: built by tests/test_host_arith.py into tests/host/_build/, never shipped.
#include "field.hpp"
#include "g1.hpp"
#include "fr_fft4096.hpp"
#include "../../tools/ab_fr_r16/fr16.hpp"   // the 256-lane form: an A/B artefact since round 6, its emulation stays tested
#include "fr_das2048.hpp"
#include "coop_inv.hpp"
#include <vector>
#include <string.h>
using namespace kzg;
// n pseudo-random (and bit-pattern-structured) elements: inv(x) * x == one and inv(x) == inv_fermat(x) on every 16th; returns mismatches
template <class F> static uint64_t inv_stress(uint64_t n, uint64_t seed) {
    uint64_t bad = 0, st = seed;
    auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    for (uint64_t i = 0; i < n; i++) {
        felem<F> x;
        uint64_t mode = next() % 4;
        for (int w = 0; w < F::N; w++) {
            uint64_t r = next();
            x.l[w] = mode == 0 ? (uint32_t)r : mode == 1 ? (uint32_t)(r & (r >> 32)) & (uint32_t)next() : mode == 2 ? ((r & 7) ? 0u : (uint32_t)(r >> 8)) : ~((uint32_t)(r & (r >> 32)) & (uint32_t)next());
        }
        x.l[F::N - 1] &= (1u << ((F::BITS - 1) % 32)) - 1;           // below the modulus
        if (next() % 8 == 0) for (int w = (int)(next() % F::N); w < F::N; w++) x.l[w] = 0;   // short values
        felem<F> y = inv<F>(x);
        if (is_zero<F>(x)) { bad += !is_zero<F>(y); continue; }
        if (!equal<F>(mul(x, y), one<F>())) bad++;
        if (i % 16 == 0 && !equal<F>(y, inv_fermat<F>(x))) bad++;
    }
    return bad;
}

// k_fr_fft_small<LOGM>: one workgroup's 4096 / m transforms of m = 2^logm points through the first passes of the 4096-point network (+ one radix-2
// pass for odd logm), lane by lane.  `batch` transforms exist (rows beyond it are zeros and are not stored).  Returns the largest raw limb seen in LDS.
template <int LOGM> static uint32_t fr_fft_small_emul(const fr *in, uint64_t in_stride, uint64_t n_in, uint64_t batch, fr *out, const uint32_t *tw, const fr *scale) {
    std::vector<uint32_t> lds(9 * fr4::NPAD, 0);
    uint32_t worst = 0;
    auto scan = [&]() { for (uint32_t v : lds) if (v > worst) worst = v; };
    constexpr int A = LOGM / 2;
    constexpr uint32_t m = 1u << LOGM, per = fr4::N / m;
    uint32_t *s = lds.data();
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_first_small<LOGM>(t, in, in_stride, n_in, 0, batch, s, tw);
    scan();
    if (A >= 2) { for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<4>(t >> 6, t & 63, s, tw); scan(); }
    if (A >= 3) { for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<16>(t >> 6, t & 63, s, tw); scan(); }
    if (A >= 4) { for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<64>(t, s, tw); scan(); }
    if (A >= 5) { for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<256>(t, s, tw); scan(); }
    if (LOGM & 1) { for (uint32_t t = 0; t < 1024; t++) fr4::pass_r2<(1u << (2 * A))>(t, t >> 6, t & 63, s, tw); scan(); }
    frl sc = frl_zero();
    if (scale) sc = frl_const_from_kilic(*scale);
    const uint64_t limit = batch >= per ? fr4::N : batch * m;
    for (uint32_t t = 0; t < 1024; t++) { if (scale) fr4::pass_store<true>(t, s, sc, out, limit); else fr4::pass_store<false>(t, s, sc, out, limit); }
    return worst;
}
// A transform of R * 4096 points as the device runs it: the R rows (every R-th element from offset bitrev(a)) through the 4096-point passes with the
// element stride of k_fr_fft4096_r4's rows_log form, then fr4::upper_lane for every k2.  roots: W + 1 Kilic images; scale: null or the image of 1 / n.
template <int LOGR> static void fr_fft_long_emul(const fr *in, uint64_t n_in, fr *out, const fr *roots, uint64_t W, const fr *scale) {
    constexpr uint32_t R = 1u << LOGR;
    std::vector<uint32_t> tw(fr4::TW_WORDS), lds(9 * fr4::NPAD);
    fr4::build_twiddles(roots, W, tw.data());
    std::vector<fr> roots_l(W + 1);
    const fr k32 = fr_from_u64(32);
    for (uint64_t i = 0; i <= W; i++) roots_l[i] = mul(roots[i], k32);
    frl sc0 = frl_zero();
    for (uint32_t a = 0; a < R; a++) {
        uint32_t off = 0;
        for (int k = 0; k < LOGR; k++) off |= ((a >> k) & 1u) << (LOGR - 1 - k);
        uint32_t *s = lds.data();
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_first(t, in, n_in, s, tw.data(), R, off);
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<4>(t >> 6, t & 63, s, tw.data());
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<16>(t >> 6, t & 63, s, tw.data());
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<64>(t, s, tw.data());
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<256>(t, s, tw.data());
        for (uint32_t t = 0; t < 1024; t++) fr4::pass_last<false>(t, s, tw.data(), sc0, out + (uint64_t)a * fr4::N);
    }
    for (uint32_t k2 = 0; k2 < fr4::N; k2++) { if (scale) fr4::upper_lane<LOGR, true>(out, k2, roots_l.data(), W, scale); else fr4::upper_lane<LOGR, false>(out, k2, roots_l.data(), W, scale); }
}
// wave_inv_fp (coop_inv.hpp) lane by lane: 16 lanes hold one limb each of f, g, d, e; the hand-overs of carry_div30 / carry_keep are array shifts.  Same scalar
// pieces as the device code (divsteps30_var, split64 / split32, de_multipliers, final_quotient, canonical_from_centred).  Also checks the invariants the device form
// relies on: centred limbs stay within 2^29 + 2, products within 2^61, hand-overs of the top lane are zero.  *rounds / *iters: rounds run, inner iterations.
template <class F> static felem<F> coop_inv_emul(const felem<F> &x, uint32_t *rounds, uint32_t *bad) {
    using namespace cinv;
    constexpr int L = F::N30;
    if (is_zero<F>(x)) return zero<F>();
    int32_t f[16] = {0}, g[16] = {0}, d[16] = {0}, e[16] = {0}, pj[16] = {0};
    for (int k = 0; k < L; k++) { g[k] = (int32_t)limb30(x.l, F::N, k); pj[k] = (int32_t)F::p30(k); f[k] = pj[k]; e[k] = (int32_t)r2_limb30<F>(k); }
    auto lim = [&](int64_t t) { if (t >= (1ll << 61) || t <= -(1ll << 61)) (*bad)++; };
    auto div30 = [&](const int64_t *t, int32_t *out) {
        int32_t lo[17] = {0}, hi[16], v[16], lo2[16], hi2[16];
        for (int j = 0; j < 16; j++) { lim(t[j]); split64(t[j], lo[j], hi[j]); }
        if (lo[0] != 0) (*bad)++;                                              // the low limb of a round's combination is zero by construction
        for (int j = 0; j < 16; j++) { v[j] = lo[j + 1] + hi[j]; split32(v[j], lo2[j], hi2[j]); }
        for (int j = 0; j < 16; j++) out[j] = lo2[j] + (j ? hi2[j - 1] : 0);
        if (hi2[15] != 0 || hi2[L - 1] != 0) (*bad)++;
        for (int j = 0; j < 16; j++) if (out[j] > (1 << 29) + 2 || out[j] < -(1 << 29) - 2 || (j >= L && out[j] != 0)) (*bad)++;
    };
    int32_t eta = -1;
    uint32_t n = 0;
    for (;;) {
        bool nz = false;
        for (int j = 0; j < 16; j++) nz |= g[j] != 0;
        if (!nz) break;
        if (++n > 64) { (*bad)++; break; }
        int32_t u, v, q, r, md, me;
        divsteps30_var(eta, (uint32_t)f[0], (uint32_t)g[0], u, v, q, r);
        de_multipliers<F>(u, v, q, r, d[0], e[0], md, me);
        int64_t tf[16], tg[16], td[16], te[16];
        for (int j = 0; j < 16; j++) {
            tf[j] = (int64_t)u * f[j] + (int64_t)v * g[j]; tg[j] = (int64_t)q * f[j] + (int64_t)r * g[j];
            td[j] = ((int64_t)u * d[j] + (int64_t)v * e[j]) + (int64_t)md * pj[j]; te[j] = ((int64_t)q * d[j] + (int64_t)r * e[j]) + (int64_t)me * pj[j];
        }
        div30(tf, f); div30(tg, g); div30(td, d); div30(te, e);
    }
    if (rounds) *rounds = n;
    if (!((f[0] == 1 || f[0] == -1))) (*bad)++;
    for (int j = 1; j < 16; j++) if (f[j]) (*bad)++;
    if (f[0] < 0) for (int j = 0; j < 16; j++) d[j] = -d[j];
    const int32_t qe = final_quotient<F>(d[L - 1], d[L - 2]);
    {
        int32_t lo[16], hi[16], v[16], lo2[16], hi2[16];
        for (int j = 0; j < 16; j++) split64((int64_t)d[j] - (int64_t)qe * pj[j], lo[j], hi[j]);
        for (int j = 0; j < 16; j++) { v[j] = lo[j] + (j ? hi[j - 1] : 0); split32(v[j], lo2[j], hi2[j]); }
        for (int j = 0; j < 16; j++) d[j] = lo2[j] + (j ? hi2[j - 1] : 0);
        if (hi[L - 1] != 0 || hi2[L - 1] != 0) (*bad)++;
    }
    return canonical_from_centred<F>(d);
}
// n structured / random elements: the cooperative form == inv<F>() word for word; returns mismatches + violated invariants; *max_rounds, *sum_rounds
template <class F> static uint64_t coop_stress(uint64_t n, uint64_t seed, uint32_t *max_rounds, uint64_t *sum_rounds) {
    uint64_t bad = 0, st = seed;
    auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    *max_rounds = 0; *sum_rounds = 0;
    for (uint64_t i = 0; i < n; i++) {
        felem<F> x;
        uint64_t mode = next() % 4;
        for (int w = 0; w < F::N; w++) {
            uint64_t r = next();
            x.l[w] = mode == 0 ? (uint32_t)r : mode == 1 ? (uint32_t)(r & (r >> 32)) & (uint32_t)next() : mode == 2 ? ((r & 7) ? 0u : (uint32_t)(r >> 8)) : ~((uint32_t)(r & (r >> 32)) & (uint32_t)next());
        }
        x.l[F::N - 1] &= (1u << ((F::BITS - 1) % 32)) - 1;                    // below the modulus
        if (next() % 8 == 0) for (int w = (int)(next() % F::N); w < F::N; w++) x.l[w] = 0;   // short values
        uint32_t rounds = 0, b = 0;
        felem<F> y = coop_inv_emul<F>(x, &rounds, &b);
        bad += b;
        if (!equal<F>(y, inv<F>(x))) bad++;
        if (rounds > *max_rounds) *max_rounds = rounds;
        *sum_rounds += rounds;
    }
    return bad;
}
extern "C" {
// the radix-4 4096-point transform of k_fr_fft4096_r4, lane by lane and pass by pass (a barrier between passes == finishing the loop over
// the lanes): roots = W + 1 Kilic images (expanded or reversed), scale = null or the Kilic image of 1/n.  Also reports the largest raw limb
// seen in LDS (the header promises < 6 * 2^29).
uint32_t he_fr_fft4096(const fr *in, uint64_t n_in, fr *out, const fr *roots, uint64_t W, const fr *scale) {
    std::vector<uint32_t> tw(fr4::TW_WORDS), lds(9 * fr4::NPAD, 0);
    fr4::build_twiddles(roots, W, tw.data());
    uint32_t worst = 0;
    auto scan = [&]() { for (uint32_t v : lds) if (v > worst) worst = v; };
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_first(t, in, n_in, lds.data(), tw.data());
    scan();
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<4>(t >> 6, t & 63, lds.data(), tw.data());
    scan();
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_lo<16>(t >> 6, t & 63, lds.data(), tw.data());
    scan();
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<64>(t, lds.data(), tw.data());
    scan();
    for (uint32_t t = 0; t < 1024; t++) fr4::pass_hi<256>(t, lds.data(), tw.data());
    scan();
    frl sc = frl_zero();
    if (scale) sc = frl_const_from_kilic(*scale);
    for (uint32_t t = 0; t < 1024; t++) { if (scale) fr4::pass_last<true>(t, lds.data(), tw.data(), sc, out); else fr4::pass_last<false>(t, lds.data(), tw.data(), sc, out); }
    return worst;
}
// k_fr_fft4096_r16 lane by lane: 256 lanes x 16 register values, the two transpositions through an LDS area of HALF the transform in two stages
// each (a barrier == finishing the loop over the lanes).  Reports the largest word seen in the LDS area; *conflicts receives the number of
// (stage, access, half wavefront, limb) groups of 32 lanes that did NOT hit 32 different banks (the address maps a1 / a2 promise zero).
uint32_t he_fr_fft4096_r16(const fr *in, uint64_t n_in, fr *out, const fr *roots, uint64_t W, const fr *scale, uint32_t *conflicts) {
    std::vector<uint32_t> tw(fr4::TW_WORDS), lds(9 * fr16::HALF, 0);
    fr4::build_twiddles(roots, W, tw.data());
    struct lane { frl a[16], b[16]; };
    std::vector<lane> L(256);
    uint32_t worst = 0, bad = 0;
    auto scan = [&]() { for (uint32_t v : lds) if (v > worst) worst = v; };
    // bank check of one access of one half wavefront: the 32 word addresses of limb 0
    auto banks = [&](int which, uint32_t t0, auto posfn) {
        uint32_t seen = 0;
        for (uint32_t t = t0; t < t0 + 32; t++) { const uint32_t p = posfn(t), a = which == 1 ? fr16::a1(p) : fr16::a2(p); seen |= 1u << (a & 31u); }
        if (seen != 0xffffffffu) bad++;
    };
    for (uint32_t t = 0; t < 256; t++) { fr x[16]; fr16::load(t, in, n_in, 1, 0, x); fr16::pass_a(x, L[t].b, tw.data()); }
    for (uint32_t s = 0; s < 2; s++) {
        for (uint32_t t = 0; t < 256; t++) fr16::stage_put<1>(lds.data(), L[t].b, 16u * fr16::bitrev8(fr16::lane_a_nat(t)), 1u, s ^ ((t >> 6) & 1u));
        scan();
        for (uint32_t t = 0; t < 256; t++) { uint32_t g, j; fr16::lane_b(t, g, j); fr16::stage_get<1>(lds.data(), L[t].a, 256u * g + j, 16u, s ^ (t >> 7)); }
        for (uint32_t t0 = 0; t0 < 256; t0 += 32)
            for (uint32_t r = 0; r < 8; r++) {
                const uint32_t kw = 8u * (s ^ ((t0 >> 6) & 1u)) + r, kr = 8u * (s ^ (t0 >> 7)) + r;
                banks(1, t0, [&](uint32_t t) { return 16u * fr16::bitrev8(fr16::lane_a_nat(t)) + kw; });
                banks(1, t0, [&](uint32_t t) { uint32_t g, j; fr16::lane_b(t, g, j); return 256u * g + j + 16u * kr; });
            }
    }
    for (uint32_t t = 0; t < 256; t++) { uint32_t g, j; fr16::lane_b(t, g, j); fr16::pass_b(L[t].a, j, tw.data()); }
    for (uint32_t s = 0; s < 2; s++) {
        for (uint32_t t = 0; t < 256; t++) { uint32_t g, j; fr16::lane_b(t, g, j); fr16::stage_put<2>(lds.data(), L[t].a, 256u * g + j, 16u, s ^ ((t >> 6) & 1u)); }
        scan();
        for (uint32_t t = 0; t < 256; t++) fr16::stage_get<2>(lds.data(), L[t].b, t, 256u, s ^ (t >> 7));
        for (uint32_t t0 = 0; t0 < 256; t0 += 32)
            for (uint32_t r = 0; r < 8; r++) {
                const uint32_t kw = 8u * (s ^ ((t0 >> 6) & 1u)) + r, kr = 8u * (s ^ (t0 >> 7)) + r;
                banks(2, t0, [&](uint32_t t) { uint32_t g, j; fr16::lane_b(t, g, j); return 256u * g + j + 16u * kw; });
                banks(2, t0, [&](uint32_t t) { return t + 256u * kr; });
            }
    }
    frl sc = frl_zero();
    if (scale) sc = frl_const_from_kilic(*scale);
    for (uint32_t t = 0; t < 256; t++) { fr16::pass_c(L[t].b, t, tw.data()); if (scale) fr16::store<true>(t, L[t].b, sc, out); else fr16::store<false>(t, L[t].b, sc, out); }
    if (conflicts) *conflicts = bad;
    return worst;
}
uint32_t he_fr_fft_small(uint32_t logm, const fr *in, uint64_t in_stride, uint64_t n_in, uint64_t batch, fr *out, const fr *roots, uint64_t W, const fr *scale) {
    std::vector<uint32_t> tw(fr4::TW_WORDS);
    fr4::build_twiddles(roots, W, tw.data());
    switch (logm) {
    case 2: return fr_fft_small_emul<2>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 3: return fr_fft_small_emul<3>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 4: return fr_fft_small_emul<4>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 5: return fr_fft_small_emul<5>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 6: return fr_fft_small_emul<6>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 7: return fr_fft_small_emul<7>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 8: return fr_fft_small_emul<8>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 9: return fr_fft_small_emul<9>(in, in_stride, n_in, batch, out, tw.data(), scale);
    case 10: return fr_fft_small_emul<10>(in, in_stride, n_in, batch, out, tw.data(), scale);
    default: return fr_fft_small_emul<11>(in, in_stride, n_in, batch, out, tw.data(), scale);
    }
}
void he_fr_fft_long(uint32_t logr, const fr *in, uint64_t n_in, fr *out, const fr *roots, uint64_t W, const fr *scale) {
    switch (logr) {
    case 1: fr_fft_long_emul<1>(in, n_in, out, roots, W, scale); break;
    case 2: fr_fft_long_emul<2>(in, n_in, out, roots, W, scale); break;
    case 3: fr_fft_long_emul<3>(in, n_in, out, roots, W, scale); break;
    default: fr_fft_long_emul<4>(in, n_in, out, roots, W, scale); break;
    }
}
// the eleven passes of k_das_ext2048_r4 lane by lane (in place on vals[2048]); returns the largest raw limb seen in LDS
uint32_t he_das_ext2048(fr *vals, const fr *expanded, const fr *reversed, uint64_t W, const fr *inv_n) {
    std::vector<uint32_t> tw(das2k::TW_WORDS), lds(9 * das2k::NPAD, 0);
    das2k::build_twiddles(expanded, reversed, W, tw.data());
    uint32_t worst = 0;
    auto scan = [&]() { for (uint32_t v : lds) if (v > worst) worst = v; };
    uint32_t *s = lds.data(); const uint32_t *w = tw.data();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_down_first(t, vals, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_down_wide<128>(t, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_down_wide<32>(t, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_down_narrow<8>(t >> 6, t & 63, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_down_narrow<2>(t >> 6, t & 63, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_middle(t >> 6, t & 63, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_up_narrow<2>(t >> 6, t & 63, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_up_narrow<8>(t >> 6, t & 63, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_up_wide<32>(t, s, w);
    scan();
    for (uint32_t t = 0; t < 512; t++) das2k::pass_up_wide<128>(t, s, w);
    scan();
    const frl sc = frl_const_from_kilic(*inv_n);
    for (uint32_t t = 0; t < 512; t++) das2k::pass_up_last(t, s, w, sc, vals);
    return worst;
}
void he_frl_reduce(fr *o, const fr *a, uint32_t k) {             // frl_reduce(a + k r) is congruent to a and below 2 r
    frl x = frl_unpack(*a);
    for (uint32_t i = 0; i < k; i++) { for (int j = 0; j < 9; j++) x.l[j] += frl_p29(j); if (i % 4 == 3) frl_sweep(x); }
    *o = frl_canon_lt2r(frl_reduce(x));
}
// frl_canon on x = a + k r (a canonical, k < 63) presented with raw limbs; frl_mul(a, const(b)) canonicalised == mul(a, b)
void he_frl_canon_of_multiple(fr *o, const fr *a, uint32_t k) {
    frl x = frl_unpack(*a);
    for (uint32_t i = 0; i < k; i++) { for (int j = 0; j < 9; j++) x.l[j] += frl_p29(j); if (i % 4 == 3) frl_sweep(x); }   // raw limbs < 6 * 2^29
    *o = frl_canon(x);
}
void he_frl_mul(fr *o, const fr *a, const fr *b, uint32_t k) {   // a presented as a + k r with raw limbs, k <= 5
    frl x = frl_unpack(*a);
    for (uint32_t i = 0; i < k; i++) for (int j = 0; j < 9; j++) x.l[j] += frl_p29(j);
    *o = frl_canon_lt2r(frl_mul(x, frl_const_from_kilic(*b)));
}
void he_fr_mul(fr *o, const fr *a, const fr *b) { *o = mul(*a, *b); }
void he_fr_add(fr *o, const fr *a, const fr *b) { *o = add(*a, *b); }
void he_fr_sub(fr *o, const fr *a, const fr *b) { *o = sub(*a, *b); }
void he_fr_inv(fr *o, const fr *a) { *o = inv<FrP>(*a); }
void he_fr_inv_fermat(fr *o, const fr *a) { *o = inv_fermat<FrP>(*a); }
void he_fp_inv(fp *o, const fp *a) { *o = inv<FpP>(*a); }
void he_fp_inv_fermat(fp *o, const fp *a) { *o = inv_fermat<FpP>(*a); }
void he_fp_inv_coop(fp *o, const fp *a, uint32_t *rounds, uint32_t *bad) { *o = coop_inv_emul<FpP>(*a, rounds, bad); }
void he_fr_inv_coop(fr *o, const fr *a, uint32_t *rounds, uint32_t *bad) { *o = coop_inv_emul<FrP>(*a, rounds, bad); }
uint64_t he_fp_inv_coop_stress(uint64_t n, uint64_t seed, uint32_t *max_rounds, uint64_t *sum_rounds) { return coop_stress<FpP>(n, seed, max_rounds, sum_rounds); }
uint64_t he_fr_inv_coop_stress(uint64_t n, uint64_t seed, uint32_t *max_rounds, uint64_t *sum_rounds) { return coop_stress<FrP>(n, seed, max_rounds, sum_rounds); }
uint64_t he_fr_inv_stress(uint64_t n, uint64_t seed) { return inv_stress<FrP>(n, seed); }
uint64_t he_fp_inv_stress(uint64_t n, uint64_t seed) { return inv_stress<FpP>(n, seed); }
void he_fr_from_u64(fr *o, uint64_t v) { *o = fr_from_u64(v); }
void he_fr_from_mont(fr *o, const fr *a) { *o = from_mont<FrP>(*a); }
#define IN(p) g1_from_kilic(*(p))
#define OUT(e) g1_to_kilic(e)
void he_g1_add(g1j *o, const g1j *a, const g1j *b) { *o = OUT(g1_add(IN(a), IN(b))); }
void he_g1_sub(g1j *o, const g1j *a, const g1j *b) { *o = OUT(g1_sub(IN(a), IN(b))); }
void he_g1_dbl(g1j *o, const g1j *a) { *o = OUT(g1_dbl(IN(a))); }
void he_g1_madd(g1j *o, const g1j *a, const g1j *b_affine_image) {   // b must have Z = R or be inf
    g1j bi = IN(b_affine_image);
    g1a q; if (is_inf(bi)) q = g1a_inf(); else { q.x = bi.x; q.y = bi.y; }
    *o = OUT(g1_madd(IN(a), q));
}
void he_g1x_madd(g1j *o, const g1j *a, const g1j *b_affine_image) {   // XYZZ accumulator path of the table walks
    g1j bi = IN(b_affine_image);
    g1a q; if (is_inf(bi)) q = g1a_inf(); else { q.x = bi.x; q.y = bi.y; }
    *o = OUT(g1x_to_jac(g1x_madd(g1x_from_jac(IN(a)), q)));
}
// table-walk accumulator (unpacked lazy fast path + generic fallback): acc = sum of the n affine images in `pts` (Z = R or inf)
void he_g1x_acc_sum(g1j *o, const g1j *pts, int n) {
    g1x_acc a; a.init();
    for (int i = 0; i < n; i++) {
        g1j bi = IN(&pts[i]);
        g1a q; if (is_inf(bi)) q = g1a_inf(); else { q.x = bi.x; q.y = bi.y; }
        a.add(q);
    }
    *o = OUT(a.to_jac());
}
void he_g1_mul(g1j *o, const g1j *a, const fr *k_mont) { g1j tbl[15]; *o = OUT(g1_mul_windowed(IN(a), from_mont<FrP>(*k_mont), tbl)); }
void he_g1_mul_small(g1j *o, const g1j *a, uint32_t k) { *o = OUT(g1_mul_small(IN(a), k)); }
void he_g1_normalize(g1j *o, const g1j *a) { *o = OUT(g1_normalize(IN(a))); }
void he_g1_mul_glv(g1j *o, const g1j *a, const fr *k_mont) { g1j tbl[16]; *o = OUT(g1_mul_glv(IN(a), glv_decompose(from_mont<FrP>(*k_mont)), tbl)); }
void he_g1_mul_glv_fast(g1j *o, const g1j *a, const fr *k_mont) {
    g1j pi = IN(a);
    if (is_inf(pi)) { *o = OUT(g1_inf()); return; }
    g1jq tbl[16]; *o = OUT(g1_mul_glv_fast(pi, glv_decompose(from_mont<FrP>(*k_mont)), tbl));
}
void he_g1_mul_glv_wnaf(g1j *o, const g1j *a, const fr *k_mont) {
    g1j pi = IN(a);
    if (is_inf(pi)) { *o = OUT(g1_inf()); return; }
    g1jq_t tbl[8]; int8_t d1[132], d2[132]; *o = OUT(g1_mul_glv_wnaf(pi, glv_decompose(from_mont<FrP>(*k_mont)), tbl, d1, d2, 1));
}
void he_g1_mul_glv_wnaf_inl(g1j *o, const g1j *a, const fr *k_mont) {   // the instantiation the G1 FFT stages run (all products inlined, dot2)
    g1j pi = IN(a);
    if (is_inf(pi)) { *o = OUT(g1_inf()); return; }
    g1jq_t tbl[8]; int8_t d1[132], d2[132]; *o = OUT((g1_mul_glv_wnaf<true, true>(pi, glv_decompose(from_mont<FrP>(*k_mont)), tbl, d1, d2, 1)));
}
// round 2: the same multiplication with the AFFINE 8-entry table (one inversion per multiplication, mixed additions); inlined and
// call-based instantiations
void he_g1_mul_glv_wnaf_affine(g1j *o, const g1j *a, const fr *k_mont, int inl) {
    g1j pi = IN(a);
    if (is_inf(pi)) { *o = OUT(g1_inf()); return; }
    g1aq tbl[8]; fq dz[7]; int8_t d1[132], d2[132]; g1jq q; g1j packed;
    fr kk = glv_decompose(from_mont<FrP>(*k_mont));
    int st = inl ? g1_mul_glv_wnaf_aq<true, true>(pi, kk, tbl, dz, d1, d2, 1, q, packed) : g1_mul_glv_wnaf_aq<false, false>(pi, kk, tbl, dz, d1, d2, 1, q, packed);
    *o = OUT(st == 0 ? g1_inf() : st == 1 ? g1jq_pack(q) : packed);
}
// the regular odd-digit schedule on the same table, scalar split on the fly (what the direct G1 FFT passes run)
void he_g1_mul_glv_regular(g1j *o, const g1j *a, const fr *k_mont, int inl) {
    g1j pi = IN(a);
    if (is_inf(pi)) { *o = OUT(g1_inf()); return; }
    g1aq tbl[8]; fq dz[7]; g1jq q; g1j packed;
    glv_halves h = glv_split_signed(from_mont<FrP>(*k_mont));
    int st = inl ? g1_mul_glv_regular_aq<true>(g1jq_unpack(pi), h, tbl, dz, q, packed) : g1_mul_glv_regular_aq<false>(g1jq_unpack(pi), h, tbl, dz, q, packed);
    *o = OUT(st == 0 ? g1_inf() : st == 1 ? g1jq_pack(q) : packed);
}
// ... with explicit halves (k1[4] | k2[4] | neg1 | neg2): zero halves, even halves, the generic fallback
int he_g1_mul_glv_regular_halves(g1j *o, const g1j *a, const uint32_t *hw, int cold) {
    g1j pi = IN(a);
    glv_halves h;
    for (int i = 0; i < 4; i++) { h.k1[i] = hw[i]; h.k2[i] = hw[4 + i]; }
    h.neg1 = hw[8]; h.neg2 = hw[9];
    if (cold) { g1j r; g1_mul_glv_signed_cold(&r, &pi, &h); *o = OUT(r); return 2; }
    g1aq tbl[8]; fq dz[7]; g1jq q; g1j packed;
    int st = g1_mul_glv_regular_aq<true>(g1jq_unpack(pi), h, tbl, dz, q, packed);
    *o = OUT(st == 0 ? g1_inf() : st == 1 ? g1jq_pack(q) : packed);
    return st;
}
// the 8 affine odd multiples of a (co-Z chain when coz != 0, Jacobian chain + Montgomery's trick otherwise): out = 8 normalised points
int he_wnaf_table(g1j *out8, const g1j *a, int coz) {
    g1jq pq = g1jq_unpack(IN(a));
    g1aq tbl[8]; g1jq jt[8]; fq dz[7];
    int ok = 1;
    fq zc = unpackq(one<FpP>());                           // the co-Z table lives at a common Z (an affine table of an isomorphic curve): (x, y, zc) is the multiple
    if (coz) ok = (coz == 2 ? g1_wnaf_table_affine_coz<true>(pq, tbl, dz, zc) : g1_wnaf_table_affine_coz<false>(pq, tbl, dz, zc)) ? 1 : 0; else g1_wnaf_table_affine_q(pq, tbl, jt);
    for (int i = 0; i < 8; i++) { g1j o; o.x = packq(tbl[i].x); o.y = packq(tbl[i].y); o.z = packq(zc); out8[i] = OUT(g1_normalize(o)); }
    return ok;
}
// acc (= a, any Jacobian image) += sign * phi? * b (normalised to affine here) through g1jq_madd_entry; 1: fast formulas, 0: slow path
int he_g1jq_madd_entry(g1j *o, const g1j *a, const g1j *b, int negate, int phi, int inl) {
    g1jq acc = g1jq_unpack(IN(a));
    g1j bn = g1_normalize(IN(b));
    g1aq t; t.x = unpackq(bn.x); t.y = unpackq(bn.y); t.bx = mulq(t.x, unpackq(glv_beta()));
    bool ok = inl ? g1jq_madd_entry<true>(acc, &t, negate != 0, phi != 0) : g1jq_madd_entry<false>(acc, &t, negate != 0, phi != 0);
    if (ok) { *o = OUT(g1jq_pack(acc)); return 1; }
    bool inf = g1jq_add_slow_copy_a(acc, &t, negate != 0, phi != 0);
    *o = OUT(inf ? g1_inf() : g1jq_pack(acc));
    return 0;
}
// (a + k b, a - k b) through the butterfly's shared lazy formulas; returns 0 when they decline (a == +-k b or an infinite operand)
int he_g1_butterfly(g1j *o_sum, g1j *o_dif, const g1j *a, const g1j *b, const fr *k_mont) {
    g1j x = IN(a), y = IN(b);
    if (is_inf(x) || is_inf(y)) return 0;
    g1aq tbl[8]; fq dz[7]; int8_t d1[132], d2[132]; g1jq yq; g1j packed;
    int st = g1_mul_glv_wnaf_aq<true, true>(y, glv_decompose(from_mont<FrP>(*k_mont)), tbl, dz, d1, d2, 1, yq, packed);   // what k_g1_fft_stage<4> runs
    if (st != 1) return 0;
    g1jq sum, dif;
    if (!g1jq_addsub(g1jq_unpack(x), yq, sum, dif)) return 0;
    *o_sum = OUT(g1jq_pack(sum)); *o_dif = OUT(g1jq_pack(dif));
    return 1;
}
// the P == +-Q handling of the width-5 NAF loop: acc (= a) += sign * b through g1jq_add_entry, falling back to g1jq_add_slow_copy when
// it declines.  Returns 1 when the fast formulas were used, 0 when the slow path ran; *o = the sum (infinity allowed).
int he_g1jq_add_entry(g1j *o, const g1j *a, const g1j *b, int negate) {
    g1jq acc = g1jq_unpack(IN(a));
    g1jq_t t; g1jq_t_make(&t, g1jq_unpack(IN(b)));
    if (g1jq_add_entry<true>(acc, &t, negate != 0, false)) { *o = OUT(g1jq_pack(acc)); return 1; }
    bool inf = g1jq_add_slow_copy(acc, &t, negate != 0, false);
    *o = OUT(inf ? g1_inf() : g1jq_pack(acc));
    return 0;
}
int he_g1_equal(const g1j *a, const g1j *b) { return g1_equal(IN(a), IN(b)); }
}

// balanced GLV split of a standard-form scalar: out = k1[4] | k2[4] | neg1 | neg2 (10 x u32)
extern "C" void he_glv_split_signed(uint32_t *out, const uint32_t *k_std) {
    kzg::fr k;
    for (int i = 0; i < 8; i++) k.l[i] = k_std[i];
    kzg::glv_halves h = kzg::glv_split_signed(k);
    for (int i = 0; i < 4; i++) { out[i] = h.k1[i]; out[4 + i] = h.k2[i]; }
    out[8] = h.neg1; out[9] = h.neg2;
}
