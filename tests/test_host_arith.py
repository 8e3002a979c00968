"""The device arithmetic headers (field.hpp / g1.hpp), compiled for the host, against the oracle. CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import koracle as ko

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host", "host_emul.cpp")
OUT = os.path.join(HERE, "host", "_build", "libhost_emul.so")


@pytest.fixture(scope="module")
def he():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = os.path.join(ROOT, "go-kzg_amd", "csrc")
    deps = [SRC] + [os.path.join(inc, h) for h in ("field.hpp", "g1.hpp", "fr_lazy.hpp", "fr_fft4096.hpp", "fr_das2048.hpp", "coop_inv.hpp")] + [os.path.join(ROOT, "tools", "ab_fr_r16", "fr16.hpp")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", inc, "-o", OUT, SRC])   # -O1: half the build time of -O2 (the unrolled passes of ten transform sizes), same run time within seconds
    return C.CDLL(OUT)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def rand_fr(rng, n):
    return ko.fr_from_ints([int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)])


def test_fr_ops(he):
    rng = np.random.default_rng(1)
    a, b = rand_fr(rng, 64), rand_fr(rng, 64)
    edge = ko.fr_from_ints([0, 1, ko.R_MOD - 1, 2, ko.R_MOD - 2])
    a[:5], b[5:10] = edge, edge
    L = ko.lib()
    for i in range(64):
        for name in ("mul", "add", "sub"):
            got, want = ko.fr_empty(1), ko.fr_empty(1)
            getattr(he, "he_fr_" + name)(p(got), p(a[i]), p(b[i]))
            getattr(L, "ko_fr_" + name)(p(want), p(a[i]), p(b[i]))
            assert np.array_equal(got, want), (name, i)
    for i in range(8):
        got, want = ko.fr_empty(1), ko.fr_empty(1)
        he.he_fr_inv(p(got), p(a[i]))
        L.ko_fr_inv(p(want), p(a[i]))
        assert np.array_equal(got, want)
    got = ko.fr_empty(1)
    he.he_fr_from_u64.argtypes = [C.c_void_p, C.c_uint64]
    he.he_fr_from_u64(p(got), 2**64 - 5)
    assert ko.fr_to_ints(got) == [2**64 - 5]


P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def test_binary_gcd_inversion_matches_fermat_and_bigint(he):
    # inv() (binary GCD on 30-bit limbs) == inv_fermat() == pow(x, -1, m) in the Montgomery domain, F_r (R = 2^256) and F_p (R' = 2^390)
    rng = np.random.default_rng(11)
    for name, mod, words, rbits in (("fr", ko.R_MOD, 8, 256), ("fp", P_MOD, 12, 390)):
        edge = [0, 1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, 2**30, 2**30 - 1, 2**60, 2**64 - 1, 2**64, 2**65 + 1,
                2**90, 2**(mod.bit_length() - 1), 2**(mod.bit_length() - 1) - 1, mod - 2**30, mod - 2**64, 0x3fffffff << 30,
                (1 << 200) - 1, 1 << 200, (1 << 120) + 1, 3**100 % mod, 5**150 % mod]
        edge += [pow(2, -k, mod) for k in (1, 30, 31, 255, 381, 390)] + [(mod >> k) for k in (1, 2, 29, 30, 31, 64, 100, 300)]
        vals = edge + [int.from_bytes(rng.bytes(48), "little") % mod for _ in range(400)]
        vals += [int.from_bytes(rng.bytes(48), "little") % mod >> int(rng.integers(0, mod.bit_length())) for _ in range(400)]
        R = pow(2, rbits, mod)
        for x in vals:
            img = np.frombuffer(((x * R) % mod).to_bytes(4 * words, "little"), dtype=np.uint32).copy()
            got, ferm = np.zeros(words, dtype=np.uint32), np.zeros(words, dtype=np.uint32)
            getattr(he, "he_%s_inv" % name)(p(got), p(img))
            getattr(he, "he_%s_inv_fermat" % name)(p(ferm), p(img))
            want = (pow(x, -1, mod) * R) % mod if x else 0
            assert int.from_bytes(got.tobytes(), "little") == want, (name, hex(x))
            assert np.array_equal(got, ferm), (name, hex(x))


def test_cooperative_inversion_emulation_matches_the_lane_form(he):
    """wave_inv_fp (coop_inv.hpp: limbs across 16 lanes, safegcd divsteps on the low limb, two carry hand-overs per round) replayed lane by lane on the host with the
    device's own scalar pieces: word for word inv<FpP>() -- which is pinned to Fermat and to Python integers above -- on edge values and on 20 000 structured / random
    elements, with the invariants the device code relies on (centred limbs within 2^29 + 2, products within 2^61, unique zero) checked in every round"""
    P = ko.P_MOD if hasattr(ko, "P_MOD") else 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    R = 1 << 390
    edge = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, 1 << 380, (1 << 381) - 1 - ((1 << 381) - P), 5, 1 << 200, (1 << 30) - 1, 1 << 30, (1 << 360) + 1]
    for x in edge:
        img = np.frombuffer((x * R % P).to_bytes(48, "little"), dtype=np.uint32).copy()
        got, want = np.zeros(12, dtype=np.uint32), np.zeros(12, dtype=np.uint32)
        rounds, bad = C.c_uint32(0), C.c_uint32(0)
        he.he_fp_inv_coop(p(got), p(img), C.byref(rounds), C.byref(bad))
        he.he_fp_inv(p(want), p(img))
        assert bad.value == 0 and np.array_equal(got, want), hex(x)
        if x % P:
            assert int.from_bytes(got.tobytes(), "little") == pow(x, -1, P) * R % P, hex(x)
            assert 1 <= rounds.value <= 30, (hex(x), rounds.value)
    he.he_fp_inv_coop_stress.restype = C.c_uint64
    he.he_fp_inv_coop_stress.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    mx, sm = C.c_uint32(0), C.c_uint64(0)
    assert he.he_fp_inv_coop_stress(20000, 2026, C.byref(mx), C.byref(sm)) == 0
    assert mx.value <= 30 and 20 < sm.value / 20000 < 28.5, (mx.value, sm.value / 20000)
    # the F_r instance (9 limbs, R = 2^256: the workgroup batch inversion of eth.ComputeKZGProof's quotient kernel)
    r_mod, R_r = ko.R_MOD, 1 << 256
    for x in [0, 1, 2, r_mod - 1, r_mod - 2, (r_mod + 1) // 2, 1 << 254, 5, (1 << 30) - 1, 1 << 30, (1 << 240) + 1]:
        img = np.frombuffer((x * R_r % r_mod).to_bytes(32, "little"), dtype=np.uint32).copy()
        got, want = np.zeros(8, dtype=np.uint32), np.zeros(8, dtype=np.uint32)
        rounds, bad = C.c_uint32(0), C.c_uint32(0)
        he.he_fr_inv_coop(p(got), p(img), C.byref(rounds), C.byref(bad))
        he.he_fr_inv(p(want), p(img))
        assert bad.value == 0 and np.array_equal(got, want), hex(x)
        if x % r_mod:
            assert int.from_bytes(got.tobytes(), "little") == pow(x, -1, r_mod) * R_r % r_mod, hex(x)
    he.he_fr_inv_coop_stress.restype = C.c_uint64
    he.he_fr_inv_coop_stress.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    assert he.he_fr_inv_coop_stress(20000, 2027, C.byref(mx), C.byref(sm)) == 0
    assert mx.value <= 21 and 12 < sm.value / 20000 < 19.5, (mx.value, sm.value / 20000)


def test_binary_gcd_inversion_stress(he):
    for f in (he.he_fr_inv_stress, he.he_fp_inv_stress):
        f.restype, f.argtypes = C.c_uint64, [C.c_uint64, C.c_uint64]
        assert f(60000, 12345) == 0


def test_g1_group_law(he):
    rng = np.random.default_rng(2)
    gen = ko.g1_generator()
    ks = rand_fr(rng, 6)
    pts = [ko.g1_mul(gen, k) for k in ks]            # Jacobian, Z != 1
    pts.append(ko.g1_zero()[0])                        # inf
    pts.append(pts[0].copy())                          # P == P (same representation)
    pts.append(ko.g1_affine(pts[0])[0])                # P == P (different representation)
    pts.append(ko.g1_sub(ko.g1_zero()[0], pts[0]))     # -P
    pts.append(gen)
    pts = np.stack(pts)
    L = ko.lib()
    for i in range(len(pts)):
        got, want = ko.g1_empty(1), ko.g1_empty(1)
        he.he_g1_dbl(p(got), p(pts[i])); L.ko_g1_dbl(p(want), p(pts[i]))
        assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
        he.he_g1_normalize(p(got), p(pts[i]))
        assert np.array_equal(got, ko.g1_affine(pts[i]))   # bit-exact normalised image
        for j in range(len(pts)):
            for name in ("add", "sub"):
                getattr(he, "he_g1_" + name)(p(got), p(pts[i]), p(pts[j]))
                getattr(L, "ko_g1_" + name)(p(want), p(pts[i]), p(pts[j]))
                assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), (name, i, j)
            qa = ko.g1_affine(pts[j])[0]
            he.he_g1_madd(p(got), p(pts[i]), p(qa)); L.ko_g1_add(p(want), p(pts[i]), p(qa))
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), ("madd", i, j)
            he.he_g1x_madd(p(got), p(pts[i]), p(qa))
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), ("xyzz madd", i, j)


def test_g1_scalar_mul(he):
    rng = np.random.default_rng(3)
    gen = ko.g1_generator()
    base = ko.g1_mul(gen, rand_fr(rng, 1)[0])
    scalars = list(rand_fr(rng, 10)) + list(ko.fr_from_ints([0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 1023, 1024, ko.R_MOD - 1, 2**64, (1 << 255) % ko.R_MOD, 0xac45a4010001a40200000000ffffffff, 0xac45a4010001a40200000000fffffffe, 0xac45a4010001a40200000000ffffffff * 5 + 3]))
    L = ko.lib()
    for k in scalars:
        for pt in (base, gen, ko.g1_zero()[0]):
            got, want = ko.g1_empty(1), ko.g1_empty(1)
            he.he_g1_mul(p(got), p(pt), p(k)); L.ko_g1_mul(p(want), p(pt), p(k))
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
            he.he_g1_mul_glv(p(got), p(pt), p(k))       # GLV path (generic arithmetic)
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
            he.he_g1_mul_glv_fast(p(got), p(pt), p(k))  # GLV on unpacked lazy coordinates, signed 5-bit windows
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
            he.he_g1_mul_glv_wnaf(p(got), p(pt), p(k))  # width-5 NAF, 8 odd multiples, every product a call
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
            he.he_g1_mul_glv_wnaf_inl(p(got), p(pt), p(k))  # round-1 instantiation: Jacobian table, inlined products, merged reductions
            assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want))
            for inl in (0, 1):                           # round 2: affine table (one inversion), mixed additions; what the G1 FFT stages run
                he.he_g1_mul_glv_wnaf_affine(p(got), p(pt), p(k), inl)
                assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), inl
                he.he_g1_mul_glv_regular(p(got), p(pt), p(k), inl)   # regular odd-digit schedule, scalar split on the fly (direct FFT passes)
                assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), inl
    he.he_g1_mul_small.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    for k in (0, 1, 2, 255, 4096, 2**32 - 1):
        got = ko.g1_empty(1)
        he.he_g1_mul_small(p(got), p(base), k)
        assert ko.g1_equal(got[0], ko.g1_mul(base, ko.fr_from_ints([k])[0]))


def test_regular_odd_digit_schedule_halves(he):
    """g1_mul_glv_regular_aq on explicit halves: s1 k1 P + s2 k2 phi(P) with zero / even / odd / full-width halves and every sign
    pattern, against the big-integer value k1 s1 + k2 s2 lambda, and the cold generic path on the same inputs."""
    rng = np.random.default_rng(11)
    base = ko.g1_mul(ko.g1_generator(), rand_fr(rng, 1)[0])
    # lambda: the eigenvalue of phi on G1, recovered from the split of a known scalar: k = s1 k1 + s2 k2 lambda (mod r)
    probe = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef % ko.R_MOD
    out = np.zeros(10, dtype=np.uint32)
    ks = np.array([(probe >> (32 * i)) & 0xffffffff for i in range(8)], dtype=np.uint32)
    he.he_glv_split_signed(p(out), p(ks))
    k1 = sum(int(out[i]) << (32 * i) for i in range(4)) * (-1 if out[8] else 1)
    k2 = sum(int(out[4 + i]) << (32 * i) for i in range(4)) * (-1 if out[9] else 1)
    lam = (probe - k1) * pow(k2, -1, ko.R_MOD) % ko.R_MOD
    assert (lam * lam + lam + 1) % ko.R_MOD == 0
    he.he_g1_mul_glv_regular_halves.restype = C.c_int
    he.he_g1_mul_glv_regular_halves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    mags = [0, 1, 2, 3, 14, 15, 16, 17, 30, 31, 32, 2**64, 2**64 - 1, 2**127, 2**128 - 1, 2**128 - 2] + [int(rng.integers(0, 2**63)) ** 2 + int(rng.integers(0, 2)) for _ in range(4)]
    cases = [(a, b) for a in mags[:8] for b in mags[:8]] + [(a, b) for a in mags[8:] for b in (0, 1, 2, mags[-1], 2**128 - 1)] + [(b, a) for a in mags[8:] for b in (0, 2, mags[-2])]
    for n, (m1, m2) in enumerate(cases):
        s1, s2 = (n >> 0) & 1, (n >> 1) & 1
        hw = np.array([(m1 >> (32 * i)) & 0xffffffff for i in range(4)] + [(m2 >> (32 * i)) & 0xffffffff for i in range(4)] + [s1, s2], dtype=np.uint32)
        k = ((-m1 if s1 else m1) + (-m2 if s2 else m2) * lam) % ko.R_MOD
        want = ko.g1_mul(base, ko.fr_from_ints([k])[0])
        got = ko.g1_empty(1)
        st = he.he_g1_mul_glv_regular_halves(p(got), p(base), p(hw), 0)
        assert ko.g1_equal(got[0], want), (m1, m2, s1, s2, st)
        assert st == (0 if (m1 == 0 and m2 == 0) else 1), (m1, m2, st)
        if n % 7 == 0:
            he.he_g1_mul_glv_regular_halves(p(got), p(base), p(hw), 1)
            assert ko.g1_equal(got[0], want), ("cold", m1, m2, s1, s2)


def test_wnaf_loop_exceptional_additions(he):
    # acc += +-entry inside g1_mul_glv_wnaf: distinct points use the fast formulas; acc == entry (doubling) and acc == -entry
    # (infinity) are declined by them and handled by the complete formulas
    rng = np.random.default_rng(6)
    gen = ko.g1_generator()
    a = ko.g1_mul(gen, rand_fr(rng, 1)[0])
    b = ko.g1_mul(gen, rand_fr(rng, 1)[0])
    out = ko.g1_empty(1)
    assert he.he_g1jq_add_entry(p(out), p(a), p(b), 0) == 1 and ko.g1_equal(out[0], ko.g1_add(a, b))
    assert he.he_g1jq_add_entry(p(out), p(a), p(b), 1) == 1 and ko.g1_equal(out[0], ko.g1_sub(a, b))
    assert he.he_g1jq_add_entry(p(out), p(a), p(a), 0) == 0 and ko.g1_equal(out[0], ko.g1_add(a, a))
    assert he.he_g1jq_add_entry(p(out), p(a), p(a), 1) == 0 and ko.g1_equal(out[0], ko.g1_zero()[0])
    a2 = ko.g1_add(a, a)                                   # same point, different Jacobian representation (Z != 1)
    assert he.he_g1jq_add_entry(p(out), p(a2), p(ko.g1_add(a, a)), 0) == 0 and ko.g1_equal(out[0], ko.g1_add(a2, a2))


def test_wnaf_affine_tables(he):
    # the 8 odd multiples P, 3P .. 15P as affine points: co-Z chain (what the kernels run) and the Jacobian chain fallback, for
    # canonical and non-canonical (Z != 1) inputs
    rng = np.random.default_rng(8)
    gen = ko.g1_generator()
    out = ko.g1_empty(8)
    for trial in range(6):
        pt = ko.g1_mul(gen, rand_fr(rng, 1)[0])
        if trial & 1:
            pt = ko.g1_add(pt, ko.g1_mul(gen, rand_fr(rng, 1)[0]))          # Jacobian image with Z != 1
        want = [ko.g1_mul(pt, ko.fr_from_ints([2 * i + 1])[0]) for i in range(8)]
        for coz in (2, 1, 0):                              # 2: products inlined
            assert he.he_wnaf_table(p(out), p(pt), coz) == 1
            for i in range(8):
                assert np.array_equal(out[i], ko.g1_affine(want[i][None])[0]), (trial, coz, i)


def test_wnaf_mixed_addition_and_its_exceptional_cases(he):
    # acc += +-(phi?) entry with an AFFINE entry (g1jq_madd_entry): accumulators in arbitrary Jacobian images, long chains (the lazy
    # bounds), doubling / cancellation declined by the fast formulas and handled by the complete ones
    rng = np.random.default_rng(16)
    gen = ko.g1_generator()
    lam = ko.fr_from_ints([0xac45a4010001a40200000000ffffffff])[0]
    a = ko.g1_add(ko.g1_mul(gen, rand_fr(rng, 1)[0]), ko.g1_mul(gen, rand_fr(rng, 1)[0]))      # Z != 1
    b = ko.g1_mul(gen, rand_fr(rng, 1)[0])
    out = ko.g1_empty(1)
    for inl in (0, 1):
        for neg in (0, 1):
            for phi in (0, 1):
                q = ko.g1_mul(b, lam) if phi else b
                want = ko.g1_sub(a, q) if neg else ko.g1_add(a, q)
                assert he.he_g1jq_madd_entry(p(out), p(a), p(b), neg, phi, inl) == 1 and ko.g1_equal(out[0], want), (inl, neg, phi)
        assert he.he_g1jq_madd_entry(p(out), p(a), p(a), 0, 0, inl) == 0 and ko.g1_equal(out[0], ko.g1_add(a, a))
        assert he.he_g1jq_madd_entry(p(out), p(a), p(a), 1, 0, inl) == 0 and ko.g1_equal(out[0], ko.g1_zero()[0])
        acc, want = a.copy(), a.copy()                   # a chain of 300 mixed additions keeps the bound invariant
        for i in range(300):
            assert he.he_g1jq_madd_entry(p(out), p(acc), p(b), i & 1, (i >> 1) & 1, inl) == 1
            q = ko.g1_mul(b, lam) if (i >> 1) & 1 else b
            want = ko.g1_sub(want, q) if i & 1 else ko.g1_add(want, q)
            acc = out[0].copy()
        assert ko.g1_equal(acc, want)


def test_fft_butterfly_shared_add_sub(he):
    # (x + w y, x - w y) of k_g1_fft_stage: shared lazy formulas == oracle; they decline (return 0) when x == +-w y
    rng = np.random.default_rng(5)
    gen = ko.g1_generator()
    L = ko.lib()
    for trial in range(6):
        kx, ky, w = rand_fr(rng, 3)
        x, y = ko.g1_mul(gen, kx), ko.g1_mul(gen, ky)
        wy = ko.g1_mul(y, w)
        s_, d_ = ko.g1_empty(1), ko.g1_empty(1)
        assert he.he_g1_butterfly(p(s_), p(d_), p(x), p(y), p(w)) == 1
        assert ko.g1_equal(s_[0], ko.g1_add(x, wy)) and ko.g1_equal(d_[0], ko.g1_sub(x, wy))
    w = rand_fr(rng, 1)[0]
    y = ko.g1_mul(gen, rand_fr(rng, 1)[0])
    wy = ko.g1_mul(y, w)
    s_, d_ = ko.g1_empty(1), ko.g1_empty(1)
    assert he.he_g1_butterfly(p(s_), p(d_), p(wy), p(y), p(w)) == 0                    # x == w y: doubling, generic path
    assert he.he_g1_butterfly(p(s_), p(d_), p(ko.g1_sub(ko.g1_zero()[0], wy)), p(y), p(w)) == 0   # x == -w y
    assert he.he_g1_butterfly(p(s_), p(d_), p(ko.g1_zero()[0]), p(y), p(w)) == 0


def test_table_walk_accumulator_fast_path(he):
    """g1x_acc (unpacked lazy XYZZ mixed additions, generic fallback for P == +-Q) over long chains and edge patterns"""
    rng = np.random.default_rng(9)
    gen = ko.g1_generator()
    he.he_g1x_acc_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    base = [ko.g1_affine(ko.g1_mul(gen, k))[0] for k in rand_fr(rng, 24)]
    inf = ko.g1_zero()[0]
    neg0 = ko.g1_affine(ko.g1_sub(inf, base[0]))[0]
    patterns = {
        "random chain": base * 12,                                    # 288 additions: the bounds invariant must hold forever
        "with infinities": [inf, base[0], inf, base[1], inf],
        "P + P": [base[0], base[0], base[1]],
        "P - P then more": [base[0], neg0, base[2], base[3]],
        "sum hits a later operand": [base[0], base[1], ko.g1_affine(ko.g1_add(base[0], base[1]))[0], base[4]],
        "sum hits minus a later operand": [base[0], base[1], ko.g1_affine(ko.g1_sub(inf, ko.g1_add(base[0], base[1])))[0], base[5]],
        "only infinity": [inf, inf],
        "empty": [],
    }
    for name, pts in patterns.items():
        arr = np.stack(pts) if pts else ko.g1_empty(1)
        got = ko.g1_empty(1)
        he.he_g1x_acc_sum(p(got), p(arr), len(pts))
        want = inf
        for q in pts:
            want = ko.g1_add(want, q)
        assert np.array_equal(ko.g1_compress(got), ko.g1_compress(want)), name


def test_glv_split_signed_for_variable_scalars(he):
    """the device-side balanced GLV split of the bucket MSM: k == s1 |k1| + s2 |k2| lambda (mod r), both magnitudes < 2^126.5
    (16 signed 8-bit windows, no carry out of the top one) -- against Python big integers, structured and random scalars"""
    lam = 0xac45a4010001a40200000000ffffffff
    r = ko.R_MOD
    assert (lam * lam + lam + 1) % r == 0
    rng = np.random.default_rng(7)
    vals = [0, 1, 2, lam - 1, lam, lam + 1, lam // 2, lam // 2 + 1, (r - 1) // 2, (r - 1) // 2 + 1, r - 1, r - 2, r - lam, 2 * lam, 2 * lam - 1,
            (lam + 1) * (lam // 2), (1 << 254) % r, (1 << 255) % r, r - lam // 2, r - lam // 2 - 1]
    vals += [(q * lam + d) % r for q in (0, 1, lam // 2 - 1, lam // 2, lam // 2 + 1, lam - 1) for d in (-2, -1, 0, 1, 2, lam // 2, lam // 2 + 1)]
    vals += [int.from_bytes(rng.bytes(32), "little") % r for _ in range(20000)]
    out = np.zeros(10, dtype=np.uint32)
    for v in vals:
        k = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint32).copy()
        he.he_glv_split_signed(p(out), p(k))
        k1 = sum(int(out[i]) << (32 * i) for i in range(4))
        k2 = sum(int(out[4 + i]) << (32 * i) for i in range(4))
        assert out[8] in (0, 1) and out[9] in (0, 1)
        s1, s2 = (-1 if out[8] else 1), (-1 if out[9] else 1)
        assert (s1 * k1 + s2 * k2 * lam - v) % r == 0, hex(v)
        assert k1 <= lam // 2 + 1 and k2 < (1 << 127) and k1 < (1 << 127), hex(v)
        assert (k1 >> 120) < 127 and (k2 >> 120) < 127, hex(v)       # top signed window: raw + carry <= 128, no carry out


# ---- lazy 29-bit F_r arithmetic and the radix-4 4096-point transform built on it (fr_lazy.hpp, fr_fft4096.hpp) ----
def test_frl_mul_and_canon(he):
    rng = np.random.default_rng(29)
    a, b = rand_fr(rng, 96), rand_fr(rng, 96)
    edge = ko.fr_from_ints([0, 1, ko.R_MOD - 1, 2, ko.R_MOD - 2, (1 << 232) - 1, 1 << 232])
    a[:7], b[7:14] = edge, edge
    L = ko.lib()
    he.he_frl_canon_of_multiple.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    he.he_frl_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    he.he_frl_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    for i in range(96):
        for k in (0, 1, 2, 17, 38, 62):
            got = ko.fr_empty(1)
            he.he_frl_canon_of_multiple(p(got), p(a[i]), k)
            assert np.array_equal(got, a[i:i + 1]), (i, k)
            he.he_frl_reduce(p(got), p(a[i]), k)
            assert np.array_equal(got, a[i:i + 1]), ("reduce", i, k)
        want = ko.fr_empty(1)
        L.ko_fr_mul(p(want), p(a[i]), p(b[i]))
        for k in (0, 1, 5):
            got = ko.fr_empty(1)
            he.he_frl_mul(p(got), p(a[i]), p(b[i]), k)
            assert np.array_equal(got, want), (i, k)


@pytest.mark.parametrize("n_in,inv", [(4096, False), (4096, True), (2048, False), (1, False), (0, False), (4095, True)])
def test_fr_fft4096_radix4_emulation_matches_oracle(he, n_in, inv):
    """the passes of k_fr_fft4096_r4 run lane by lane on the host == the oracle's recursive radix-2 FFT (fft_fr.go:30-105), bit for bit;
    raw limbs in LDS stay below 6 * 2^29"""
    rng = np.random.default_rng(4096 + n_in + inv)
    fs = ko.FFTSettings(13)                                   # W = 8192: the twiddle file strides through a wider root table
    vals = rand_fr(rng, 4096)
    if n_in >= 8:
        vals[:3] = ko.fr_from_ints([0, ko.R_MOD - 1, 1])
    padded = vals.copy()
    padded[n_in:] = 0
    want = fs.fft(padded, inv=inv)
    roots = fs.reverse_roots() if inv else fs.expanded_roots()
    out = ko.fr_empty(4096)
    scale = None
    if inv:
        scale = ko.fr_from_ints([pow(4096, -1, ko.R_MOD)])
    he.he_fr_fft4096.restype = C.c_uint32
    he.he_fr_fft4096.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    worst = he.he_fr_fft4096(p(np.ascontiguousarray(vals[:max(n_in, 1)])), n_in, p(out), p(roots), 8192, p(scale) if inv else None)
    assert np.array_equal(out, want)
    assert worst < 6 * 2**29


@pytest.mark.parametrize("n_in,inv", [(4096, False), (4096, True), (2048, False), (1, False), (0, False), (3000, True)])
def test_fr_fft4096_r16_emulation_matches_oracle(he, n_in, inv):
    """k_fr_fft4096_r16 (256 lanes x 16 register-resident values, two transpositions through an LDS area of half the transform) lane by lane on the
    host == the oracle's FFT (fft_fr.go:30-105), bit for bit; what crosses the LDS stays below 6 * 2^29; every half wavefront hits 32 different
    banks in every write and every read of both transpositions (the address maps a1 / a2)"""
    rng = np.random.default_rng(1600 + n_in + inv)
    fs = ko.FFTSettings(13)
    vals = rand_fr(rng, 4096)
    if n_in >= 8:
        vals[:3] = ko.fr_from_ints([0, ko.R_MOD - 1, 1])
    padded = vals.copy()
    padded[n_in:] = 0
    want = fs.fft(padded, inv=inv)
    roots = fs.reverse_roots() if inv else fs.expanded_roots()
    out = ko.fr_empty(4096)
    scale = ko.fr_from_ints([pow(4096, -1, ko.R_MOD)]) if inv else None
    conflicts = C.c_uint32(99)
    he.he_fr_fft4096_r16.restype = C.c_uint32
    he.he_fr_fft4096_r16.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    worst = he.he_fr_fft4096_r16(p(np.ascontiguousarray(vals[:max(n_in, 1)])), n_in, p(out), p(roots), 8192, p(scale) if inv else None, C.byref(conflicts))
    assert np.array_equal(out, want)
    assert worst < 6 * 2**29
    assert conflicts.value == 0


@pytest.mark.parametrize("logm", list(range(2, 12)))
def test_fr_fft_small_emulation_matches_oracle(he, logm):
    """k_fr_fft_small lane by lane on the host: 4096 / m transforms of m = 4 .. 2048 points through the first passes of the 4096-point network (odd
    log2 m: one more radix-2 pass) == the oracle's FFT of every row, both directions, a partly filled workgroup, zero padding; in a settings object of
    exactly m points (the partial twiddle file) and in a 8192-wide one; raw limbs in LDS below 6 * 2^29"""
    m = 1 << logm
    per = 4096 // m
    rng = np.random.default_rng(100 + logm)
    he.he_fr_fft_small.restype = C.c_uint32
    he.he_fr_fft_small.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    for max_scale in (logm, 13):
        fs = ko.FFTSettings(max_scale)
        for batch, n_in, inv in ((per, m, False), (per, m, True), (max(per // 2, 1), m, False), (per, m // 2 + 1, False)):
            rows = rand_fr(rng, per * m).reshape(per, m, 4)
            rows[0, :2] = ko.fr_from_ints([ko.R_MOD - 1, 0])
            padded = rows.copy()
            padded[:, n_in:] = 0
            out = np.full((per, m, 4), 0xAA, dtype=np.uint64)
            scale = ko.fr_from_ints([pow(m, -1, ko.R_MOD)]) if inv else None
            roots = fs.reverse_roots() if inv else fs.expanded_roots()
            worst = he.he_fr_fft_small(logm, p(np.ascontiguousarray(rows)), m, n_in, batch, p(out), p(roots), 1 << max_scale, p(scale) if inv else None)
            assert worst < 6 * 2**29
            for b in sorted({0, batch // 2, batch - 1}):
                assert np.array_equal(out[b], fs.fft(padded[b], inv=inv)), (max_scale, batch, n_in, inv, b)
            if batch < per:
                assert (out[batch:] == 0xAA).all()                               # rows that do not exist are not stored


@pytest.mark.parametrize("logr", [1, 2, 3, 4])
def test_fr_fft_long_emulation_matches_oracle(he, logr):
    """a transform of R * 4096 points as the device runs it -- rows through the 4096-point passes with element stride R, then fr4::upper_lane for every
    k2 (k_fr_fft_upper) -- == the oracle's FFT, forward with zero padding and inverse with the 1 / n scale, in a wider settings object"""
    n = 4096 << logr
    rng = np.random.default_rng(200 + logr)
    fs = ko.FFTSettings(12 + logr + (1 if logr < 4 else 0))
    he.he_fr_fft_long.restype = None
    he.he_fr_fft_long.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    vals = rand_fr(rng, n)
    vals[:3] = ko.fr_from_ints([0, ko.R_MOD - 1, 1])
    for inv, n_in in ((False, n), (True, n), (False, n // 2 + 5)):
        padded = vals.copy()
        padded[n_in:] = 0
        out = ko.fr_empty(n)
        scale = ko.fr_from_ints([pow(n, -1, ko.R_MOD)]) if inv else None
        roots = fs.reverse_roots() if inv else fs.expanded_roots()
        he.he_fr_fft_long(logr, p(np.ascontiguousarray(vals)), n_in, p(out), p(roots), fs.max_width, p(scale) if inv else None)
        assert np.array_equal(out, fs.fft(padded, inv=inv)), (logr, inv, n_in)


@pytest.mark.parametrize("scale", [12, 13])
def test_das_ext2048_lazy_emulation_matches_oracle(he, scale):
    """the passes of k_das_ext2048_r4 on the host == the oracle's recursive dASFFTExtension (das_extension.go:7-84) bit for bit, with the
    exact-width domain (scale 12) and with a wider one (scale 13: the reference walks the full-width tables without rescaling the indices)"""
    rng = np.random.default_rng(2048 + scale)
    fs = ko.FFTSettings(scale)
    vals = rand_fr(rng, 2048)
    vals[:3] = ko.fr_from_ints([0, ko.R_MOD - 1, 1])
    want = fs.das_fft_extension(vals.copy())
    got = vals.copy()
    inv_n = ko.fr_from_ints([pow(2048, -1, ko.R_MOD)])
    he.he_das_ext2048.restype = C.c_uint32
    he.he_das_ext2048.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    worst = he.he_das_ext2048(p(got), p(fs.expanded_roots()), p(fs.reverse_roots()), 1 << scale, p(inv_n))
    assert np.array_equal(got, want)
    assert worst < 6 * 2**29
