"""Pins the CPU oracle (oracle/kzg_oracle.c) to every known-answer datum the reference holds for the
hot path and to an independent big-int restatement (oracle/pyref.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import koracle as ko
from oracle import pyref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
DERIVED = json.load(open(os.path.join(GOLDEN, "derived_vectors.json")))
PINS = json.load(open(os.path.join(GOLDEN, "trusted_setup_sha256.json")))
S_TEST = int(KATS["test_secret"]["value"])
TEST_POLY = KATS["test_poly"]["values"]


def hexpt(p):
    return ko.g1_compress(p)[0].tobytes().hex()


def pt_from_affine_ints(a):
    """affine big-int point -> oracle image via the compressed format"""
    return ko.g1_decompress(np.frombuffer(pyref.g1_compress(a), dtype=np.uint8))[0]


# ---------------------------------------------------------------- constants
def test_modulus_and_roots_of_unity_table():
    assert int(KATS["modulus"]["value"]) == ko.R_MOD == pyref.R
    out = ko.fr_empty(1)
    for k, dec in enumerate(KATS["scale2_root_of_unity"]["values"]):
        ko.lib().ko_scale2_root_of_unity(out.ctypes.data, k)
        assert ko.fr_to_ints(out)[0] == int(dec), k  # bls/globals.go:27-60


def test_generator_matches_in_tree_decimals():
    g = ko.g1_generator()
    assert hexpt(g) == pyref.g1_compress((int(KATS["g1_generator"]["x"]), int(KATS["g1_generator"]["y"]))).hex()


def test_fr_montgomery_image_constants():
    # R_r = 2^256 mod r is the image of ONE
    one = ko.fr_from_ints([1])[0]
    assert sum(int(one[i]) << (64 * i) for i in range(4)) == (1 << 256) % ko.R_MOD
    g = ko.g1_generator()
    rp = (1 << 384) % pyref.P
    assert sum(int(g[2][i]) << (64 * i) for i in range(6)) == rp
    assert sum(int(g[0][i]) << (64 * i) for i in range(6)) == pyref.GX * rp % pyref.P


def test_fr_from32_range_rule():
    # bls.ValidFr (bls/bignum_all.go:12-35): v < r
    ok, _ = ko.fr_from_le32((ko.R_MOD - 1).to_bytes(32, "little"))
    assert ok
    ok, _ = ko.fr_from_le32(ko.R_MOD.to_bytes(32, "little"))
    assert not ok
    ok, _ = ko.fr_from_le32(b"\xff" * 32)
    assert not ok


# ---------------------------------------------------------------- reference known-answer tests
def test_inv_fft_kat():
    k = KATS["test_inv_fft"]  # fft_fr_test.go:32-71
    fs = ko.FFTSettings(k["scale"])
    res = ko.fr_to_ints(fs.fft(ko.fr_from_ints(k["input"]), inv=True))
    assert res == [int(v) for v in k["expected"]]
    assert pyref.FFTSettings(k["scale"]).fft(k["input"], inv=True) == res


def test_fft_roundtrip():
    fs = ko.FFTSettings(4)  # fft_fr_test.go:9-30
    data = ko.fr_from_ints(range(16))
    assert np.array_equal(fs.fft(fs.fft(data), inv=True), data)


def test_das_fft_extension_kat():
    k = KATS["test_das_fft_extension"]  # das_extension_test.go:11-40
    fs = ko.FFTSettings(k["scale"])
    res = ko.fr_to_ints(fs.das_fft_extension(ko.fr_from_ints(k["input"])))
    assert res == [int(v) for v in k["expected"]]
    assert pyref.FFTSettings(k["scale"]).das_fft_extension(k["input"]) == res


@pytest.mark.parametrize("scale", [4, 5, 7, 9])
def test_parametrized_das_fft_extension(scale):
    fs = ko.FFTSettings(scale)  # das_extension_test.go:42-86
    rng = np.random.default_rng(scale)
    even = ko.fr_from_ints([int(v) for v in rng.integers(0, 2**63, size=fs.max_width // 2)])
    odd = fs.das_fft_extension(even)
    data = np.empty((fs.max_width, 4), dtype=np.uint64)
    data[0::2], data[1::2] = even, odd
    coeffs = fs.fft(data, inv=True)
    assert not coeffs[fs.max_width // 2:].any()


def test_point_compression_kat():
    k = KATS["test_point_compression"]  # bls/bls_test.go:11-23
    p = ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([int(k["scalar"])])[0])
    assert list(ko.g1_compress(p)[0]) == k["expected_bytes"]
    q = ko.g1_decompress(np.array(k["expected_bytes"], dtype=np.uint8))[0]
    assert ko.g1_equal(p, q)
    assert pyref.g1_compress(pyref.g1_mul(pyref.G, int(k["scalar"]))) == bytes(k["expected_bytes"])


def test_decompress_rejects_points_outside_the_subgroup():
    # (4, sqrt(68)) is on y^2 = x^3 + 4 but has a cofactor-order component: [r]P != inf (checked with the big-int restatement);
    # Kilic's G1.FromCompressed returns "point is not on correct subgroup" for it
    x = 4
    y = pow((x ** 3 + 4) % pyref.P, (pyref.P + 1) // 4, pyref.P)
    assert y * y % pyref.P == (x ** 3 + 4) % pyref.P
    assert pyref.g1_add(pyref.g1_mul((x, y), pyref.R - 1), (x, y)) is not None      # not the identity -> outside G1
    for flag in (0x80, 0xA0):
        b = np.zeros((1, 48), dtype=np.uint8)
        b[0, 0], b[0, 47] = flag, 4
        with pytest.raises(ValueError):
            ko.g1_decompress(b)
    ko.g1_decompress(ko.g1_compress(ko.g1_generator()[None]))                        # members still pass


def test_empty_lincomb_is_zero():
    out = ko.lincomb_g1(ko.g1_empty(0), ko.fr_empty(0))  # bls/bls_test.go:69-78
    assert ko.g1_equal(out, ko.g1_zero()[0])
    assert hexpt(out) == "c0" + "00" * 47


def test_reverse_bit_order():
    # reverse_bit_order_test.go: reverseBitsLimited + permutation is an involution
    assert ko.reverse_bits_limited(32, 9) == 18  # fk20_single_test.go uses position 9 <-> index 18
    a = ko.fr_from_ints(range(32))
    b = ko.reverse_bit_order(a)
    assert ko.fr_to_ints(b) == pyref.bitrev(list(range(32)))
    assert np.array_equal(ko.reverse_bit_order(b), a)


# ---------------------------------------------------------------- eth/trusted_setup.json as a KAT
@pytest.fixture(scope="module")
def setup_1337():
    return ko.generate_testing_setup_g1(1337, 4096)


def test_trusted_setup_is_powers_of_1337(setup_1337):
    comp = ko.g1_compress(setup_1337).tobytes()
    assert hashlib.sha256(comp).hexdigest() == PINS["setup_G1"]
    assert comp == open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read()


def test_trusted_setup_roots_of_unity():
    fs = ko.FFTSettings(12)
    roots = fs.expanded_roots()[:4096]
    raw = b"".join(v.to_bytes(32, "little") for v in ko.fr_to_ints(roots))
    assert hashlib.sha256(raw).hexdigest() == PINS["roots_of_unity_le32"]


def test_trusted_setup_lagrange_is_ifft_g1(setup_1337):
    """setup_G1_lagrange == FFTG1(setup_G1, inv=true): a 4096 x 48 B known answer for fft_g1.go:58-94."""
    fs = ko.FFTSettings(12)
    lag = fs.fft_g1(setup_1337, inv=True)
    comp = ko.g1_compress(lag).tobytes()
    assert hashlib.sha256(comp).hexdigest() == PINS["setup_G1_lagrange"]
    assert comp == open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read()


# ---------------------------------------------------------------- derived vectors A-F
@pytest.fixture(scope="module")
def ks16():
    fs = ko.FFTSettings(4)
    return ko.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 17))


def test_vector_A_commit(ks16):
    assert hexpt(ks16.commit_to_poly(ko.fr_from_ints(TEST_POLY))) == DERIVED["A_commit_test_poly"]


def test_vector_B_proof_single(ks16):
    poly = ko.fr_from_ints(TEST_POLY)
    proof = ks16.compute_proof_single(poly, 17)
    assert hexpt(proof) == DERIVED["B_proof_single_x17"]
    assert pyref.eval_poly(TEST_POLY, 17) == int(DERIVED["B_y"])
    # pairing-free validity: proof == [(p(s) - y)/(s - x)]G1
    d = pyref.single_proof_dlog(TEST_POLY, S_TEST, 17)
    assert ko.g1_equal(proof, ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([d])[0]))
    assert ko.fr_to_ints(ko.poly_quotient_linear(poly, 17)) == pyref.quotient_linear(TEST_POLY, 17)


def test_commit_by_eval_equals_commit_by_coeffs(ks16):
    # kzg_single_proofs_test.go:11-31 -- the reference's only direct FFTG1 test
    fs = ks16.fs
    poly = ko.fr_from_ints(TEST_POLY)
    eval_poly = fs.fft(poly)
    secret_ifft = fs.fft_g1(ks16.secret_g1[:16], inv=True)
    assert ko.g1_equal(ko.lincomb_g1(secret_ifft, eval_poly), ks16.commit_to_poly(poly))


def test_vector_C_da_using_fk20():
    fs = ko.FFTSettings(5)  # fk20_single_test.go:11-44
    ks = ko.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 33))
    fk = ko.FK20SingleSettings(ks, 32)
    proofs = fk.da_using_fk20(ko.fr_from_ints(TEST_POLY))
    comp = ko.g1_compress(proofs)
    v = DERIVED["C_da_using_fk20_scale5"]
    assert hashlib.sha256(comp.tobytes()).hexdigest() == v["sha256"]
    for idx in ("0", "18", "31"):
        assert comp[int(idx)].tobytes().hex() == v[idx]
    # every position against the dlog identity and against ComputeProofSingle-equivalent dlog
    pfs = pyref.FFTSettings(5)
    dl = pyref.bitrev(pyref.fk20_single_da_dlogs(pfs, TEST_POLY, S_TEST))
    gen = ko.g1_generator()
    for i in range(32):
        x = pfs.expanded[pyref.rev_bits(i, 5)]
        assert dl[i] == pyref.single_proof_dlog(TEST_POLY, S_TEST, x)
        assert ko.g1_equal(proofs[i], ko.g1_mul(gen, ko.fr_from_ints([dl[i]])[0]))


def fk20_multi_test_poly(chunk_len=16, chunk_count=32):
    # fk20_multi_test.go:21-32
    poly = []
    for i in range(chunk_count):
        vals = [1, 2, 3, 4 + i, 7, 8 + i * i, 9, 10, 13, 14, 1, 15, 0, 1000, 0, 33]
        vals[12] = ko.R_MOD - 1
        vals[14] = ko.R_MOD - 134
        poly += vals
    return poly


def test_vectors_D_E_fk20_multi():
    chunk_len, chunk_count = 16, 32
    n = chunk_len * chunk_count
    fs = ko.FFTSettings(10)
    ks = ko.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 2 * n))
    poly_i = fk20_multi_test_poly()
    poly = ko.fr_from_ints(poly_i)
    assert hexpt(ks.commit_to_poly(poly)) == DERIVED["D_commit_fk20_multi_poly"]
    fk = ko.FK20MultiSettings(ks, 2 * n, chunk_len)
    proofs = fk.da_using_fk20_multi(poly)
    comp = ko.g1_compress(proofs)
    v = DERIVED["E_da_using_fk20_multi_scale10_l16"]
    assert hashlib.sha256(comp.tobytes()).hexdigest() == v["sha256"]
    assert comp[0].tobytes().hex() == v["0"] and comp[63].tobytes().hex() == v["63"]
    # coset-proof identity at a few positions (fk20_multi_test.go:60-90 does the pairing version)
    pfs = pyref.FFTSettings(10)
    gen = ko.g1_generator()
    for pos in (0, 1, 37, 63):
        x = pfs.expanded[pyref.rev_bits(pos, 6)]  # domainStride = MaxWidth / n2 = 1 (fk20_multi_test.go:58-63)
        d = pyref.coset_proof_dlog(poly_i, S_TEST, x, chunk_len)
        assert ko.g1_equal(proofs[pos], ko.g1_mul(gen, ko.fr_from_ints([d])[0]))


def test_vector_F_synthetic_blob(setup_1337):
    blob = ko.synthetic_blob(1)
    v = DERIVED["F_blob_seed1"]
    raw = b"".join(x.to_bytes(32, "little") for x in ko.fr_to_ints(blob))
    assert hashlib.sha256(raw).hexdigest() == v["blob_sha256_le32"]
    assert hexpt(ko.lincomb_g1(setup_1337, blob)) == v["commit_monomial_s1337"]
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    lag_br = ko.reverse_bit_order(lag)  # eth/globals.go:48
    assert hexpt(ko.lincomb_g1(lag_br, blob)) == v["commit_eth_bitrev_lagrange"]


# ---------------------------------------------------------------- C oracle vs independent big-int restatement
def test_c_oracle_matches_pyref_small():
    rng = np.random.default_rng(7)
    ints = [int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(16)]
    fs, pfs = ko.FFTSettings(6), pyref.FFTSettings(6)
    assert ko.fr_to_ints(fs.fft(ko.fr_from_ints(ints))) == pfs.fft(ints)
    assert ko.fr_to_ints(fs.fft(ko.fr_from_ints(ints[:11]))) == pfs.fft(ints[:11])  # padding path fft_fr.go:60-68
    # G1: FFT, MSM, add/sub edge cases
    ppts = [pyref.g1_mul(pyref.G, k) for k in ints[:8]]
    ppts[3] = None
    ppts[5] = ppts[4]
    ppts[6] = pyref.g1_neg(ppts[4])
    pts = np.stack([pt_from_affine_ints(a) for a in ppts])
    got = ko.g1_compress(fs.fft_g1(pts))
    want = pfs.fft_g1(ppts)
    assert [g.tobytes() for g in got] == [pyref.g1_compress(w) for w in want]
    got = ko.g1_compress(fs.fft_g1(pts, inv=True))
    want = pfs.fft_g1(ppts, inv=True)
    assert [g.tobytes() for g in got] == [pyref.g1_compress(w) for w in want]
    assert hexpt(ko.lincomb_g1(pts, ko.fr_from_ints(ints[:8]))) == pyref.g1_compress(pyref.lincomb(ppts, ints[:8])).hex()
    for a in range(8):
        for b in range(8):
            assert hexpt(ko.g1_add(pts[a], pts[b])) == pyref.g1_compress(pyref.g1_add(ppts[a], ppts[b])).hex()
            assert hexpt(ko.g1_sub(pts[a], pts[b])) == pyref.g1_compress(pyref.g1_add(ppts[a], pyref.g1_neg(ppts[b]))).hex()


def test_lincomb_window_sizes():
    # n < 32 -> c = 3, n >= 32 -> c = ceil(ln n); both against the naive sum
    rng = np.random.default_rng(11)
    gen = ko.g1_generator()
    for n in (1, 5, 31, 32, 100):
        ks = [int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)]
        ds = [int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)]
        pts = np.stack([ko.g1_mul(gen, ko.fr_from_ints([d])[0]) for d in ds])
        want = sum(k * d for k, d in zip(ks, ds)) % ko.R_MOD
        assert ko.g1_equal(ko.lincomb_g1(pts, ko.fr_from_ints(ks)), ko.g1_mul(gen, ko.fr_from_ints([want])[0]))


def test_error_codes():
    fs = ko.FFTSettings(4)
    with pytest.raises(ko.OracleError) as e:
        fs.fft(ko.fr_empty(17))  # fft_fr.go:57-59
    assert e.value.status == ko.ERR_TOO_WIDE
    with pytest.raises(ko.OracleError) as e:
        fs.inplace_fft(ko.fr_empty(12))  # fft_fr.go:81-83
    assert e.value.status == ko.ERR_NOT_POW2
    with pytest.raises(ko.OracleError) as e:
        fs.fft_g1(ko.g1_zero(12))  # fft_g1.go:63-65
    assert e.value.status == ko.ERR_NOT_POW2
    with pytest.raises(ko.OracleError) as e:
        fs.das_fft_extension(ko.fr_empty(16))  # das_extension.go:72-74
    assert e.value.status == ko.ERR_TOO_WIDE
    ks = ko.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 17))
    fk = ko.FK20SingleSettings(ks, 16)
    bad = ko.fr_from_ints([1] * 16)
    with pytest.raises(ko.OracleError) as e:
        fk.fk20_single_da_optimized(bad)  # fk20_single.go:150-154
    assert e.value.status == ko.ERR_UPPER_HALF


def test_check_proof_multi_scenario():
    """TestKZGSettings_CheckProofMulti (kzg_multi_proofs_test.go:12-51) with the pairing replaced by the dlog identity
    (p(s) - I(s)) == pi * (s^n - x^n): holds for the reference's quirky ComputeProofMulti because len(poly) <= 2n."""
    fs = ko.FFTSettings(4)
    ks = ko.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 17))
    poly = ko.fr_from_ints(TEST_POLY)
    x, n = 5431, 8
    proof = ks.compute_proof_multi(poly, x, n)
    assert ko.g1_equal(proof, ko.lincomb_g1(ks.secret_g1[:8], poly[8:]))          # divisor X^n: quotient = poly[n:]
    ks8 = ko.KZGSettings(ko.FFTSettings(3), ko.generate_testing_setup_g1(S_TEST, 9))
    w8 = pyref.root_of_unity(3)
    ys_i = [pyref.eval_poly(TEST_POLY, x * pow(w8, i, ko.R_MOD) % ko.R_MOD) for i in range(n)]
    is1, xpow = ks8.check_proof_multi_interpolation(ko.fr_from_ints(ys_i), ko.fr_from_ints([x])[0])
    assert ko.fr_to_ints(xpow.reshape(1, 4))[0] == pow(x, n, ko.R_MOD)
    R = ko.R_MOD
    xl = pow(x, n, R)
    rem = list(TEST_POLY)
    for i in range(15, n - 1, -1):
        rem[i - n] = (rem[i - n] + rem[i] * xl) % R
    i_s = pyref.eval_poly(rem[:n], S_TEST)
    gen = ko.g1_generator()
    assert ko.g1_equal(is1, ko.g1_mul(gen, ko.fr_from_ints([i_s])[0]))
    pi = pyref.eval_poly(TEST_POLY[8:], S_TEST)
    assert (pyref.eval_poly(TEST_POLY, S_TEST) - i_s) % R == pi * (pow(S_TEST, n, R) - xl) % R


# ---------------------------------------------------------------- erasure recovery
def test_zero_poly_python_kat():
    k = KATS["test_zero_poly_python"]                      # zero_poly_test.go:133-198
    fs = ko.FFTSettings(k["scale"])
    missing = [i for i, e in enumerate(k["exists"]) if not e]
    ze, zp = fs.zero_poly_via_multiplication(missing, 16)
    assert ko.fr_to_ints(ze) == [int(v) for v in k["expected_eval"]]
    assert ko.fr_to_ints(zp) == [int(v) for v in k["expected_poly"]]


@pytest.mark.parametrize("scale,seed", [(5, 0), (8, 1), (10, 2), (12, 3)])
def test_zero_poly_tree_matches_direct_product(scale, seed):
    # zero_poly_test.go:83-131 (tree reduction == direct product), here against big-int products
    fs = ko.FFTSettings(scale)
    n = 1 << scale
    rng = np.random.default_rng(seed)
    missing = sorted(rng.choice(n, size=n // 2, replace=False).tolist())
    ze, zp = fs.zero_poly_via_multiplication(missing, n)
    w = pyref.root_of_unity(scale)
    zp_i = ko.fr_to_ints(zp)
    assert zp_i[len(missing)] == 1 and not any(zp_i[len(missing) + 1:])
    ms = set(missing)
    ze_i = ko.fr_to_ints(ze)
    for k in list(range(0, n, max(1, n // 16))) + missing[:4]:
        x = pow(w, k, ko.R_MOD)
        assert ze_i[k] == pyref.eval_poly(zp_i, x)
        assert (ze_i[k] == 0) == (k in ms)


@pytest.mark.parametrize("scale", [2, 5, 10])
def test_recover_poly_from_samples(scale):
    # recover_from_samples_test.go:10-137: half the coefficients zero, drop up to half of the samples, recover
    fs = ko.FFTSettings(scale)
    n = 1 << scale
    rng = np.random.default_rng(scale)
    poly = np.concatenate([ko.fr_from_ints([int(v) for v in rng.integers(0, 2**63, size=n // 2)]), ko.fr_empty(n // 2)])
    data = fs.fft(poly)
    present = np.ones(n, dtype=np.uint8)
    present[rng.choice(n, size=n // 2, replace=False)] = 0
    if scale == 2:
        present[:] = [1, 0, 0, 1]                           # TestFFTSettings_RecoverPolyFromSamples_Simple
    rec = fs.recover_poly_from_samples(np.where(present[:, None].astype(bool), data, 0), present)
    assert np.array_equal(rec, data)
