"""Parity of the HIP path (through the C ABI, via the gokzg_amd binding) with the CPU oracle and with the
golden fixtures.  Every test here needs a real MI355X: run with `-m gpu`.  Integer work: bit-exact.

The structure follows the reference's tests: fft_fr_test.go, das_extension_test.go, bls/bls_test.go,
kzg_single_proofs_test.go, fk20_single_test.go, fk20_multi_test.go.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import koracle as ko
from oracle import pyref

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
DERIVED = json.load(open(os.path.join(GOLDEN, "derived_vectors.json")))
PINS = json.load(open(os.path.join(GOLDEN, "trusted_setup_sha256.json")))
S_TEST = int(KATS["test_secret"]["value"])
TEST_POLY = KATS["test_poly"]["values"]


@pytest.fixture(scope="module")
def kz():
    import gokzg_amd
    assert gokzg_amd.device_count() >= 1, "no gfx950 device: the HIP path is the only path"
    return gokzg_amd


def rand_fr(rng, n):
    return ko.fr_from_ints([int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)])


def comp_hex(pts):
    return [c.tobytes().hex() for c in ko.g1_compress(pts)]


def assert_points_equal(got, want):
    """bit-exact on the normalised images AND on the compressed bytes"""
    got, want = np.asarray(got).reshape(-1, 3, 6), np.asarray(want).reshape(-1, 3, 6)
    assert got.shape == want.shape
    wn = ko.g1_affine(want)
    bad = np.nonzero((got != wn).any(axis=(1, 2)))[0]
    assert bad.size == 0, "first mismatching indices: %s" % bad[:8]


# ------------------------------------------------------------------ F_r FFT (fft_fr_test.go)
FK20_PINS = json.load(open(os.path.join(GOLDEN, "fk20_pins.json")))    # oracle byte pins of the full-size FK20 configs (make_fk20_pins.py)


def proofs_sha256(fs, proofs):
    return hashlib.sha256(fs.to_compressed_g1(proofs).tobytes()).hexdigest()


def test_fft_settings_roots(kz):
    for scale in (4, 12):
        fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
        assert np.array_equal(fs.expanded_roots_of_unity(), ofs.expanded_roots())
        assert np.array_equal(fs.reverse_roots_of_unity(), ofs.reverse_roots())
        fs.close()


def test_inv_fft_kat(kz):
    k = KATS["test_inv_fft"]
    fs = kz.FFTSettings(k["scale"])
    res = fs.fft(ko.fr_from_ints(k["input"]), inv=True)
    assert ko.fr_to_ints(res) == [int(v) for v in k["expected"]]
    fs.close()


def test_fft_roundtrip(kz):
    fs = kz.FFTSettings(4)
    data = ko.fr_from_ints(range(16))
    coeffs = fs.fft(data)
    assert np.array_equal(fs.fft(coeffs, inv=True), data)
    fs.close()


@pytest.mark.parametrize("n", [1, 2, 4, 8, 64, 100, 1024, 4096, 5000, 8192, 65536])
def test_fft_fr_matches_oracle(kz, n):
    scale = 16
    fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
    rng = np.random.default_rng(n)
    vals = rand_fr(rng, n)
    for inv in (False, True):
        assert np.array_equal(fs.fft(vals, inv), ofs.fft(vals, inv)), (n, inv)
    fs.close()


@pytest.mark.parametrize("logn", [17, 20])
def test_fft_fr_above_65536(kz, logn):
    """sizes above 2^16 (bit-reversal copy, LDS-resident 4096-point tiles, one launch per remaining stage; fft.go:44-61 allows them, the reference's
    benchmark loop stops at scale 15): forward, inverse and the round trip at 2^17 and 2^20 points vs the C oracle, one and three rows, a zero-padded
    input (FFT pads to the next power of two, fft_fr.go:60), in a settings object of exactly that width and a wider one"""
    n = 1 << logn
    for max_scale in ((logn, logn + 1) if logn == 17 else (logn,)):
        fs, ofs = kz.FFTSettings(max_scale), ko.FFTSettings(max_scale)
        vals = ko.synthetic_blob(logn, n)
        vals[:4] = ko.fr_from_ints([0, ko.R_MOD - 1, 1, ko.R_MOD - 2])
        fwd = fs.fft(vals, False)
        assert np.array_equal(fwd, ofs.fft(vals, False)), (logn, max_scale)
        assert np.array_equal(fs.fft(fwd, True), vals)
        assert np.array_equal(fs.fft(vals, True), ofs.fft(vals, True)), (logn, max_scale)
        if logn == 17:
            short = vals[:n - 12345]
            assert np.array_equal(fs.fft(short, False), ofs.fft(short, False))
            rows = np.stack([vals, fwd, np.roll(vals, 7, axis=0)])
            for inv in (False, True):
                got = fs.fft_batch(rows, inv=inv)
                for b in range(3):
                    assert np.array_equal(got[b], ofs.fft(rows[b], inv)), (inv, b)
        fs.close()


def test_das_extension_and_recovery_at_scale_17(kz):
    """DASFFTExtension of 2^16 values in the exact-width scale-17 domain (one row and three) and RecoverPolyFromSamples / ZeroPolyViaMultiplication at
    2^17 points with half of the samples missing: the odd values and the vanishing polynomial against the C oracle, the recovered data against the
    transform of the known polynomial (recover_from_samples_test.go:62-137 at a size the reference's tests do not reach)"""
    fs, ofs = kz.FFTSettings(17), ko.FFTSettings(17)
    n = 1 << 17
    even = ko.synthetic_blob(171, n // 2)
    odd = fs.das_fft_extension(even)
    assert np.array_equal(odd, ofs.das_fft_extension(even.copy()))
    rows = np.stack([even, odd, np.roll(even, 3, axis=0)])
    got = fs.das_fft_extension_batch(rows.copy())
    for b in range(3):
        assert np.array_equal(got[b], ofs.das_fft_extension(rows[b].copy())), b
    data = np.empty((n, 4), dtype=np.uint64)
    data[0::2], data[1::2] = even, odd
    assert not fs.fft(data, inv=True)[n // 2:].any()                      # das_extension_test.go:59-77
    rng = np.random.default_rng(17)
    missing = sorted(rng.choice(n, size=n // 2, replace=False).tolist())
    ze, zp = fs.zero_poly_via_multiplication(missing, n)
    oze, ozp = ofs.zero_poly_via_multiplication(missing, n)
    assert np.array_equal(ze, oze) and np.array_equal(zp, ozp)
    present = np.ones(n, dtype=np.uint8)
    present[missing] = 0
    samples = np.where(present[:, None].astype(bool), data, 0)
    assert np.array_equal(fs.recover_poly_from_samples(samples, present), data)
    fs.close()


def test_settings_scale_limits(kz):
    """NewFFTSettings (fft.go:44-61): scales above KZG_HIP_MAX_SCALE = 24 are refused with their own status instead of attempting tables of tens of
    GB; above 31 the reference indexes past its root table (panic)"""
    h = C.c_void_p()
    L = kz.lib()
    assert L.kzg_hip_fft_settings_new(0, 25, C.byref(h)) == kz.ERR_UNSUPPORTED and not h.value
    assert L.kzg_hip_fft_settings_new(0, 31, C.byref(h)) == kz.ERR_UNSUPPORTED and not h.value
    assert L.kzg_hip_fft_settings_new(0, 32, C.byref(h)) == kz.ERR_BAD_ARG and not h.value


def test_fr_lazy_kernels_4096_and_das2048(kz):
    """the radix-4 kernels on lazy 29-bit limbs (k_fr_fft4096_r4, k_das_ext2048_r4): padding (n_in < n), both directions, batches with
    distinct rows, the exact-width domain and a wider one (the twiddle files stride through the settings' own tables), edge values"""
    rng = np.random.default_rng(4096)
    for scale in (12, 14):
        fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
        rows = np.stack([rand_fr(rng, 4096) for _ in range(5)])
        rows[0, :4] = ko.fr_from_ints([0, ko.R_MOD - 1, 1, ko.R_MOD - 2])
        rows[1] = ko.fr_from_ints([ko.R_MOD - 1] * 4096)
        rows[2] = 0
        for inv in (False, True):
            got = fs.fft_batch(rows, inv=inv)
            for b in range(5):
                assert np.array_equal(got[b], ofs.fft(rows[b], inv)), (scale, inv, b)
        d = rows[:, :2048].copy()
        gotd = fs.das_fft_extension_batch(d)
        for b in range(5):
            assert np.array_equal(gotd[b], ofs.das_fft_extension(rows[b, :2048].copy())), (scale, b)
        fs.close()


def test_fr_fft_below_4096_points_share_a_workgroup(kz):
    """4 .. 2048 points: 4096 / m transforms per workgroup through the first passes of the 4096-point network (k_fr_fft_small; odd log2 m adds a
    radix-2 pass): every size, batches that do not fill the last workgroup, distinct rows with edge values, zero padding, both directions, in a
    settings object of width 4096 and a wider one"""
    rng = np.random.default_rng(2048)
    for max_scale in (12, 14):
        fs, ofs = kz.FFTSettings(max_scale), ko.FFTSettings(max_scale)
        for logm in range(2, 12):
            m = 1 << logm
            per = 4096 // m
            for batch in sorted({1, max(per - 1, 1), per, per + 1, 2 * per + 3}):
                rows = rand_fr(rng, batch * m).reshape(batch, m, 4)
                rows[0, :2] = ko.fr_from_ints([ko.R_MOD - 1, 0])
                rows[-1, m // 2:] = ko.fr_from_ints([ko.R_MOD - 1])[0]
                for inv in (False, True):
                    got = fs.fft_batch(rows, inv=inv)
                    for b_ in sorted({0, batch // 2, batch - 1}):
                        assert np.array_equal(got[b_], ofs.fft(rows[b_], inv)), (max_scale, m, batch, inv, b_)
            short = rand_fr(rng, m // 2 + 1)                                  # zero-padded up to m (fft_fr.go:60-68)
            assert np.array_equal(fs.fft(short), ofs.fft(short)), (max_scale, m)
        fs.close()


def test_fr_fft_above_4096_points(kz):
    """8192 .. 65 536 points: rows through the 4096-point LDS kernel (every R-th element), upper stages in registers (k_fr_fft_upper): each size
    in a settings object of exactly its width and in a wider one (twiddle strides), batches with distinct rows, edge values (0, r - 1), zero
    padding (FFT pads to the next power of two, fft_fr.go:60-68), both directions; 131 072 points still take the radix-2 stages"""
    rng = np.random.default_rng(8192)
    for max_scale, logn in ((13, 13), (16, 13), (14, 14), (15, 15), (16, 16), (17, 16), (17, 17)):
        n = 1 << logn
        fs, ofs = kz.FFTSettings(max_scale), ko.FFTSettings(max_scale)
        rows = np.stack([rand_fr(rng, n) for _ in range(3)])
        rows[0, :4] = ko.fr_from_ints([0, ko.R_MOD - 1, 1, ko.R_MOD - 2])
        rows[1, n // 2:] = ko.fr_from_ints([ko.R_MOD - 1])[0]
        for inv in (False, True):
            got = fs.fft_batch(rows, inv=inv)
            for b in range(3):
                assert np.array_equal(got[b], ofs.fft(rows[b], inv)), (max_scale, logn, inv, b)
        short = rows[2, : n // 2 + 37]                                      # padded with zeros up to n
        assert np.array_equal(fs.fft(short), ofs.fft(short)), (max_scale, logn)
        assert np.array_equal(fs.fft(fs.fft(rows[2]), inv=True), rows[2])
        fs.close()


def test_fr_shared_workgroup_kernel_in_a_fresh_process():
    """below 2^20 values per launch short transforms keep one workgroup each (their small workgroups spread over the chip); the shared-workgroup
    kernel is forced at every batch size in a child process (KZG_HIP_FR_FFT=shared) for the tests that walk all sizes and ragged batches, and run
    here on full-size launches"""
    import subprocess
    import sys
    if os.environ.get("KZG_HIP_FR_FFT"):
        pytest.skip("already a forced child")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                          "fft_fr or fr_fft_below or inv_fft or roundtrip or zero_poly or recover or full_das_flow or evaluation_form"],
                         env=dict(os.environ, KZG_HIP_FR_FFT="shared"), capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-1500:]
    import gokzg_amd as kz_
    fs, ofs = kz_.FFTSettings(12), ko.FFTSettings(12)
    rng = np.random.default_rng(31)
    for m in (16, 512, 2048):
        batch = (1 << 20) // m + 3                                           # above the threshold, last workgroup partly filled
        rows = rand_fr(rng, batch * m).reshape(batch, m, 4)
        for inv in (False, True):
            got = fs.fft_batch(rows, inv=inv)
            for b_ in (0, 1, batch // 2, batch - 2, batch - 1):
                assert np.array_equal(got[b_], ofs.fft(rows[b_], inv)), (m, inv, b_)
    fs.close()


def test_fr_radix2_kernels_in_a_fresh_process():
    """sizes other than 4096 / 2048 (and tiles of longer transforms) still run the radix-2 kernels; at the hot sizes they are re-run in a
    child process that forces them (KZG_HIP_FR_FFT=radix2), so both families stay pinned to the oracle and the reference's KATs"""
    import subprocess
    import sys
    if os.environ.get("KZG_HIP_FR_FFT") == "radix2":
        pytest.skip("already the forced child")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                          "fft_fr or das or fr_lazy or fr_fft_above or fr_fft_below or vector_C or full_das_flow"],
                         env=dict(os.environ, KZG_HIP_FR_FFT="radix2"), capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-1500:]


def test_cooperative_fp_inversion_matches_the_lane_form(kz):
    """wave_inv_fp (coop_inv.hpp: one wavefront per inversion, limbs across lanes, safegcd divsteps on the scalar unit) against inv<FpP>() on one lane, on the device:
    word for word equal on edge values and 4096 random elements, and equal to Python's pow(x, -1, p) in the device's Montgomery domain (R' = 2^390)"""
    import ctypes as C
    P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    R = 1 << 390
    rng = np.random.default_rng(606)
    xs = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, 1 << 380, 5, 1 << 200, (1 << 30) - 1, 1 << 30, (1 << 360) + 1]
    xs += [int.from_bytes(rng.bytes(48), "little") % P for _ in range(4096 - len(xs))]
    img = np.frombuffer(b"".join((x * R % P).to_bytes(48, "little") for x in xs), dtype=np.uint8).copy()
    a, b = np.zeros_like(img), np.zeros_like(img)
    fs = kz.FFTSettings(4)
    tc, tl = C.c_double(0), C.c_double(0)
    st = kz.lib().kzg_hip_test_fp_inv(fs.h, img.ctypes.data, len(xs), a.ctypes.data, b.ctypes.data, C.byref(tc), C.byref(tl))
    assert st == 0, kz.lib().kzg_hip_last_error()
    assert np.array_equal(a, b)
    got = [int.from_bytes(a[48 * i:48 * i + 48].tobytes(), "little") for i in range(len(xs))]
    for x, y in zip(xs[:64], got[:64]):
        assert y == (pow(x, -1, P) * R % P if x else 0), hex(x)
    # one element: the latency of ONE inversion either way (the launch's HIP-event time)
    st = kz.lib().kzg_hip_test_fp_inv(fs.h, img[48 * 20:].ctypes.data, 1, a.ctypes.data, b.ctypes.data, C.byref(tc), C.byref(tl))
    assert st == 0 and np.array_equal(a[:48], b[:48])
    print("one F_p inversion alone on the chip: wave-cooperative %.1f us, one lane %.1f us" % (tc.value * 1e3, tl.value * 1e3))
    assert tc.value < tl.value
    fs.close()


def test_cooperative_fr_inversion_and_workgroup_batch_inversion(kz):
    """the F_r instance of the cooperative inversion (9 limbs) and the workgroup batch inversion built on it (block_batch_inverse: a product tree in LDS, ONE inversion per
    1024 values -- what eth.ComputeKZGProof's quotient kernel runs instead of a binary GCD per lane): word for word the lane form and Python's pow(x, -1, r)"""
    r, Rr = ko.R_MOD, 1 << 256
    rng = np.random.default_rng(607)
    xs = [1, 2, 3, r - 1, r - 2, (r + 1) // 2, 1 << 254, 5, (1 << 30) - 1, 1 << 30, (1 << 240) + 1]
    xs += [int.from_bytes(rng.bytes(32), "little") % (r - 1) + 1 for _ in range(3000 - len(xs))]     # 3000: two full workgroups and a ragged third
    img = np.frombuffer(b"".join((x * Rr % r).to_bytes(32, "little") for x in xs), dtype=np.uint8).copy()
    a, b, c = np.zeros_like(img), np.zeros_like(img), np.zeros_like(img)
    fs = kz.FFTSettings(4)
    st = kz.lib().kzg_hip_test_fr_inv(fs.h, img.ctypes.data, len(xs), a.ctypes.data, b.ctypes.data, c.ctypes.data)
    assert st == 0, kz.lib().kzg_hip_last_error()
    assert np.array_equal(a, b) and np.array_equal(c, b)
    for i in list(range(16)) + [1023, 1024, 2047, 2048, 2999]:
        assert int.from_bytes(c[32 * i:32 * i + 32].tobytes(), "little") == pow(xs[i], -1, r) * Rr % r, i
    fs.close()


def test_fr_fft4096_r16_ab_artefact_is_bit_exact():
    """the 256-lane x 16-register form of the 4096-point transform left the library in round 6 (25 % slower: profiles/r05_fr_fft_ab.md, r06_fr_r16_fate.md); its kernel
    lives on as a stand-alone A/B harness (tools/ab_fr_r16) that sends the same rows through it and through the library's kernel: still word for word the same"""
    import subprocess
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ab_fr_r16")
    exe = os.path.join(d, "r16_ab")
    if not os.path.exists(exe):
        subprocess.check_call([os.path.join(d, "build.sh")])
    res = subprocess.run([exe, "64", "2"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "bit-exact: yes" in res.stdout, res.stdout + res.stderr


def test_fft_fr_batch_and_config1_roundtrip(kz):
    # BASELINE config 1: FFT_Fr scale 12 forward + inverse round trip on blob(seed 12)
    fs, ofs = kz.FFTSettings(12), ko.FFTSettings(12)
    blobs = np.stack([ko.synthetic_blob(12 + b) for b in range(5)])
    f = fs.fft_batch(blobs)
    assert np.array_equal(f[0], ofs.fft(blobs[0]))
    assert np.array_equal(f[4], ofs.fft(blobs[4]))
    assert np.array_equal(fs.fft_batch(f, inv=True), blobs)
    fs.close()


def test_fft_errors(kz):
    fs = kz.FFTSettings(4)
    with pytest.raises(kz.KzgError) as e:
        fs.fft(ko.fr_empty(17))
    assert e.value.status == kz.ERR_TOO_WIDE
    with pytest.raises(kz.KzgError) as e:
        fs.inplace_fft(ko.fr_empty(12))
    assert e.value.status == kz.ERR_NOT_POW2
    with pytest.raises(kz.KzgError) as e:
        fs.fft_g1(ko.g1_zero(12))
    assert e.value.status == kz.ERR_NOT_POW2
    with pytest.raises(kz.KzgError) as e:
        fs.fft_g1(ko.g1_zero(32))
    assert e.value.status == kz.ERR_TOO_WIDE
    with pytest.raises(kz.KzgPanic) as e:
        fs.das_fft_extension(ko.fr_empty(16))
    assert e.value.status == kz.ERR_TOO_WIDE
    fs.close()


# ------------------------------------------------------------------ DAS extension (das_extension_test.go)
def test_das_fft_extension_kat(kz):
    k = KATS["test_das_fft_extension"]
    fs = kz.FFTSettings(k["scale"])
    res = fs.das_fft_extension(ko.fr_from_ints(k["input"]))
    assert ko.fr_to_ints(res) == [int(v) for v in k["expected"]]
    fs.close()


@pytest.mark.parametrize("scale", [2, 4, 5, 9, 12, 13, 14, 15])
def test_parametrized_das_fft_extension(kz, scale):
    fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
    rng = np.random.default_rng(scale)
    even = rand_fr(rng, fs.max_width // 2)
    odd = fs.das_fft_extension(even)
    assert np.array_equal(odd, ofs.das_fft_extension(even))
    data = np.empty((fs.max_width, 4), dtype=np.uint64)
    data[0::2], data[1::2] = even, odd
    coeffs = fs.fft(data, inv=True)       # das_extension_test.go:59-77: upper half of the coefficients is zero
    assert not coeffs[fs.max_width // 2:].any()
    fs.close()


def test_das_extension_long_rows_batches_and_widths(kz):
    """rows of 4096 / 8192 values: in a settings object of exactly twice the row the extension runs as inverse transform, coefficient shift, forward
    transform; in a wider one the reference's recursion (which walks the full-width tables, das_extension.go:38,59) stage by stage -- both against
    the oracle's restatement of the recursion, batches with distinct rows and edge values"""
    rng = np.random.default_rng(77)
    for max_scale, n in ((13, 4096), (14, 4096), (14, 8192), (15, 8192)):
        fs, ofs = kz.FFTSettings(max_scale), ko.FFTSettings(max_scale)
        rows = np.stack([rand_fr(rng, n) for _ in range(3)])
        rows[0, :3] = ko.fr_from_ints([0, ko.R_MOD - 1, 1])
        rows[1] = ko.fr_from_ints([ko.R_MOD - 1])[0]
        got = fs.das_fft_extension_batch(rows.copy())
        for b in range(3):
            assert np.array_equal(got[b], ofs.das_fft_extension(rows[b].copy())), (max_scale, n, b)
        fs.close()


def test_das_extension_short_rows_in_full_launches(kz):
    """launches of 2^20 values in rows of at most 64 (longer rows stay on the recursion kernels), exact-width settings: the extension as inverse transform, shift, forward transform on the
    shared-workgroup kernel; sampled rows against the oracle's recursion"""
    rng = np.random.default_rng(78)
    for scale in (3, 6, 7, 10):
        n = 1 << (scale - 1)
        fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
        batch = (1 << 20) // n + 5
        rows = rand_fr(rng, batch * n).reshape(batch, n, 4)
        rows[0, :2] = ko.fr_from_ints([ko.R_MOD - 1, 0])
        got = fs.das_fft_extension_batch(rows.copy())
        for b_ in (0, 1, batch // 3, batch - 1):
            assert np.array_equal(got[b_], ofs.das_fft_extension(rows[b_].copy())), (scale, b_)
        fs.close()


def test_das_smaller_than_domain(kz):
    # the reference walks the full-width tables whatever the input length (das_extension.go:38,59)
    fs, ofs = kz.FFTSettings(8), ko.FFTSettings(8)
    even = rand_fr(np.random.default_rng(5), 16)
    assert np.array_equal(fs.das_fft_extension(even), ofs.das_fft_extension(even))
    fs.close()


# ------------------------------------------------------------------ G1 primitives (bls/bls_test.go)
@pytest.fixture(scope="module")
def fs16(kz):
    fs = kz.FFTSettings(16)
    yield fs
    fs.close()


def edge_points():
    gen = ko.g1_generator()
    rng = np.random.default_rng(99)
    pts = [ko.g1_mul(gen, k) for k in rand_fr(rng, 5)]
    pts += [ko.g1_zero()[0], gen, ko.g1_affine(pts[0])[0], ko.g1_sub(ko.g1_zero()[0], pts[0]), pts[1].copy()]
    return np.stack(pts)


def test_point_compression_kat(kz, fs16):
    k = KATS["test_point_compression"]
    x = ko.fr_from_ints([int(k["scalar"])])
    pt = fs16.mul_g1_vec(ko.g1_generator()[None], x)
    assert list(fs16.to_compressed_g1(pt)[0]) == k["expected_bytes"]
    back = fs16.from_compressed_g1(np.array(k["expected_bytes"], dtype=np.uint8))
    assert_points_equal(back, pt)


def test_compress_decompress_edge_cases(kz, fs16):
    pts = edge_points()
    comp = fs16.to_compressed_g1(pts)
    assert np.array_equal(comp, ko.g1_compress(pts))
    assert_points_equal(fs16.from_compressed_g1(comp), pts)
    bad = comp.copy()
    bad[0, 0] &= 0x7F                         # not flagged as compressed
    with pytest.raises(kz.KzgPanic) as e:
        fs16.from_compressed_g1(bad)
    assert e.value.status == kz.ERR_BAD_POINT
    notoncurve = np.zeros((1, 48), dtype=np.uint8)
    notoncurve[0, 0] = 0x80
    notoncurve[0, 47] = 0x01                  # x = 1: 1 + 4 = 5 is not a square mod p? checked against the oracle
    try:
        ko.g1_decompress(notoncurve)
        ok = True
    except ValueError:
        ok = False
    if not ok:
        with pytest.raises(kz.KzgPanic):
            fs16.from_compressed_g1(notoncurve)
    # on the curve but outside the order-r subgroup (x = 4; cofactor-order component): Kilic's FromCompressed rejects it
    # ("point is not on correct subgroup"), so do the oracle and the device -- for both roots
    for flag in (0x80, 0xA0):
        outside = np.zeros((1, 48), dtype=np.uint8)
        outside[0, 0] = flag
        outside[0, 47] = 0x04
        with pytest.raises(ValueError):
            ko.g1_decompress(outside)
        with pytest.raises(kz.KzgPanic) as e:
            fs16.from_compressed_g1(outside)
        assert e.value.status == kz.ERR_BAD_POINT
        mixed = np.concatenate([comp[:3], outside, comp[3:]])            # one bad point poisons the slice, as UnmarshalText's error does
        with pytest.raises(kz.KzgPanic):
            fs16.from_compressed_g1(mixed)


def test_fr_from_32_to_32(kz, fs16):
    # bls.FrFrom32 / FrTo32 (bls/bignum_kilic.go:33-55) over a slice, incl. the range rule (TestValidFr, bls/bignum_test.go:91)
    vals = [0, 1, 5, ko.R_MOD - 1, 2**255 % ko.R_MOD, 123456789 << 200]
    raw = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
    imgs, ok = fs16.fr_from_32(raw)
    assert ok and np.array_equal(imgs, ko.fr_from_ints(vals))
    assert np.array_equal(fs16.fr_to_32(imgs), raw)
    bad = raw.copy()
    bad[2] = np.frombuffer(ko.R_MOD.to_bytes(32, "little"), dtype=np.uint8)
    _, ok = fs16.fr_from_32(bad)
    assert not ok


def test_g1_text_marshalling_roundtrip(kz, fs16):
    # TestPointG1Marshalling (bls/bls_test.go:25-45) over a slice, plus the JSON-setup form (eth/globals.go:33-49)
    pts = edge_points()
    texts = fs16.g1_marshal_text(pts)
    assert texts == comp_hex(pts)
    assert_points_equal(fs16.g1_unmarshal_text(texts), pts)
    raw = open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read()
    hexes = [raw[48 * i:48 * i + 48].hex() for i in range(8)]          # first entries of "setup_G1"
    assert_points_equal(fs16.g1_unmarshal_text(hexes), ko.generate_testing_setup_g1(1337, 8))


def test_mul_g1_vec_matches_oracle(kz, fs16):
    pts = edge_points()
    rng = np.random.default_rng(4)
    scalars = np.concatenate([rand_fr(rng, 5), ko.fr_from_ints([0, 1, 2, ko.R_MOD - 1, 16])])
    want = np.stack([ko.g1_mul(p, k) for p, k in zip(pts, scalars)])
    assert_points_equal(fs16.mul_g1_vec(pts, scalars), want)


def test_empty_lincomb(kz, fs16):
    out = fs16.lin_comb_g1(ko.g1_empty(0), ko.fr_empty(0))
    assert ko.g1_equal(out, ko.g1_zero()[0])
    assert np.array_equal(out, ko.g1_zero()[0])


@pytest.mark.parametrize("n", [1, 2, 5, 31, 32, 100, 1000])
def test_lincomb_matches_oracle(kz, fs16, n):
    rng = np.random.default_rng(n)
    gen = ko.g1_generator()
    pts = np.stack([ko.g1_mul(gen, k) for k in rand_fr(rng, n)])
    if n >= 5:
        pts[1] = ko.g1_zero()[0]
        pts[2] = pts[3]                                  # duplicate points: P + P inside a bucket
        pts[4] = ko.g1_sub(ko.g1_zero()[0], pts[3])      # and P + (-P)
    scalars = rand_fr(rng, n)
    if n >= 5:
        scalars[0] = ko.fr_from_ints([0])[0]
        scalars[2] = scalars[3]
        scalars[4] = scalars[3]
    assert_points_equal(fs16.lin_comb_g1(pts, scalars), ko.lincomb_g1(pts, scalars))
    with pytest.raises(kz.KzgPanic):
        fs16.lin_comb_g1(pts, ko.fr_empty(n + 1))


def test_lincomb_promotes_a_repeated_point_set_and_notices_a_changed_point(kz, setup_1337):
    """bls.LinCombG1(setup, coeffs) as the reference's callers use it (eth/helpers.go:99,159,199: the same slice every time): from the call after
    KZG_HIP_LINCOMB_PROMOTE_AFTER + 1 sightings on, kzg_hip_lincomb_g1 serves the set from a cached table walk -- same bytes as the one-shot bucket pipeline and as the
    oracle.  Identity is a byte-for-byte comparison on every call: one changed coordinate (in place, same pointer, same fingerprint window or not) takes the one-shot
    path with the NEW points; the old set stays promoted for callers that still hold it.  Concurrent callers of one set get one table."""
    import threading
    fs = kz.FFTSettings(4)
    n = 512
    pts = setup_1337[:n].copy()
    rng = np.random.default_rng(99)
    sc = [rand_fr(rng, n) for _ in range(8)]
    want = [ko.g1_affine(ko.lincomb_g1(pts, x))[0] for x in sc[:3]]
    assert fs.lincomb_promotions() == (0, 0)
    got = [fs.lin_comb_g1(pts, sc[0]), fs.lin_comb_g1(pts, sc[1])]          # sightings 1 and 2: one-shot
    assert fs.lincomb_promotions() == (0, 0)
    got.append(fs.lin_comb_g1(pts, sc[2]))                                  # sighting 3: promoted, served by the set
    assert fs.lincomb_promotions() == (1, 0) or fs.lincomb_promotions() == (1, 1)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)
    before = fs.lincomb_promotions()[1]
    for x in sc[3:6]:
        assert np.array_equal(fs.lin_comb_g1(pts, x), ko.g1_affine(ko.lincomb_g1(pts, x))[0])
    assert fs.lincomb_promotions() == (1, before + 3)
    # one point in the MIDDLE changes in place (outside the fingerprint's first / last 4 KiB): the comparison sees it, the result is that of the new points
    changed = pts
    changed[n // 2] = setup_1337[n + 7]
    served = fs.lincomb_promotions()[1]
    assert np.array_equal(fs.lin_comb_g1(changed, sc[6]), ko.g1_affine(ko.lincomb_g1(changed, sc[6]))[0])
    assert fs.lincomb_promotions() == (1, served)                           # not served by the stale set
    # ... and a change inside the first 4 KiB (a different fingerprint)
    changed[1] = setup_1337[n + 9]
    assert np.array_equal(fs.lin_comb_g1(changed, sc[7]), ko.g1_affine(ko.lincomb_g1(changed, sc[7]))[0])
    # a caller that still holds the original points is still served
    orig = setup_1337[:n].copy()
    assert np.array_equal(fs.lin_comb_g1(orig, sc[0]), want[0]) and fs.lincomb_promotions()[1] == served + 1
    # ragged use of a promoted pointer: a shorter prefix is a different set (n is part of the identity), correct either way
    assert np.array_equal(fs.lin_comb_g1(orig[:100], sc[1][:100]), ko.g1_affine(ko.lincomb_g1(orig[:100], sc[1][:100]))[0])
    # eight threads, one new set: one promotion, every result right
    pts2 = setup_1337[1000:1000 + n].copy()
    outs, errs = [None] * 8, []

    def worker(t):
        try:
            for rep in range(4):
                outs[t] = fs.lin_comb_g1(pts2, sc[t])
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    for t in range(8):
        assert np.array_equal(outs[t], ko.g1_affine(ko.lincomb_g1(pts2, sc[t]))[0]), t
    assert fs.lincomb_promotions()[0] == 2
    fs.close()


LAMBDA = 0xac45a4010001a40200000000ffffffff


def test_lincomb_edge_scalars_and_cached_points(kz, fs16, setup_1337):
    """bls.LinCombG1 through the GLV bucket pipeline: scalars around the split boundaries (0, +-1, lambda, r/2, digits 0x80 / 0x7f
    that carry through every window) and a cached point set (kzg_hip_points_*: the same points, 2^64 multiples resident) --
    every result against the oracle's Kilic-style MultiExp, ragged lengths, a batch, infinity and duplicate points included"""
    r = ko.R_MOD
    edge = [0, 1, 2, r - 1, r - 2, LAMBDA, LAMBDA - 1, LAMBDA + 1, LAMBDA // 2, LAMBDA // 2 + 1, (r - 1) // 2, (r + 1) // 2, r - LAMBDA,
            int("80" * 31, 16), int("7f" * 31, 16), int("ff" * 31, 16), int("0180" * 15, 16), (1 << 254) % r, (1 << 128) - 1, 1 << 64, (1 << 64) - 1]
    n = 300
    rng = np.random.default_rng(300)
    pts = setup_1337[:n].copy()
    pts[5] = ko.g1_zero()[0]
    pts[7] = pts[6]
    pts[9] = ko.g1_sub(ko.g1_zero()[0], pts[8])
    scal = rand_fr(rng, n)
    scal[:len(edge)] = ko.fr_from_ints(edge)
    scal[8] = scal[9]                                       # k P + k (-P): a bucket that sums to infinity
    want = ko.lincomb_g1(pts, scal)
    assert_points_equal(fs16.lin_comb_g1(pts, scal), want)
    cached = kz.G1Points(fs16, pts)
    assert_points_equal(cached.lin_comb(scal), want)
    for m in (1, 2, 21, 299):                               # LinCombG1(points[:m], scalars[:m])
        assert_points_equal(cached.lin_comb(scal[:m]), ko.lincomb_g1(pts[:m], scal[:m]))
    batch = np.stack([scal, np.roll(scal, 3, axis=0), rand_fr(rng, n)])
    got = cached.lin_comb_batch(batch)
    for b in range(3):
        assert_points_equal(got[b], ko.lincomb_g1(pts, batch[b]))
    assert np.array_equal(cached.lin_comb(ko.fr_empty(0)), ko.g1_zero()[0])
    with pytest.raises(kz.KzgPanic):
        cached.lin_comb(rand_fr(rng, n + 1))
    # a cached set walks its own fixed-base table by default; budget 0 sends the same set through the bucket pipeline: identical bytes
    cached.set_table_budget_gb(0)
    assert_points_equal(cached.lin_comb(scal), want)
    assert np.array_equal(cached.lin_comb_batch(batch), got)
    for m in (1, 21, 299):
        assert_points_equal(cached.lin_comb(scal[:m]), ko.lincomb_g1(pts[:m], scal[:m]))
    cached.close()
    # full size: CommitToEvalPoly's use (kzg_single_proofs.go:12-14) with the Lagrange setup as the cached set == vector F (eth form)
    lag = ko.reverse_bit_order(ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8)))
    fs12 = kz.FFTSettings(12)
    lag_set = kz.G1Points(fs12, lag)
    blobs = np.stack([ko.synthetic_blob(1 + b) for b in range(4)])
    got = lag_set.lin_comb_batch(blobs)
    assert comp_hex(got[:1])[0] == DERIVED["F_blob_seed1"]["commit_eth_bitrev_lagrange"]
    assert_points_equal(got[3], ko.lincomb_g1(lag, blobs[3]))
    assert np.array_equal(lag_set.lin_comb(blobs[2]), got[2])
    lag_set.set_table_budget_gb(0)                            # bucket pipeline at full size
    assert np.array_equal(lag_set.lin_comb_batch(blobs), got)
    lag_set.set_table_budget_gb(8)                            # and a smaller table than the default
    assert np.array_equal(lag_set.lin_comb_batch(blobs), got)
    # bls.LinCombG1 is ONE linear combination per call: 24 concurrent callers (ragged lengths among them) share batched bucket MSMs
    import threading
    lens = [4096 if i % 3 else 4096 - 7 * i - 1 for i in range(24)]
    want = [lag_set.lin_comb_batch(blobs[i % 4][None, :lens[i]])[0] for i in range(24)]
    res, errs = [None] * 24, []

    def work(i):
        try:
            for _ in range(3):
                res[i] = lag_set.lin_comb(blobs[i % 4][:lens[i]])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    for i in range(24):
        assert np.array_equal(res[i], want[i]), i
    assert_points_equal(res[1], ko.lincomb_g1(lag[:lens[1]], blobs[1][:lens[1]]))
    assert_points_equal(res[3], ko.lincomb_g1(lag[:lens[3]], blobs[3][:lens[3]]))
    lag_set.close(); fs12.close()


def _bucket_edge_case(setup_1337, B, seed=64):
    """caller-type points with an infinity, P == Q and P == -Q among them, and B scalar rows built to stress the segment / chunk logic of the bucket
    pipeline: one scalar for every point (a window's entries all fall into ONE bucket), halves with a single non-zero window, the zero row, ones,
    r - 1 (negated halves), a short list (most buckets of a 4-bucket chunk empty: S_c == T_c takes the doubling branch of the merge), every other point"""
    pts = setup_1337.copy()
    pts[17] = ko.g1_zero()[0]                                  # an infinity among the points
    pts[19] = pts[18]                                          # P == Q inside a bucket whenever their digits agree
    pts[21] = ko.g1_sub(ko.g1_zero()[0], pts[20])              # ... and P == -Q
    rng = np.random.default_rng(seed)
    r = ko.R_MOD
    rows = np.stack([rand_fr(rng, 4096) for _ in range(6)])
    rows = np.concatenate([rows] * (B // 6 + 1))[:B].copy()
    same = ko.fr_from_ints([0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef % r])[0]
    rows[3, :] = same                                          # every point with the same scalar: 32 buckets of 4096 entries each
    rows[4, :] = 0                                             # no entries at all
    rows[5, :] = ko.fr_from_ints([1])[0]                       # one bucket in one window, plain half only
    rows[6, :] = ko.fr_from_ints([r - 1])[0]
    rows[7, :] = ko.fr_from_ints([LAMBDA])[0]                  # only the phi half
    rows[8, 100:] = 0                                          # a short list: most segments are empty
    rows[9, :] = ko.fr_from_ints([(LAMBDA << 8) % r])[0]
    rows[10, ::2] = same
    rows[11, 3:] = 0                                           # three entries per window: lone buckets inside otherwise empty chunks
    return pts, rows


def test_bucket_pipeline_balanced_accumulate(kz, setup_1337):
    """Batches that fill the GPU walk the sorted entries in segments of 64 (k_msm_accumulate_seg + k_msm_merge_segs; phi applied once per bucket
    half): 80 linear combinations over 4096 caller-type points on the bucket pipeline (table budget 0) against the fixed-base walk of the same
    points, with the edge rows of _bucket_edge_case, plus oracle rows"""
    fs = kz.FFTSettings(12)
    pts, rows = _bucket_edge_case(setup_1337, 80)
    cached = kz.G1Points(fs, pts)
    want = cached.lin_comb_batch(rows)                         # the set's own fixed-base table
    cached.set_table_budget_gb(0)
    got = cached.lin_comb_batch(rows)                          # bucket pipeline, 80 x 2048 segments: the balanced form
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=(1, 2)))[0][:8]
    for b in (0, 3, 5, 7, 10):
        assert_points_equal(got[b], ko.lincomb_g1(pts, rows[b]))
    assert np.array_equal(got[4], ko.g1_zero()[0])
    assert np.array_equal(cached.lin_comb_batch(rows[:8]), want[:8])   # a small batch takes the lane-per-bucket form: same bytes
    cached.close(); fs.close()


def test_bucket_pipeline_reduce_chunks(kz, setup_1337):
    """The throughput form of the bucket reduce (k_msm_reduce_chunks: a lane owns 4 buckets, double running sum, suffix scan over 32 lanes) is taken
    from 256 MSMs on a cached set (8 window groups) and from 128 on the 16-group layout of one-shot points / a settings object without a table:
    both thresholds crossed here, edge rows included, against the fixed-base walk of the same points (all rows) and the oracle (edge rows)"""
    fs = kz.FFTSettings(12)
    pts, rows = _bucket_edge_case(setup_1337, 256, seed=65)
    cached = kz.G1Points(fs, pts)
    want = cached.lin_comb_batch(rows)                         # the set's own fixed-base table (k_fb_accumulate)
    cached.set_table_budget_gb(0)
    got = cached.lin_comb_batch(rows)                          # 256 x 8 groups x 32 lanes = one round of SIMD lanes: reduce_chunks
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=(1, 2)))[0][:8]
    for b in (1, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        assert_points_equal(got[b], ko.lincomb_g1(pts, rows[b]))
    cached.close()
    ks = kz.KZGSettings(fs, pts)                               # CommitToPoly's no-table fallback: 16 window groups, no 2^64 rows
    ks.set_table_budget_gb(0)
    got2 = ks.commit_to_poly_batch(rows[:130])
    assert np.array_equal(got2, want[:130]), np.nonzero((got2 != want[:130]).any(axis=(1, 2)))[0][:8]
    ks.close(); fs.close()


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 16, 17, 48, 65])
def test_bucket_pipeline_combine_inversion_dispatch(kz, setup_1337, batch):
    """k_msm_combine normalises one result per blob; a wavefront holds 16 blobs: with up to three live ones it inverts them one after the other with the
    wave-cooperative form, with more in its lanes side by side (wave_inv_any, coop_inv.hpp).  Batch sizes on both sides of the switch, with rows whose sum is the point at
    infinity (no operand for the inversion) mixed in, on the bucket pipeline of a cached set -- against the set's table walk and the oracle"""
    fs = kz.FFTSettings(12)
    n = 256
    pts = setup_1337[:n].copy()
    rng = np.random.default_rng(1000 + batch)
    rows = np.stack([rand_fr(rng, n) for _ in range(batch)])
    if batch >= 2:
        rows[1] = 0                                              # all-zero scalars: the sum is infinity
    cached = kz.G1Points(fs, pts)
    want = cached.lin_comb_batch(rows)                         # fixed-base walk of the set (k_fb_finish / k_fb_finish_lanes)
    cached.set_table_budget_gb(0)
    got = cached.lin_comb_batch(rows)                          # bucket pipeline: k_msm_combine
    assert np.array_equal(got, want)
    for b in sorted({0, 1 % batch, batch - 1}):
        assert_points_equal(got[b], ko.lincomb_g1(pts, rows[b]))
    cached.close(); fs.close()


def test_bucket_pipeline_forms_in_fresh_processes():
    """the balanced accumulate forced at every batch size (KZG_HIP_MSM_SEG=1: lone MSMs on caller-supplied points, ragged lengths, edge scalars) and
    never (=0), through the linear-combination tests"""
    import subprocess
    import sys
    for mode, reduce in (("1", "chunks"), ("0", "scan"), ("1", "scan"), ("0", "chunks")):   # KZG_HIP_MSM_REDUCE forces a reduce form at every batch size, lone MSMs included
        env = dict(os.environ, KZG_HIP_MSM_SEG=mode, KZG_HIP_MSM_REDUCE=reduce)
        res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                              "lincomb_matches_oracle or lincomb_edge or empty_lincomb or bucket_pipeline_balanced or bucket_pipeline_reduce or commit_to_eval or proof_multi"],
                             env=env, capture_output=True, text=True, timeout=1200)
        assert res.returncode == 0, (mode, reduce, res.stdout[-1500:])


def test_generate_testing_setup(kz, fs16):
    s = ko.fr_from_ints([S_TEST])
    got = fs16.generate_testing_setup_g1(s, 33)
    assert_points_equal(got, ko.generate_testing_setup_g1(S_TEST, 33))


# ------------------------------------------------------------------ commitments / single proofs (kzg_single_proofs_test.go)
@pytest.fixture(scope="module")
def ks16(kz):
    fs = kz.FFTSettings(4)
    ks = kz.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 17))
    yield ks
    ks.close()
    fs.close()


def test_vector_A_commit_to_poly(kz, ks16):
    c = ks16.commit_to_poly(ko.fr_from_ints(TEST_POLY))
    assert comp_hex(c)[0] == DERIVED["A_commit_test_poly"]


def test_vector_B_compute_proof_single(kz, ks16):
    poly = ko.fr_from_ints(TEST_POLY)
    proof = ks16.compute_proof_single(poly, 17)
    assert comp_hex(proof)[0] == DERIVED["B_proof_single_x17"]
    d = pyref.single_proof_dlog(TEST_POLY, S_TEST, 17)     # pairing-free CheckProofSingle (kzg_single_proofs_test.go:58)
    assert ko.g1_equal(proof, ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([d])[0]))


def test_commit_to_eval_poly(kz, ks16):
    # kzg_single_proofs_test.go:11-31
    fs = ks16.fs
    poly = ko.fr_from_ints(TEST_POLY)
    eval_poly = fs.fft(poly)
    setup = ko.generate_testing_setup_g1(S_TEST, 17)
    secret_ifft = fs.fft_g1(setup[:16], inv=True)
    by_eval = kz.commit_to_eval_poly(fs, secret_ifft, eval_poly)
    by_coeffs = ks16.commit_to_poly(poly)
    assert ko.g1_equal(by_eval, by_coeffs)
    assert np.array_equal(by_eval, by_coeffs)


def test_kzg_settings_errors(kz):
    fs = kz.FFTSettings(4)
    with pytest.raises(kz.KzgPanic) as e:
        kz.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 8))    # kzg.go:25-27
    assert e.value.status == kz.ERR_LEN_MISMATCH
    fs.close()


@pytest.fixture(scope="module")
def setup_1337():
    raw = np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    return ko.g1_decompress(raw)


GLV_WALK = os.environ.get("KZG_HIP_FB_GLV") != "0"   # the table layout of this process (read once by the library): both GLV halves of a scalar walk one
                                                      # table of ceil(128 / c) windows (default), or the plain layout of rounds 1-4 (one window per c bits of the scalar)


def test_commit_c16_table_opt_in(kz, setup_1337):
    """the table of signed 16-bit windows, 16 additions per coefficient: 8 windows = 103 GB with the endomorphism (the default budget reaches it),
    16 windows = 206 GB in the plain layout (an explicit opt-in: set_table_budget_gb(210)).
    Runs before the module's shared 4096-point settings exist, so the device is empty enough for it."""
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    ks.set_table_budget_gb(210.0)
    blobs = np.stack([ko.synthetic_blob(1 + b) for b in range(3)])
    got = ks.commit_to_poly_batch(blobs)
    c, w, nbytes = ks.table_info()
    assert (c, w) == ((16, 8) if GLV_WALK else (16, 16)) and (100e9 < nbytes < 105e9 if GLV_WALK else 200e9 < nbytes < 210e9)
    assert comp_hex(got[:1])[0] == DERIVED["F_blob_seed1"]["commit_monomial_s1337"]
    assert_points_equal(got[2], ko.lincomb_g1(setup_1337, blobs[2]))
    ks.close(); fs.close()


@pytest.fixture(scope="module")
def ks4096(kz, setup_1337):
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    ks.set_table_budget_gb(62.0)     # 58 GB (c = 15, 2 x 9 windows; plain layout: c = 14, 61 GB): the module's shared settings leave room for the tests that build the
                                     # default 103 GB tables of their own (test_table_budget_setter_and_default, test_commit_c16_table_opt_in, tests/test_cold_paths.py)
    yield ks
    ks.close()
    fs.close()


def test_vector_F_and_batch_commit_4096(kz, ks4096, setup_1337):
    # BASELINE config 2: CommitToPoly on 4096-coefficient blobs, eth/trusted_setup.json
    blobs = np.stack([ko.synthetic_blob(1 + b) for b in range(6)])
    got = ks4096.commit_to_poly_batch(blobs)
    assert comp_hex(got[0])[0] == DERIVED["F_blob_seed1"]["commit_monomial_s1337"]
    assert_points_equal(got[5], ko.lincomb_g1(setup_1337, blobs[5]))
    assert_points_equal(ks4096.commit_to_poly(blobs[3]), got[3])
    short = blobs[2][:1000]                                # CommitToPoly uses SecretG1[:len(coeffs)]
    assert_points_equal(ks4096.commit_to_poly(short), ko.lincomb_g1(setup_1337[:1000], short))
    # eth.PolynomialToKZGCommitment: MSM against the bit-reversed Lagrange setup (eth/globals.go:48)
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    c = ks4096.fs.lin_comb_g1(ko.reverse_bit_order(lag), blobs[0])
    assert comp_hex(c)[0] == DERIVED["F_blob_seed1"]["commit_eth_bitrev_lagrange"]


@pytest.mark.parametrize("budget_gb,want_c,want_c_plain", [(0.05, 0, 0), (0.2, 5, 0), (0.4, 6, 5), (0.6, 7, 6), (1.0, 8, 7), (2.0, 9, 8), (3.0, 10, 9), (6.0, 11, 10), (10.0, 12, 11),
                                                           (20.0, 13, 12), (40.0, 13, 13), (70.0, 15, 14), (120.0, 16, 14)])
def test_commit_with_every_table_size(kz, setup_1337, budget_gb, want_c, want_c_plain, monkeypatch):
    # the fixed-base table adapts to the HBM budget: every window size must give the same commitments (incl. edge scalars:
    # zero digits, digits that carry through several windows, r - 1, scalars around the GLV split's boundaries); 0.05 GB fits no table (bucket
    # path); the sizing rule skips window sizes that are dominated (GLV c = 14: 2 x 10 additions like c = 13 at twice the size; plain c = 15)
    monkeypatch.setenv("KZG_HIP_FB_BUDGET_GB", str(budget_gb))
    want_c = want_c if GLV_WALK else want_c_plain
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    try:
        edge = [0, 1, 2**14, 2**15, 2**16 - 1, 2**16, (1 << 255) % ko.R_MOD, ko.R_MOD - 1, ko.R_MOD - 2**15, 0x8000800080008000, int("7fff" * 15, 16), int("8000" * 15, 16)]
        hl = LAMBDA // 2
        edge += [LAMBDA, LAMBDA - 1, LAMBDA + 1, hl, hl + 1, hl - 1, ko.R_MOD - LAMBDA, ko.R_MOD - hl, ko.R_MOD // 2, ko.R_MOD // 2 + 1, (LAMBDA * (2**126 + 12345)) % ko.R_MOD,
                 (LAMBDA * int("7fff" * 7, 16) + int("8000" * 7, 16)) % ko.R_MOD, (LAMBDA << 64) % ko.R_MOD, 2**127 - 1, 2**127, 2**128 - 1]
        blob = ko.synthetic_blob(77)[:256].copy()
        blob[:len(edge)] = ko.fr_from_ints(edge)
        got = ks.commit_to_poly(blob)
        assert ks.table_info()[0] == want_c
        assert_points_equal(got, ko.lincomb_g1(setup_1337[:256], blob))
    finally:
        ks.close()
        fs.close()


def test_projective_outputs_are_the_same_group_elements(kz, setup_1337):
    """kzg_hip_kzg_set_projective_outputs: CommitToPoly / ComputeProofSingle (single = coalesced, batch, one workgroup per blob and several) return un-normalised
    Jacobian images -- the reference's own return type (bls/bls_kilic.go:30-35) -- of exactly the points the default, normalised outputs are; infinity keeps its
    image; switching back restores Z = one; a settings object without a table (bucket path) is unaffected"""
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    ks.set_table_budget_gb(10)
    rng = np.random.default_rng(77)
    blobs = np.stack([rand_fr(rng, 4096) for _ in range(6)])
    blobs[4] = 0
    big = np.concatenate([blobs] * 100)[:520].copy()                     # one workgroup per blob, one lane finishes each
    xs = np.arange(3, 9, dtype=np.uint64)
    want_c, want_p, want_big = ks.commit_to_poly_batch(blobs), ks.compute_proof_single_batch(blobs, xs), ks.commit_to_poly_batch(big)
    one_k = ko.g1_affine(ko.g1_generator())[0][2]                        # Kilic image of Z = one
    assert all(np.array_equal(w[2], one_k) for i, w in enumerate(want_c) if i != 4)
    ks.set_projective_outputs(True)
    got_c, got_p, got_big = ks.commit_to_poly_batch(blobs), ks.compute_proof_single_batch(blobs, xs), ks.commit_to_poly_batch(big)
    singles = np.stack([ks.commit_to_poly(b) for b in blobs]).reshape(6, 3, 6)
    single_p = ks.compute_proof_single(blobs[1], 4)
    for got, want in ((got_c, want_c), (got_p, want_p), (got_big, want_big), (singles, want_c)):
        assert np.array_equal(fs.to_compressed_g1(got), fs.to_compressed_g1(want))
        assert all(ko.g1_equal(got[i], want[i]) for i in range(4))        # bls.EqualG1 on the oracle's side: projective equality
    assert np.array_equal(fs.to_compressed_g1(single_p.reshape(1, 3, 6)), fs.to_compressed_g1(want_p[1:2]))
    assert not np.array_equal(got_c[0], want_c[0]) and not np.array_equal(got_c[0][2], one_k)      # really un-normalised
    assert np.array_equal(got_c[4], want_c[4])                                                      # infinity: (0, 1, 0) either way
    ks.set_projective_outputs(False)
    assert np.array_equal(ks.commit_to_poly_batch(blobs), want_c) and np.array_equal(ks.commit_to_poly(blobs[2]).reshape(3, 6), want_c[2])
    ks.set_table_budget_gb(0)                                             # bucket path: normalised whatever the flag says
    ks.set_projective_outputs(True)
    assert np.array_equal(ks.commit_to_poly_batch(blobs), want_c)
    ks.close(); fs.close()


def test_commit_batch_shapes_fuzz(kz, ks4096, setup_1337):
    # batch sizes that exercise every launch shape (blocks per blob 32 .. 1, the chunked upload of the host-buffer form at >= 512
    # blobs, ragged last chunk) and ragged polynomial lengths; the first five rows are random (oracle MSM each), the others are
    # (b + 1) times one of them, so every commitment of the batch is checked at the cost of one oracle scalar multiplication
    rng = np.random.default_rng(2024)
    for batch, n in ((1, 4096), (3, 4095), (7, 1), (33, 130), (257, 4096), (600, 2048)):
        ints = [[int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)] for _ in range(min(batch, 5))]
        rows = [ints[b % len(ints)] if b < len(ints) else [(v * (b + 1)) % ko.R_MOD for v in ints[b % len(ints)]] for b in range(batch)]
        blobs = np.stack([ko.fr_from_ints(r) for r in rows])
        got = ks4096.commit_to_poly_batch(blobs)
        # rows beyond the first five are (b + 1) times one of them: their commitments must be (b + 1) times that commitment
        base = [ko.lincomb_g1(setup_1337[:n], blobs[b]) for b in range(len(ints))]
        for b in range(batch):
            want = base[b] if b < len(ints) else ko.g1_mul(base[b % len(ints)], ko.fr_from_ints([b + 1])[0])
            assert ko.g1_equal(got[b], want), (batch, n, b)


def test_commit_carry_chains_through_every_launch_shape(kz, ks4096, setup_1337):
    """signed-window recoding carries from window to window; below 32 polynomials the windows of a point are divided among lanes and each
    lane re-derives the carry into its first window from the lower digits (k_fb_accumulate<true>).  Scalars whose windows sit on the
    carry boundary for every window width in use (all-ones, 0x8000.., 0x7fff.., 0x8001.., r - 1, powers of two and their neighbours)
    must give the same commitment through the split walk (1, 2, 8, 16 polynomials), the plain walk (32, 130) and the oracle."""
    R = ko.R_MOD
    pats = [R - 1, R - 2, (1 << 255) % R, (1 << 254) - 1, 1, 2, 0]
    for c in (11, 13, 14, 16):
        half, full = 1 << (c - 1), (1 << c) - 1
        for d in (half, half + 1, half - 1, full, full - 1, 1):
            pats.append(sum(d << (c * w) for w in range(256 // c)) % R)
        pats.append(sum((half if w % 2 else full) << (c * w) for w in range(256 // c)) % R)
    pats += [(1 << k) % R for k in range(0, 255, 7)] + [((1 << k) - 1) % R for k in range(1, 255, 11)]
    rng = np.random.default_rng(7)
    ints = [pats[i % len(pats)] if i % 3 else int.from_bytes(rng.bytes(32), "little") % R for i in range(4096)]
    blob = ko.fr_from_ints(ints)
    want = ko.lincomb_g1(setup_1337, blob)
    assert_points_equal(ks4096.commit_to_poly(blob), want)
    for batch in (2, 8, 16, 32, 130):
        got = ks4096.commit_to_poly_batch(np.stack([blob] * batch))
        for b in (0, batch // 2, batch - 1):
            assert_points_equal(got[b], want)


def test_commit_linearity_full_size(kz, ks4096):
    # size-independent property at full size: commit(a) + commit(b) == commit(a + b)
    a, b = ko.synthetic_blob(101), ko.synthetic_blob(102)
    s = ko.fr_from_ints([(x + y) % ko.R_MOD for x, y in zip(ko.fr_to_ints(a), ko.fr_to_ints(b))])
    ca, cb, cs = ks4096.commit_to_poly_batch(np.stack([a, b, s]))
    assert ko.g1_equal(ko.g1_add(ca, cb), cs)


def test_proof_single_4096(kz, ks4096, setup_1337):
    blob = ko.synthetic_blob(7)
    proof = ks4096.compute_proof_single(blob, 17)
    d = pyref.single_proof_dlog(ko.fr_to_ints(blob), 1337, 17)
    assert ko.g1_equal(proof, ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([d])[0]))


def test_g1_mul_vec_edge_scalars(kz):
    """element-wise k_i P_i (ToeplitzPart2's loop, fk20_single.go:72-74) runs the regular odd-digit GLV schedule with the scalar split on
    the device: zero and even halves, small and full-width scalars, the point at infinity -- against the oracle's MulG1"""
    fs = kz.FFTSettings(4)
    R = ko.R_MOD
    lam = 0xac45a4010001a40200000000ffffffff
    ks_ = [0, 1, 2, 3, 15, 16, 17, 31, 32, R - 1, R - 2, lam, lam + 1, lam - 1, 2 * lam, (lam * lam) % R, 1 << 64, (1 << 64) - 1, 1 << 127, (1 << 128) - 1,
           1 << 128, (1 << 255) % R, 0x5555555555555555555555555555555555555555555555555555555555555555 % R, 7 * lam + 2, (R - 1) // 2, (R + 1) // 2]
    rng = np.random.default_rng(5)
    ks_ += [int.from_bytes(rng.bytes(32), "little") % R for _ in range(38)]
    base = ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([987654321])[0])
    pts = np.stack([base if i % 5 else ko.g1_mul(base, ko.fr_from_ints([i + 2])[0]) for i in range(len(ks_))])
    pts[7] = ko.g1_zero()[0]
    sc = ko.fr_from_ints(ks_)
    got = fs.mul_g1_vec(pts, sc)
    for i in range(len(ks_)):
        assert_points_equal(got[i], ko.g1_mul(pts[i], sc[i]))
    fs.close()


# ------------------------------------------------------------------ G1 FFT (fft_g1.go)
@pytest.mark.parametrize("n", [1, 2, 4, 8, 32])
def test_fft_g1_small_matches_oracle(kz, n):
    fs, ofs = kz.FFTSettings(6), ko.FFTSettings(6)
    pts = edge_points()
    vals = np.stack([pts[i % len(pts)] for i in range(n)])
    for inv in (False, True):
        assert_points_equal(fs.fft_g1(vals, inv), ofs.fft_g1(vals, inv))
    fs.close()


@pytest.mark.parametrize("n", [16, 256, 1024, 2048])
def test_fft_g1_lone_transforms_on_quads_and_pairs(kz, setup_1337, n):
    """a lone transform of up to 1024 points runs its direct radix-16 passes with four lanes per (output, term), 2048 points with two
    (k_g1_fft_direct_coop); setup points mixed with infinity, duplicates and opposite points, both directions, against the oracle"""
    fs, ofs = kz.FFTSettings(11), ko.FFTSettings(11)
    vals = setup_1337[:n].copy()
    edge = edge_points()
    for i in range(len(edge)):
        vals[(i * 7) % n] = edge[i]
    for inv in (False, True):
        assert_points_equal(fs.fft_g1(vals, inv), ofs.fft_g1(vals, inv))
    fs.close()


def test_fft_g1_4096_trusted_setup_lagrange(kz, setup_1337):
    # BASELINE config 3: FFTG1(setup_G1, inv) == setup_G1_lagrange, 4096 x 48 B from eth/trusted_setup.json
    fs = kz.FFTSettings(12)
    lag = fs.fft_g1(setup_1337, inv=True)
    comp = fs.to_compressed_g1(lag).tobytes()
    assert hashlib.sha256(comp).hexdigest() == PINS["setup_G1_lagrange"]
    assert comp == open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read()
    back = fs.fft_g1(lag, inv=False)
    assert_points_equal(back, setup_1337)
    fs.close()


# ------------------------------------------------------------------ FK20 (fk20_single_test.go, fk20_multi_test.go)
def test_vector_C_da_using_fk20(kz):
    fs = kz.FFTSettings(5)
    setup = ko.generate_testing_setup_g1(S_TEST, 33)
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20SingleSettings(ks, 32)
    ofs = ko.FFTSettings(5)
    ofk = ko.FK20SingleSettings(ko.KZGSettings(ofs, setup), 32)
    assert_points_equal(fk.x_ext_fft(), ofk.x_ext_fft())
    poly = ko.fr_from_ints(TEST_POLY)
    proofs = fk.da_using_fk20(poly)
    comp = ko.g1_compress(proofs)
    v = DERIVED["C_da_using_fk20_scale5"]
    assert hashlib.sha256(comp.tobytes()).hexdigest() == v["sha256"]
    assert comp[18].tobytes().hex() == v["18"]             # position 9, fk20_single_test.go:30-41
    assert_points_equal(proofs, ofk.da_using_fk20(poly))
    ext = np.concatenate([poly, ko.fr_empty(16)])
    assert_points_equal(fk.fk20_single_da_optimized(ext), ofk.fk20_single_da_optimized(ext))
    bad = ext.copy()
    bad[20] = poly[1]
    with pytest.raises(kz.KzgPanic) as e:
        fk.fk20_single_da_optimized(bad)
    assert e.value.status == kz.ERR_UPPER_HALF
    # FK20Single (no DA): 16 coefficients -> 16 proofs needs settings with n2 = 32
    assert_points_equal(fk.fk20_single(poly), ofk.fk20_single(poly))
    # ToeplitzPart2 / ToeplitzPart3 as public methods (fk20_single.go:59-87)
    tc = ko.toeplitz_coeffs_step_strided(poly, 0, 1)
    oks = ko.KZGSettings(ofs, setup)
    h_ext = ks.toeplitz_part2(tc, ofk.x_ext_fft())
    assert_points_equal(h_ext, oks.toeplitz_part2(tc, ofk.x_ext_fft()))
    assert_points_equal(ks.toeplitz_part3(h_ext), oks.toeplitz_part3(h_ext))
    fk.close(); ks.close(); fs.close()


def fk20_multi_test_poly(chunk_count=32):
    poly = []
    for i in range(chunk_count):
        vals = [1, 2, 3, 4 + i, 7, 8 + i * i, 9, 10, 13, 14, 1, 15, 0, 1000, 0, 33]
        vals[12] = ko.R_MOD - 1
        vals[14] = ko.R_MOD - 134
        poly += vals
    return poly


def test_vectors_D_E_da_using_fk20_multi(kz):
    chunk_len, chunk_count = 16, 32
    n = chunk_len * chunk_count
    fs = kz.FFTSettings(10)
    setup = ko.generate_testing_setup_g1(S_TEST, 2 * n)
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20MultiSettings(ks, 2 * n, chunk_len)
    poly = ko.fr_from_ints(fk20_multi_test_poly())
    assert comp_hex(ks.commit_to_poly(poly))[0] == DERIVED["D_commit_fk20_multi_poly"]
    proofs = fk.da_using_fk20_multi(poly)
    comp = ko.g1_compress(proofs)
    v = DERIVED["E_da_using_fk20_multi_scale10_l16"]
    assert hashlib.sha256(comp.tobytes()).hexdigest() == v["sha256"]
    assert comp[0].tobytes().hex() == v["0"] and comp[63].tobytes().hex() == v["63"]
    ofk = ko.FK20MultiSettings(ko.KZGSettings(ko.FFTSettings(10), setup), 2 * n, chunk_len)
    assert_points_equal(proofs, ofk.da_using_fk20_multi(poly))
    ext = np.concatenate([poly, ko.fr_empty(n)])
    assert_points_equal(fk.fk20_multi_da_optimized(ext), ofk.fk20_multi_da_optimized(ext))
    assert_points_equal(fk.fk20_multi(poly), ofk.fk20_multi(poly))
    with pytest.raises(kz.KzgPanic):
        kz.FK20MultiSettings(ks, 2 * n, 24)               # kzg.go:86-88
    fk.close(); ks.close(); fs.close()


def test_fk20_single_4096_config4a(kz, ks4096):
    # BASELINE config 4a: DAUsingFK20 at scale 12, poly = blob(seed 4)[:2048], setup s = 1337 -> 4096 proofs,
    # every proof checked against the pairing-free identity proof_i == [(p(s) - p(x_i)) / (s - x_i)] G1
    fk = kz.FK20SingleSettings(ks4096, 4096)
    poly = ko.synthetic_blob(4)[:2048]
    proofs = fk.da_using_fk20(poly)
    pfs = pyref.FFTSettings(12)
    poly_i = ko.fr_to_ints(poly)
    dl = pyref.bitrev(pyref.fk20_single_da_dlogs(pfs, poly_i, 1337))
    # spot-check the dlog pipeline itself against the proof identity
    for i in (0, 1, 2047, 4095):
        x = pfs.expanded[pyref.rev_bits(i, 12)]
        assert dl[i] == pyref.single_proof_dlog(poly_i, 1337, x)
    want = ko.g1_empty(4096)
    gen = ko.g1_generator()
    ks_fr = ko.fr_from_ints(dl)
    for i in range(4096):
        want[i] = ko.g1_mul(gen, ks_fr[i])
    assert_points_equal(proofs, want)
    # byte pin: SHA-256 of all 4096 compressed proofs as the CPU oracle produced them (tests/golden/fk20_pins.json)
    assert proofs_sha256(ks4096.fs, proofs) == FK20_PINS["config4a_da_using_fk20_seed4"]["sha256"]
    fk.close()


def test_fk20_multi_batched_file_accumulation(kz):
    """batches with >= 65536 output positions sum all files of a position in one lane (k_fb_mul_vec_files) instead of a lane per
    (file, position) + a summation pass: same proofs as the single-polynomial calls (which take the other path), scale 10, chunk 16"""
    import torch
    n2, l, B = 1024, 16, 1024
    fs = kz.FFTSettings(10)
    ks = kz.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, n2))
    fk = kz.FK20MultiSettings(ks, n2, l)
    rng = np.random.default_rng(1024)
    polys = np.stack([rand_fr(rng, n2 // 2) for _ in range(4)])
    polys = np.concatenate([polys] * (B // 4))                                  # 1024 rows, 4 distinct
    d_in = torch.from_numpy(polys.view(np.int64)).cuda()
    d_out = torch.zeros((B, 2 * (n2 // 2) // l, 18), dtype=torch.int64, device="cuda")
    st = kz.lib().kzg_hip_da_using_fk20_multi_batch_dev(fk.h, d_in.data_ptr(), n2 // 2, B, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint64).reshape(B, -1, 3, 6)
    for r in range(4):
        want = fk.da_using_fk20_multi(polys[r])
        assert np.array_equal(got[r], want) and np.array_equal(got[B - 4 + r], want), r
    fk.close(); ks.close(); fs.close()


# ------------------------------------------------------------------ BASELINE config 5: FK20Multi at scale 16
def test_fk20_multi_scale16_config5(kz):
    """FK20MultiDAOptimized / DAUsingFK20Multi, n2 = 65536, chunk length 16 (fk20_multi_test.go:13) -> 4096 coset proofs.
    The oracle needs minutes for the settings alone at this size, so parity is established through the coset-proof
    identity at ALL 4096 positions (pairing-free form of CheckProofMulti, fk20_multi_test.go:86), the byte pin of the oracle's
    full-size run, and linearity over all positions (a size-independent property)."""
    l, n = 16, 32768
    fs = kz.FFTSettings(16)
    setup = fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), 65536)      # GenerateTestingSetup on the device
    osetup = ko.generate_testing_setup_g1(S_TEST, 40)
    assert np.array_equal(setup[:40], ko.g1_affine(osetup))
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20MultiSettings(ks, 2 * n, l)
    a, b = ko.synthetic_blob(5, n), ko.synthetic_blob(6, n)
    ai, bi = ko.fr_to_ints(a), ko.fr_to_ints(b)
    s = ko.fr_from_ints([(x + y) % ko.R_MOD for x, y in zip(ai, bi)])
    pa, pb, ps = fk.da_using_fk20_multi(a), fk.da_using_fk20_multi(b), fk.da_using_fk20_multi(s)
    assert pa.shape == (4096, 3, 6)
    # byte pin: all 4096 coset proofs of seed 5 against the oracle's full-size run (tests/golden/make_fk20_pins.py, 7 minutes
    # of CPU there): a permutation or offset confined to positions the sampled identity below does not visit cannot hide
    pin = FK20_PINS["config5_da_using_fk20_multi_seed5"]
    assert proofs_sha256(fs, pa) == pin["sha256"]
    assert comp_hex(pa[:1])[0] == pin["first"] and comp_hex(pa[-1:])[0] == pin["last"]
    L = ko.lib()
    tmp = ko.g1_empty(1)
    for j in range(4096):                                                      # linearity, every position
        L.ko_g1_add(tmp.ctypes.data, pa[j].ctypes.data, pb[j].ctypes.data)
        assert L.ko_g1_equal(tmp.ctypes.data, ps[j].ctypes.data), j
    gen = ko.g1_generator()
    dl = _coset_proof_dlogs_scale16(ai, l)
    dfr = ko.fr_from_ints(dl)
    for pos in range(4096):
        assert ko.g1_equal(pa[pos], ko.g1_mul(gen, dfr[pos])), pos
    fk.close()
    # config 5's second variant ("also l = 128 (integration_test.go:74)"): the same 32 768 coefficients, chunk length 128 -> k = 256, 512 coset proofs.
    # Byte pin of the oracle's full-size run + the coset identity at all 512 positions + linearity
    l2 = 128
    fk2 = kz.FK20MultiSettings(ks, 2 * n, l2)
    qa, qb, qs = fk2.da_using_fk20_multi(a), fk2.da_using_fk20_multi(b), fk2.da_using_fk20_multi(s)
    assert qa.shape == (512, 3, 6)
    pin2 = FK20_PINS["config5_l128_da_using_fk20_multi_seed5"]
    assert pin2["count"] == 512 and proofs_sha256(fs, qa) == pin2["sha256"]
    assert comp_hex(qa[:1])[0] == pin2["first"] and comp_hex(qa[-1:])[0] == pin2["last"]
    for j in range(512):
        L.ko_g1_add(tmp.ctypes.data, qa[j].ctypes.data, qb[j].ctypes.data)
        assert L.ko_g1_equal(tmp.ctypes.data, qs[j].ctypes.data), j
    dfr2 = ko.fr_from_ints(_coset_proof_dlogs_scale16(ai, l2))
    for pos in range(512):
        assert ko.g1_equal(qa[pos], ko.g1_mul(gen, dfr2[pos])), pos
    # batch rows == single calls on the l = 128 settings
    qq = fk2.da_using_fk20_multi_batch(np.stack([a, b]))
    assert np.array_equal(qq[0], qa) and np.array_equal(qq[1], qb)
    fk2.close(); ks.close(); fs.close()


def _coset_proof_dlogs_scale16(ai, l):
    """discrete logs (known test secret) of all 2 k = 65536 / l coset proofs of the 32768-coefficient polynomial `ai` in DAUsingFK20Multi's returned (bit-reversed)
    order: proof = [(p(s) - I(s)) / (s^l - x^l)] G with I = p mod (X^l - x^l) (pairing-free form of CheckProofMulti, fk20_multi_test.go:86).  The l coefficients of I
    at all cosets at once: I_i = sum_t p[i + t l] (x^l)^t, and x^l runs over the 2k-th roots of unity, so coefficient i of every coset is one 2k-point transform
    (the oracle's) of the stride-l subsequence p[i::l]."""
    R = ko.R_MOD
    n = len(ai)
    k2 = 2 * n // l
    lg = k2.bit_length() - 1
    ofs = ko.FFTSettings(lg)
    sub = [ko.fr_to_ints(ofs.fft(ko.fr_from_ints(ai[i::l] + [0] * (k2 // 2)))) for i in range(l)]
    spow = [pow(S_TEST, i, R) for i in range(l + 1)]
    ps_ = pyref.eval_poly(ai, S_TEST)
    wk2 = pyref.root_of_unity(lg)
    w2n = pyref.root_of_unity(16)
    dl = []
    for pos in range(k2):
        k = pyref.rev_bits(pos, lg)                                             # domainStride = MaxWidth / n2 = 1: x = w_2n^bitrev(pos)
        i_s = sum(spow[i] * sub[i][k] for i in range(l)) % R
        dl.append((ps_ - i_s) * pow(spow[l] - pow(wk2, k, R), -1, R) % R)
    for pos in (0, 1, 2, k2 // 5, k2 // 2, k2 - 1):                             # the vectorised form against the plain restatement
        assert dl[pos] == pyref.coset_proof_dlog(ai, S_TEST, pow(w2n, pyref.rev_bits(pos, lg), R), l), pos
    return dl


# ------------------------------------------------------------------ eth/ byte-level path
def test_eth_blob_to_kzg_commitment_and_compute_kzg_proof(kz):
    fs = kz.FFTSettings(12)
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    eth = kz.EthSettings(fs, lag)                              # eth/globals.go:39-72 (applies the bit reversal itself)
    blob_i = ko.fr_to_ints(ko.synthetic_blob(1))
    blob = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in blob_i), dtype=np.uint8).reshape(4096, 32)
    c, ok = eth.blob_to_kzg_commitment(blob)                   # eth/eth.go:145-151
    assert ok and c.tobytes().hex() == DERIVED["F_blob_seed1"]["commit_eth_bitrev_lagrange"]
    # batch with an invalid blob: one element == r (bls.ValidFr, bls/bignum_all.go:12-35) -> (KZGCommitment{}, false)
    bad = blob.copy()
    bad[77] = np.frombuffer(ko.R_MOD.to_bytes(32, "little"), dtype=np.uint8)
    edge = blob.copy()
    edge[5] = np.frombuffer((ko.R_MOD - 1).to_bytes(32, "little"), dtype=np.uint8)
    outs, oks = eth.blob_to_kzg_commitment_batch(np.stack([blob, bad, edge]))
    assert list(oks) == [True, False, True]
    assert outs[0].tobytes() == c.tobytes() and not outs[1].any()
    edge_i = list(blob_i)
    edge_i[5] = ko.R_MOD - 1
    lag_br = ko.reverse_bit_order(lag)
    assert outs[2].tobytes().hex() == comp_hex(ko.lincomb_g1(lag_br, ko.fr_from_ints(edge_i)))[0]
    # ComputeKZGProof (eth/helpers.go:179-203) against the same formulas evaluated with Python integers + the oracle's MSM
    R = ko.R_MOD
    pfs = pyref.FFTSettings(12)
    dom = [pfs.expanded[pyref.rev_bits(i, 12)] for i in range(4096)]          # DomainFr, eth/globals.go:61-66
    z = 0x1234567890abcdef1234567890abcdef % R
    coeffs = pfs.fft(pyref.bitrev(blob_i), inv=True)                          # evaluations are in bit-reversed order
    y_ref = pyref.eval_poly(coeffs, z)
    q = [(p - y_ref) * pow(w - z, -1, R) % R for p, w in zip(blob_i, dom)]
    proof_ref = comp_hex(ko.lincomb_g1(lag_br, ko.fr_from_ints(q)))[0]
    proof, y = eth.compute_kzg_proof(ko.fr_from_ints(blob_i), ko.fr_from_ints([z]))
    assert ko.fr_to_ints(y)[0] == y_ref
    assert proof.tobytes().hex() == proof_ref
    d = (pyref.eval_poly(coeffs, 1337) - y_ref) * pow(1337 - z, -1, R) % R    # pairing-free VerifyKZGProof: s = 1337
    assert proof.tobytes().hex() == comp_hex(ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([d])[0]))[0]
    with pytest.raises(kz.KzgError, match="invalid z challenge"):
        eth.compute_kzg_proof(ko.fr_from_ints(blob_i), ko.fr_from_ints([dom[9]]))
    with pytest.raises(kz.KzgError, match="invalid length"):
        eth.compute_kzg_proof(ko.fr_from_ints(blob_i[:2048]), ko.fr_from_ints([z]))
    # eth.BlobToKZGCommitment is ONE blob per call: 24 concurrent callers (one of them with an invalid element) share batched launches
    import threading
    many = np.stack([blob, bad, edge] * 8)
    want, want_ok = eth.blob_to_kzg_commitment_batch(many)
    got, errs = [None] * 24, []

    def work(i):
        try:
            for _ in range(3):
                got[i] = eth.blob_to_kzg_commitment(many[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    for i in range(24):
        assert got[i][1] == bool(want_ok[i]) and got[i][0].tobytes() == want[i].tobytes(), i
    eth.close(); fs.close()


def test_eth_compute_kzg_proof_batch_and_concurrent_callers(kz):
    """kzg_hip_eth_compute_kzg_proof_batch: every row against the formulas of eth/helpers.go:179-203 evaluated with Python integers and
    the oracle's MSM over the bit-reversed Lagrange setup, incl. a row whose z lies in the domain (per-row "invalid z challenge", the
    other rows unaffected) and z = 0; then the one-polynomial entry from 20 threads (coalesced) against the batch, the invalid row
    raising for its caller only"""
    import threading
    fs = kz.FFTSettings(12)
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    eth = kz.EthSettings(fs, lag)
    R = ko.R_MOD
    pfs = pyref.FFTSettings(12)
    dom = [pfs.expanded[pyref.rev_bits(i, 12)] for i in range(4096)]
    lag_br = ko.reverse_bit_order(lag)
    polys_i = [ko.fr_to_ints(ko.synthetic_blob(300 + b)) for b in range(5)]
    zs_i = [0x1234567890abcdef % R, dom[1234], 0, R - 1, 7]              # row 1: z in the domain; row 3: z = -1 = w^(n/2) is in the domain too
    polys = np.stack([ko.fr_from_ints(p_) for p_ in polys_i])
    zs = ko.fr_from_ints(zs_i)
    proofs, ys, ok = eth.compute_kzg_proof_batch(polys, zs)
    assert list(ok) == [True, False, True, False, True]
    for b in range(5):
        if not ok[b]:
            assert not proofs[b].any() and not ys[b].any()
            continue
        coeffs = pfs.fft(pyref.bitrev(polys_i[b]), inv=True)
        y_ref = pyref.eval_poly(coeffs, zs_i[b])
        q = [(p_ - y_ref) * pow(w - zs_i[b], -1, R) % R for p_, w in zip(polys_i[b], dom)]
        assert ko.fr_to_ints(ys[b:b + 1])[0] == y_ref, b
        assert proofs[b].tobytes().hex() == comp_hex(ko.lincomb_g1(lag_br, ko.fr_from_ints(q)))[0], b
    # one polynomial per call from 20 threads: rows 0..4 repeated, four callers hold an invalid z
    got, errs = [None] * 20, [None] * 20

    def work(i):
        try:
            for _ in range(3):
                got[i] = eth.compute_kzg_proof(polys[i % 5], zs[i % 5:i % 5 + 1])
        except kz.KzgError as e:
            errs[i] = e
    ts = [threading.Thread(target=work, args=(i,)) for i in range(20)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(20):
        b = i % 5
        if ok[b]:
            assert errs[i] is None and got[i][0].tobytes() == proofs[b].tobytes() and np.array_equal(got[i][1], ys[b]), i
        else:
            assert errs[i] is not None and "invalid z challenge" in str(errs[i]), i
    eth.close(); fs.close()


def test_evaluate_poly_in_evaluation_form(kz):
    """TestEvaluatePolyInEvaluationForm (fft_fr_test.go:73-99): coefficients -> FFT -> barycentric evaluation at random x == Horner on the
    coefficients (bls.EvalPolyAt, restated in pyref.eval_poly); at scale 4 as in the reference and at 4096 points with a strided domain
    (scale 1 of a 8192-wide settings object); eth's form on the bit-reversed domain; x inside the domain gives the reference's 0"""
    R = ko.R_MOD
    rng = np.random.default_rng(11)

    def rand_ints(k):
        return [int.from_bytes(rng.bytes(32), "little") % R for _ in range(k)]
    for max_scale, scale in ((4, 0), (13, 1), (12, 0)):
        fs = kz.FFTSettings(max_scale)
        n = (1 << max_scale) >> scale
        coeffs = rand_ints(n)
        pfs = pyref.FFTSettings(max_scale - scale)
        evals = pfs.fft(coeffs)
        assert ko.fr_to_ints(fs.fft(ko.fr_from_ints(coeffs))) == evals if scale == 0 else True
        for x in rand_ints(5 if n > 100 else 100) + [0]:
            y = fs.evaluate_poly_in_evaluation_form(ko.fr_from_ints(evals), ko.fr_from_ints([x]), scale)
            assert ko.fr_to_ints(y.reshape(1, 4))[0] == pyref.eval_poly(coeffs, x)
        # x inside the domain: the reference's last factor (x^n - 1) / n is zero there, so it returns 0 (bls/globals.go:141-152), not f(x)
        y = fs.evaluate_poly_in_evaluation_form(ko.fr_from_ints(evals), ko.fr_from_ints([pfs.expanded[3]]), scale)
        assert ko.fr_to_ints(y.reshape(1, 4))[0] == 0
        with pytest.raises(kz.KzgPanic):
            fs.evaluate_poly_in_evaluation_form(ko.fr_from_ints(evals[: n // 2]), ko.fr_from_ints([5]), scale)
        if max_scale == 12:
            lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
            eth = kz.EthSettings(fs, lag)
            y = eth.evaluate_polynomial_in_evaluation_form(ko.fr_from_ints(pyref.bitrev(evals)), ko.fr_from_ints([12345]))
            assert ko.fr_to_ints(y.reshape(1, 4))[0] == pyref.eval_poly(coeffs, 12345)
            eth.close()
        fs.close()


def test_eth_compute_aggregate_kzg_proof(kz):
    """eth.ComputeAggregateKZGProof (eth/eth.go:175-182) and the prover-side pieces of VerifyAggregateKZGProof (:155-172) against the restatement of
    eth/helpers.go:113-176,215-260 in oracle/pyref.py (hashlib transcript, Python-integer aggregation) + the oracle's MSM, and against the
    pairing-free identity with s = 1337; blocks of 3, 1 and 0 blobs; an invalid field element; an undecodable commitment"""
    fs = kz.FFTSettings(12)
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    eth = kz.EthSettings(fs, lag)
    R = ko.R_MOD
    pfs = pyref.FFTSettings(12)
    dom = [pfs.expanded[pyref.rev_bits(i, 12)] for i in range(4096)]
    lag_br = ko.reverse_bit_order(lag)
    polys_i = [ko.fr_to_ints(ko.synthetic_blob(700 + b)) for b in range(3)]
    polys_i[1][17] = R - 1                                           # the largest valid element
    blobs = np.stack([np.frombuffer(b"".join(v.to_bytes(32, "little") for v in p_), dtype=np.uint8).reshape(4096, 32) for p_ in polys_i])
    for count in (3, 1, 0):
        proof, comm = eth.compute_aggregate_kzg_proof(blobs[:count])
        want_comm, ok = eth.blob_to_kzg_commitment_batch(blobs[:count]) if count else (np.zeros((0, 48), dtype=np.uint8), np.zeros(0, dtype=bool))
        assert ok.all() and comm.tobytes() == want_comm.tobytes()
        agg, powers, z = pyref.compute_aggregated_poly(polys_i[:count], [c.tobytes() for c in comm])
        y_ref = pyref.eval_in_evaluation_form(agg, z, dom)
        q = [(p_ - y_ref) * pow(w - z, -1, R) % R for p_, w in zip(agg, dom)]
        assert proof.tobytes().hex() == comp_hex(ko.lincomb_g1(lag_br, ko.fr_from_ints(q)))[0], count
        coeffs = pfs.fft(pyref.bitrev(agg), inv=True)
        assert pyref.eval_poly(coeffs, z) == y_ref
        d = (pyref.eval_poly(coeffs, 1337) - y_ref) * pow(1337 - z, -1, R) % R       # pairing-free VerifyKZGProof with the setup's secret
        assert proof.tobytes().hex() == comp_hex(ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([d])[0]))[0], count
        if count == 0:
            assert proof.tobytes().hex() == "c0" + "00" * 47                              # the zero polynomial's proof: the point at infinity
        # the verifier's side on the same block
        poly, c_agg, z_got, y_got = eth.compute_aggregated_poly_and_commitment(blobs[:count], comm)
        assert ko.fr_to_ints(poly) == agg and ko.fr_to_ints(z_got.reshape(1, 4))[0] == z and ko.fr_to_ints(y_got.reshape(1, 4))[0] == y_ref
        want_c = ko.lincomb_g1(ko.g1_decompress(comm.reshape(-1)), ko.fr_from_ints(powers)) if count else ko.g1_zero(1)
        assert comp_hex(c_agg.reshape(1, -1)) == comp_hex(want_c), count
        # the aggregated commitment commits to the aggregated polynomial
        assert comp_hex(c_agg.reshape(1, -1)) == comp_hex(ko.lincomb_g1(lag_br, ko.fr_from_ints(agg)))
    bad = blobs.copy()
    bad[2, 4095] = np.frombuffer(R.to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(kz.KzgError, match="could not convert blobs"):
        eth.compute_aggregate_kzg_proof(bad)
    _, comm = eth.compute_aggregate_kzg_proof(blobs)
    with pytest.raises(kz.KzgError, match="could not convert blobs"):
        eth.compute_aggregated_poly_and_commitment(bad, comm)
    broken = comm.copy()
    broken[1, 47] ^= 1                                               # x no longer on the curve (or not in the subgroup)
    with pytest.raises(kz.KzgError, match="invalid commitment"):
        eth.compute_aggregated_poly_and_commitment(blobs, broken)
    eth.close(); fs.close()


def test_host_buffer_calls_from_many_threads_use_the_stream_pool(kz):
    """FFT / FFTG1 / DASFFTExtension / uncached LinCombG1 on host buffers lease a stream of the handle's pool instead of serialising on one
    stream under the handle mutex: 16 threads get bit-identical results, and their FFT_Fr(4096) calls overlap (aggregate rate well above one
    thread's; the measured ratio is printed for DESIGN.md)"""
    import threading
    fs, ofs = kz.FFTSettings(12), ko.FFTSettings(12)
    rng = np.random.default_rng(16)
    T = 16
    vals = [rand_fr(rng, 4096) for _ in range(T)]
    want = [ofs.fft(v) for v in vals[:4]]
    pts = ko.g1_decompress(np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)[:48 * 64])
    want_g1 = ofs.fft_g1(pts)
    want_lc = ko.lincomb_g1(pts, vals[0][:64])
    errs = []

    def mixed(i):
        try:
            for _ in range(3):
                assert np.array_equal(fs.fft(vals[i % 4]), want[i % 4])
                assert np.array_equal(fs.fft(fs.fft(vals[i]), inv=True), vals[i])
                if i % 4 == 0:
                    assert ko.g1_equal(fs.fft_g1(pts), want_g1)
                if i % 4 == 1:
                    assert ko.g1_equal(fs.lin_comb_g1(pts, vals[0][:64]).reshape(1, 3, 6), want_lc.reshape(1, 3, 6))
                if i % 4 == 2:
                    assert np.array_equal(fs.das_fft_extension(vals[i][:2048].copy()), ofs.das_fft_extension(vals[i][:2048].copy()))
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    ts = [threading.Thread(target=mixed, args=(i,)) for i in range(T)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]

    rows = np.stack(vals)
    fs.bench_threads_fft(rows, 2, 20)
    r1, _ = fs.bench_threads_fft(rows, 1, 400)
    r16, outs = fs.bench_threads_fft(rows, 16, 200)              # native threads: the interpreter lock would cap Python threads near 25k calls/s
    print("FFT_Fr(4096) host-buffer calls/s: 1 thread %.0f, 16 threads %.0f (x%.1f)" % (r1, r16, r16 / r1))
    for i in range(4):
        assert np.array_equal(outs[i], want[i])
    assert r16 > 2.5 * r1                                     # x3.9 with ROCm's default 4 hardware queues per process, x5.4 with GPU_MAX_HW_QUEUES=8
    fs.close()


# ------------------------------------------------------------------ host-buffer batch APIs, concurrency, C-level misuse
def test_da_using_fk20_batch_host_buffers(kz, ks4096):
    fk = kz.FK20SingleSettings(ks4096, 4096)
    polys = np.stack([ko.synthetic_blob(40 + b)[:2048] for b in range(6)])
    got = fk.da_using_fk20_batch(polys[:3])                  # 3 and 4 transforms: direct passes of radix 8; 1-2: radix 16; 5+: radix-2 network
    assert got.shape == (3, 4096, 3, 6)
    singles = [fk.da_using_fk20(polys[b]) for b in range(6)]
    for b in range(3):
        assert np.array_equal(got[b], singles[b])
    for nb in (2, 4, 6):
        gotb = fk.da_using_fk20_batch(polys[:nb])
        for b in range(nb):
            assert np.array_equal(gotb[b], singles[b]), (nb, b)
    # 9: two lanes per butterfly; 17 and 63: ragged batches run padded with copies of their last polynomial (to 32 / 64), the copies' proofs dropped
    many = np.stack([ko.synthetic_blob(140 + b)[:2048] for b in range(63)])
    for nb, rows in ((9, range(9)), (17, (0, 8, 16)), (63, (0, 31, 62))):
        gotb = fk.da_using_fk20_batch(many[:nb])
        assert gotb.shape[0] == nb
        for b in rows:
            assert np.array_equal(gotb[b], fk.da_using_fk20(many[b])), (nb, b)
    fk.close()


def test_fk20_single_batch_host_buffers(kz):
    """FK20Single (fk20_single.go:122-137) on batches: from 5 polynomials on the Toeplitz stage is fused with two DIF stages and the second
    transform (k points) continues from the even positions of the bit-reversed layout; 1-4 polynomials take the direct passes.  Against the
    oracle at scale 8 and against one-polynomial calls; the same rows through the DA form of the same settings"""
    n2 = 256
    fs = kz.FFTSettings(8)
    setup = ko.generate_testing_setup_g1(S_TEST, n2)
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20SingleSettings(ks, n2)
    ofk = ko.FK20SingleSettings(ko.KZGSettings(ko.FFTSettings(8), setup), n2)
    rng = np.random.default_rng(256)
    polys = np.stack([rand_fr(rng, n2 // 2) for _ in range(40)])
    polys[3, 100:] = 0
    polys[7] = 0                                              # the zero polynomial: every proof is the point at infinity
    singles = [fk.fk20_single(p) for p in polys[:9]]
    for b in (0, 3, 7):
        assert_points_equal(singles[b], ofk.fk20_single(polys[b]))
    for nb in (1, 3, 5, 9, 17, 40):
        got = fk.fk20_single_batch(polys[:nb])
        assert got.shape == (nb, n2 // 2, 3, 6)
        for b in range(min(nb, 9)):
            assert np.array_equal(got[b], singles[b]), (nb, b)
    assert np.array_equal(fk.fk20_single_batch(polys)[39], fk.fk20_single(polys[39]))
    da = fk.da_using_fk20_batch(polys[:9])
    for b in (0, 7, 8):
        assert np.array_equal(da[b], fk.da_using_fk20(polys[b])), b
    with pytest.raises(kz.KzgPanic) as e:
        fk.fk20_single_batch(polys[:, :64])
    assert e.value.status == kz.ERR_LEN_MISMATCH
    fk.close(); ks.close(); fs.close()


def test_fk20_paths_agree_in_a_fresh_process():
    """FK20 runs through three pipelines depending on size: direct radix-16 passes (a lone transform), the radix-2 network, and -- for
    DA forms of single-file settings with a resident table -- the Toeplitz stage fused with two decimation-in-frequency stages.  The
    choice is made once per process, so the small-scale KATs and the config-4a byte pin are re-run in child processes that force the
    radix-2 / fused pipeline (KZG_HIP_G1_FFT=radix2) and the unfused one (KZG_HIP_FK20_FUSE=0) at every size."""
    import subprocess
    import sys
    # ... without the Toeplitz stage fused into the first direct pass of a lone polynomial (KZG_HIP_FK20_PASS1=0) ...
    # ... and with four lanes per butterfly everywhere (KZG_HIP_G1_QUAD=1: both digit schedules of g1_quad.hpp), two everywhere (=2) and one (=0: the radix-8 direct passes return)
    for extra in ({"KZG_HIP_G1_FFT": "radix2"}, {"KZG_HIP_G1_FFT": "radix2", "KZG_HIP_FK20_FUSE": "0"}, {"KZG_HIP_G1_FFT": "direct"},
                  {"KZG_HIP_G1_FFT": "radix2", "KZG_HIP_G1_QUAD": "1"}, {"KZG_HIP_G1_FFT": "radix2", "KZG_HIP_G1_QUAD": "1", "KZG_HIP_G1_MUL": "regular"},
                  {"KZG_HIP_G1_FFT": "radix2", "KZG_HIP_G1_QUAD": "2"}, {"KZG_HIP_G1_FFT": "radix2", "KZG_HIP_G1_QUAD": "2", "KZG_HIP_G1_MUL": "regular"},
                  {"KZG_HIP_G1_QUAD": "0"}, {"KZG_HIP_FK20_PASS1": "0"}):
        env = dict(os.environ, **extra)
        # (the lone-transform test spends seconds in the oracle: only where the forced setting changes what it runs)
        lone = " or fft_g1_lone" if extra.get("KZG_HIP_G1_QUAD") in ("0", "2") and "KZG_HIP_G1_MUL" not in extra else ""
        res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                              "vector_C or vectors_D or config4a or batch_host_buffers or fft_g1_small or full_das_flow" + lone],
                             env=env, capture_output=True, text=True, timeout=1200)
        assert res.returncode == 0, (extra, res.stdout[-1500:])


def test_concurrent_callers_share_a_handle(kz, ks4096, setup_1337):
    """the reference's settings are read-only after construction (SURVEY.md 8b threading); the library serialises per handle"""
    import threading
    blobs = [ko.synthetic_blob(60 + i) for i in range(4)]
    want = [ks4096.commit_to_poly(b) for b in blobs]
    got, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                got[i] = ks4096.commit_to_poly(blobs[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for i in range(4):
        assert np.array_equal(got[i], want[i])


@pytest.mark.timeout(300)
def test_coalescer_with_callers_joining_and_leaving(kz, ks4096):
    """coalesce.hpp sleeps its callers on futex words and elects batch leaders among them: threads that make different numbers of
    calls (so the concurrency keeps changing, batches of every size form, leaders are elected while others leave) must all return,
    each with its own result.  A lost wake-up would hang this test (pytest-timeout)."""
    import threading
    T = 48
    blobs = np.stack([ko.synthetic_blob(1200 + i) for i in range(8)])
    want = ks4096.commit_to_poly_batch(blobs)
    xs = np.arange(3, 3 + 8, dtype=np.uint64)
    want_p = ks4096.compute_proof_single_batch(blobs, xs)
    bad, errs = [], []

    def work(i):
        try:
            for r in range((i % 7 + 1) * 3):
                j = (i + r) % 8
                if (i + r) % 5 == 0:
                    if not np.array_equal(ks4096.compute_proof_single(blobs[j], int(xs[j])), want_p[j]):
                        bad.append((i, r, "proof"))
                elif not np.array_equal(ks4096.commit_to_poly(blobs[j]), want[j]):
                    bad.append((i, r, "commit"))
                if i % 3 == 0 and r % 4 == 3:
                    import time
                    time.sleep(0.0007)                       # some callers pause: batches close on the window, not on the count
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    for _ in range(3):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    assert not errs, errs[:2]
    assert not bad, bad[:5]


@pytest.mark.timeout(300)
def test_table_budget_changes_while_coalesced_commitments_run(kz, setup_1337):
    """kzg_hip_kzg_set_table_budget_gb frees the fixed-base table; coalesced batches walk it outside the handle mutex.  The table's
    lifetime lock must make the two safe together: 8 threads keep committing while the budget flips between two table sizes."""
    import threading
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    ks.set_table_budget_gb(1.0)
    blobs = np.stack([ko.synthetic_blob(1500 + i) for i in range(4)])
    want = ks.commit_to_poly_batch(blobs)
    stop, bad, errs = threading.Event(), [], []

    def work(i):
        try:
            r = 0
            while not stop.is_set():
                j = (i + r) % 4
                if not np.array_equal(ks.commit_to_poly(blobs[j]), want[j]):
                    bad.append((i, r))
                r += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    try:
        for k in range(6):
            ks.set_table_budget_gb(3.0 if k % 2 == 0 else 1.0)
            assert np.array_equal(ks.commit_to_poly(blobs[0]), want[0])
    finally:
        stop.set()
        [t.join() for t in ts]
    assert not errs, errs[:2]
    assert not bad, bad[:5]
    ks.close(); fs.close()


def test_lone_caller_pays_no_gather_window_after_a_burst(kz, ks4096):
    """coalesce.hpp: the gather target of a batch leader follows the recent concurrency and must decay back to ONE caller -- a lone
    caller after a burst of 16 threads would otherwise wait the 150 us window on every call (regression: the decay stalled at 3)."""
    import time
    blobs = np.stack([ko.synthetic_blob(900 + i) for i in range(16)])

    def median_ms(reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ks4096.commit_to_poly(blobs[0])
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    for _ in range(40):
        ks4096.commit_to_poly(blobs[0])                      # table built, any earlier concurrency forgotten
    before = median_ms(40)
    rate, _ = ks4096.bench_drop_in(blobs, 16, 10)
    assert rate > 0
    for _ in range(40):
        ks4096.commit_to_poly(blobs[0])                      # the decay: a quarter per batch
    after = median_ms(40)
    assert after < before + 0.10, (before, after)            # the window is 0.15 ms


def test_64_threads_single_blob_calls_are_coalesced_and_bit_exact(kz, ks4096, setup_1337):
    """the reference API is one polynomial per call (kzg_single_proofs.go:17-19,36-54); 64 host threads calling it concurrently
    are merged into batched launches (coalesce.hpp).  Every result is compared with the batched call, a sample with the oracle;
    ragged lengths and per-call x values share batches."""
    import threading
    rng = np.random.default_rng(64)
    T, ROUNDS = 64, 3
    blobs = np.stack([ko.synthetic_blob(500 + i) for i in range(T)])
    lens = [4096 if i % 4 else int(rng.integers(2, 4096)) for i in range(T)]
    xs = [int(rng.integers(1, 2**63)) for _ in range(T)]
    want_c = ks4096.commit_to_poly_batch(blobs)
    want_p = ks4096.compute_proof_single_batch(blobs, np.array(xs, dtype=np.uint64))
    got_c = [[None] * ROUNDS for _ in range(T)]
    got_r = [None] * T
    got_p = [None] * T
    errs = []
    start = threading.Barrier(T)

    def work(i):
        try:
            start.wait()
            for r in range(ROUNDS):
                got_c[i][r] = ks4096.commit_to_poly(blobs[i])
            got_r[i] = ks4096.commit_to_poly(blobs[i][:lens[i]])
            got_p[i] = ks4096.compute_proof_single(blobs[i], xs[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    for i in range(T):
        for r in range(ROUNDS):
            assert np.array_equal(got_c[i][r], want_c[i]), (i, r)
        assert np.array_equal(got_p[i], want_p[i]), i
    for i in (0, 4, 17, 40, 63):
        assert_points_equal(got_c[i][0], ko.lincomb_g1(setup_1337, blobs[i]))
        assert_points_equal(got_r[i], ko.lincomb_g1(setup_1337[:lens[i]], blobs[i][:lens[i]]))
    oks = ko.KZGSettings(ko.FFTSettings(12), setup_1337)
    for i in (3, 33):
        assert_points_equal(got_p[i], oks.compute_proof_single(blobs[i], xs[i]))
    # FK20 all-proofs through the same mechanism: 6 threads, one polynomial each
    fk = kz.FK20SingleSettings(ks4096, 4096)
    polys = np.stack([ko.synthetic_blob(700 + i)[:2048] for i in range(6)])
    want_f = fk.da_using_fk20_batch(polys)
    got_f = [None] * 6

    def fwork(i):
        try:
            got_f[i] = fk.da_using_fk20(polys[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=fwork, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    for i in range(6):
        assert np.array_equal(got_f[i], want_f[i]), i
    fk.close()


def test_dev_entry_points_on_two_streams_do_not_share_scratch(kz, ks4096):
    """_dev calls enqueue and return; two callers on DIFFERENT streams used to share one per-handle partial-sum workspace
    (round-1 advisor finding).  Interleave launches of different batch shapes on two streams and compare with serial results."""
    import torch
    L = kz.lib()
    blobs = np.stack([ko.synthetic_blob(300 + i) for i in range(40)])
    d_in = torch.from_numpy(blobs.view(np.int64)).cuda()
    want = ks4096.commit_to_poly_batch(blobs)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        o1 = torch.zeros((33, 18), dtype=torch.int64, device="cuda")
        o2 = torch.zeros((7, 18), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):                                   # several launches in flight on each stream
            assert L.kzg_hip_commit_to_poly_batch_dev(ks4096.h, d_in.data_ptr(), 4096, 33, o1.data_ptr(), s1.cuda_stream) == 0
            assert L.kzg_hip_commit_to_poly_batch_dev(ks4096.h, d_in[33:].data_ptr(), 4096, 7, o2.data_ptr(), s2.cuda_stream) == 0
        torch.cuda.synchronize()
        assert np.array_equal(o1.cpu().numpy().view(np.uint64).reshape(-1, 3, 6), want[:33])
        assert np.array_equal(o2.cpu().numpy().view(np.uint64).reshape(-1, 3, 6), want[33:])


def test_trusted_setup_from_json(kz, setup_1337):
    """JSONTrustedSetup (eth/globals.go:33-49) through the C ABI: the document is rebuilt here in the format of
    eth/trusted_setup.json from the committed 48-byte fixtures (the 2 MB file itself is not committed)"""
    fs = kz.FFTSettings(12)
    mono = open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read()
    lag = open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read()
    hexes = lambda raw: [raw[48 * i:48 * i + 48].hex() for i in range(len(raw) // 48)]      # noqa: E731
    doc = json.dumps({"setup_G1": hexes(mono), "setup_G2": ["00" * 96] * 3, "setup_G1_lagrange": hexes(lag), "roots_of_unity": [1, 2, 3]}, indent=2)
    got_mono, got_lag = fs.trusted_setup_from_json(doc)
    assert got_mono.shape == (4096, 3, 6) and got_lag.shape == (4096, 3, 6)
    assert np.array_equal(got_mono, ko.g1_affine(setup_1337))
    assert hashlib.sha256(fs.to_compressed_g1(got_lag).tobytes()).hexdigest() == PINS["setup_G1_lagrange"]
    # text round trip and the error behaviour of UnmarshalText (bls/bls_all.go:24-39)
    assert fs.g1_marshal_text(got_mono[:5]) == hexes(mono)[:5]
    assert np.array_equal(fs.g1_unmarshal_text(hexes(mono)[:5]), got_mono[:5])
    with pytest.raises(kz.KzgPanic) as e:
        fs.g1_unmarshal_text(["zz" + hexes(mono)[1][2:]])
    assert e.value.status == kz.ERR_BAD_POINT
    with pytest.raises(kz.KzgPanic):
        fs.trusted_setup_from_json(json.dumps({"setup_G1": [hexes(mono)[0][:-2]]}))           # 47 bytes
    with pytest.raises(kz.KzgPanic):
        fs.trusted_setup_from_json(json.dumps({"something": "else"}))
    only_lag, = [fs.trusted_setup_from_json(json.dumps({"setup_G1_lagrange": hexes(lag)[:8]}))]
    assert only_lag[0].shape[0] == 0 and only_lag[1].shape[0] == 8
    fs.close()


def test_table_budget_setter_and_default(kz, setup_1337, monkeypatch):
    """default budget 110 GB -> c = 16 on 8 windows (103 GB) for n = 4096 with the endomorphism (plain layout: c = 14, 61 GB -- c = 15 is skipped);
    the setter rebuilds at another size; results identical"""
    monkeypatch.delenv("KZG_HIP_FB_BUDGET_GB", raising=False)
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    blob = ko.synthetic_blob(1)
    c0 = ks.commit_to_poly(blob)
    assert ks.table_info()[:2] == ((16, 8) if GLV_WALK else (14, 19)) and ks.table_info()[2] <= 110e9
    ks.set_table_budget_gb(1.0)
    assert ks.table_info() == (0, 0, 0)
    c1 = ks.commit_to_poly(blob)
    assert ks.table_info()[0] == (8 if GLV_WALK else 7)
    assert np.array_equal(c0, c1)
    assert comp_hex(c0[None])[0] == DERIVED["F_blob_seed1"]["commit_monomial_s1337"]
    ks.close(); fs.close()


def test_c_abi_misuse_returns_status_codes(kz):
    import ctypes as C
    L = kz.lib()
    fs = kz.FFTSettings(4)
    buf = ko.fr_empty(16)
    assert L.kzg_hip_fft_fr(None, buf.ctypes.data, 16, 0, buf.ctypes.data, None) == kz.ERR_BAD_ARG
    assert L.kzg_hip_fft_fr(fs.h, None, 16, 0, buf.ctypes.data, None) == kz.ERR_BAD_ARG
    assert L.kzg_hip_fft_g1(fs.h, buf.ctypes.data, 0, 0, buf.ctypes.data) == kz.ERR_BAD_ARG      # reference divides by zero (fft_g1.go:76)
    assert L.kzg_hip_inplace_fft_fr(fs.h, buf.ctypes.data, buf.ctypes.data, 0, 0) == kz.OK       # IsPowerOfTwo(0) is true (bls/globals.go:72-74)
    h = C.c_void_p()
    assert L.kzg_hip_fft_settings_new(99, 4, C.byref(h)) == kz.ERR_NO_DEVICE
    assert L.kzg_hip_fft_settings_new(0, 40, C.byref(h)) == kz.ERR_BAD_ARG
    ks = kz.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, 17))
    out = ko.g1_empty(1)
    assert L.kzg_hip_commit_to_poly(ks.h, buf.ctypes.data, 18, out.ctypes.data) == kz.ERR_LEN_MISMATCH   # SecretG1[:18] out of range
    assert L.kzg_hip_compute_proof_single(ks.h, buf.ctypes.data, 1, 17, out.ctypes.data) == kz.ERR_BAD_ARG
    with pytest.raises(kz.KzgPanic):
        kz.FK20SingleSettings(ks, 64)                                                                  # kzg.go:44-46
    with pytest.raises(kz.KzgPanic):
        kz.FK20SingleSettings(ks, 12)                                                                  # kzg.go:47-49
    fk = kz.FK20SingleSettings(ks, 16)
    with pytest.raises(kz.KzgPanic) as e:
        fk.da_using_fk20(ko.fr_from_ints(range(4)))                                                     # settings built for n2 = 16
    assert e.value.status == kz.ERR_LEN_MISMATCH
    fk.close(); ks.close(); fs.close()


# ------------------------------------------------------------------ BASELINE config 4b: FK20Single, full 4096-coefficient blob
def test_fk20_single_scale13_config4b(kz):
    """FK20Single (fk20_single.go:122-134) on a full 4096-coefficient blob needs scale 13 and an 8192-point setup
    (GenerateTestingSetup with the reference's test secret, generated on the device).  out[i] is the proof at w_4096^i.  Byte pins of the
    oracle's full-size runs plus the proof identity at every position of both forms."""
    n = 4096
    fs = kz.FFTSettings(13)
    setup = fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), 8192)
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20SingleSettings(ks, 2 * n)
    blob = ko.synthetic_blob(1)
    poly_i = ko.fr_to_ints(blob)
    proofs = fk.fk20_single(blob)
    assert proofs.shape == (n, 3, 6)
    assert proofs_sha256(fs, proofs) == FK20_PINS["config4b_fk20_single_seed1"]["sha256"]      # all 4096 proofs, oracle byte pin
    w = pyref.root_of_unity(12)
    gen = ko.g1_generator()
    for i in (0, 1, 2, 1234, 4095):
        d = pyref.single_proof_dlog(poly_i, S_TEST, pow(w, i, ko.R_MOD))
        assert ko.g1_equal(proofs[i], ko.g1_mul(gen, ko.fr_from_ints([d])[0])), i
    # ... and at EVERY position: (p(s) - p(w^i)) / (s - w^i) with all p(w^i) from one oracle transform
    R = ko.R_MOD
    ofs13 = ko.FFTSettings(13)
    ps_ = pyref.eval_poly(poly_i, S_TEST)
    ev = ko.fr_to_ints(ofs13.fft(blob))                                        # p on the 4096-point domain
    dl = ko.fr_from_ints([(ps_ - ev[i]) * pow(S_TEST - pow(w, i, R), -1, R) % R for i in range(n)])
    for i in range(n):
        assert ko.g1_equal(proofs[i], ko.g1_mul(gen, dl[i])), i
    # the batch form (what bench.py times as fk20_4096): 6 blobs in one call, row 0 is the pinned blob, rows equal one-polynomial calls
    six = np.stack([ko.synthetic_blob(1 + b) for b in range(6)])
    got6 = fk.fk20_single_batch(six)
    assert np.array_equal(got6[0], proofs) and np.array_equal(got6[5], fk.fk20_single(six[5]))
    da6 = fk.da_using_fk20_batch(six)
    assert proofs_sha256(fs, da6[0]) == FK20_PINS["config4b_da_using_fk20_seed1"]["sha256"] and np.array_equal(da6[5], fk.da_using_fk20(six[5]))
    # ... and equals ComputeProofSingle at an integer point as a cross-check of the same settings
    assert ko.g1_equal(ks.compute_proof_single(blob, 17), ko.g1_mul(gen, ko.fr_from_ints([pyref.single_proof_dlog(poly_i, S_TEST, 17)])[0]))
    # DA form on the same settings: 4096 coefficients -> 8192 proofs, sample + linearity
    pa = fk.da_using_fk20(blob)
    assert pa.shape == (2 * n, 3, 6)
    assert proofs_sha256(fs, pa) == FK20_PINS["config4b_da_using_fk20_seed1"]["sha256"]        # all 8192 proofs of the DA form
    w2 = pyref.root_of_unity(13)
    for pos in (0, 3, 8191):
        d = pyref.single_proof_dlog(poly_i, S_TEST, pow(w2, pyref.rev_bits(pos, 13), ko.R_MOD))
        assert ko.g1_equal(pa[pos], ko.g1_mul(gen, ko.fr_from_ints([d])[0])), pos
    ev2 = ko.fr_to_ints(ofs13.fft(np.concatenate([blob, ko.fr_empty(n)])))  # p on the 8192-point domain
    dl2 = []
    for pos in range(2 * n):
        k = pyref.rev_bits(pos, 13)
        dl2.append((ps_ - ev2[k]) * pow(S_TEST - pow(w2, k, R), -1, R) % R)
    dl2 = ko.fr_from_ints(dl2)
    for pos in range(2 * n):
        assert ko.g1_equal(pa[pos], ko.g1_mul(gen, dl2[pos])), pos
    fk.close(); ks.close(); fs.close()


# ------------------------------------------------------------------ KZG multi proofs, prover side
def test_compute_proof_multi_and_interpolation_commitment(kz):
    fs = kz.FFTSettings(4)
    setup = ko.generate_testing_setup_g1(S_TEST, 17)
    ks, oks = kz.KZGSettings(fs, setup), ko.KZGSettings(ko.FFTSettings(4), setup)
    poly = ko.fr_from_ints(TEST_POLY)
    assert_points_equal(ks.compute_proof_multi(poly, 5431, 8), oks.compute_proof_multi(poly, 5431, 8))   # kzg_multi_proofs_test.go:46
    fs3 = kz.FFTSettings(3)
    setup9 = ko.generate_testing_setup_g1(S_TEST, 9)
    ks8, oks8 = kz.KZGSettings(fs3, setup9), ko.KZGSettings(ko.FFTSettings(3), setup9)
    w8 = pyref.root_of_unity(3)
    ys = ko.fr_from_ints([pyref.eval_poly(TEST_POLY, 5431 * pow(w8, i, ko.R_MOD) % ko.R_MOD) for i in range(8)])
    x = ko.fr_from_ints([5431])
    is1, xpow = ks8.check_proof_multi_interpolation(ys, x)
    o_is1, o_xpow = oks8.check_proof_multi_interpolation(ys, x[0])
    assert_points_equal(is1, o_is1)
    assert np.array_equal(xpow, o_xpow)
    # full size: 4096 coefficients, coset of 16
    fs12 = kz.FFTSettings(12)
    raw = np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    s1337 = ko.g1_decompress(raw)
    ksb = kz.KZGSettings(fs12, s1337)
    blob = ko.synthetic_blob(21)
    assert_points_equal(ksb.compute_proof_multi(blob, 99, 16), ko.lincomb_g1(s1337[:4080], blob[16:]))
    ksb.close(); ks8.close(); ks.close(); fs12.close(); fs3.close(); fs.close()


# ------------------------------------------------------------------ erasure recovery
def test_zero_poly_python_kat_gpu(kz):
    k = KATS["test_zero_poly_python"]                      # zero_poly_test.go:133-198
    fs = kz.FFTSettings(k["scale"])
    missing = [i for i, e in enumerate(k["exists"]) if not e]
    ze, zp = fs.zero_poly_via_multiplication(missing, 16)
    assert ko.fr_to_ints(ze) == [int(v) for v in k["expected_eval"]]
    assert ko.fr_to_ints(zp) == [int(v) for v in k["expected_poly"]]
    ze0, zp0 = fs.zero_poly_via_multiplication([], 16)     # zero_poly.go:117-119
    assert not ze0.any() and not zp0.any()
    with pytest.raises(kz.KzgPanic):
        fs.zero_poly_via_multiplication([1], 32)           # domain too small
    fs.close()


def test_zero_poly_ragged_erasure_sets(kz):
    """erasure counts around the product tree's leaf size (16) and its power-of-two padding, contiguous and scattered, against the oracle's
    restatement of the reference's algorithm (whichever pipeline the process runs: see the forced children below)"""
    fs, ofs = kz.FFTSettings(12), ko.FFTSettings(12)
    rng = np.random.default_rng(99)
    # (4032 = 64 leaves of 63: the most the reference's own tree handles in a 4096-wide settings object -- one more leaf and its convolutions need
    # 8192 roots of unity, InplaceFFT fails and ZeroPolyViaMultiplication panics)
    for cnt in (1, 2, 15, 16, 17, 31, 33, 255, 1000, 2049, 4032):
        for missing in (list(range(cnt)), sorted(rng.choice(4096, size=cnt, replace=False).tolist())):
            ze, zp = fs.zero_poly_via_multiplication(missing, 4096)
            oze, ozp = ofs.zero_poly_via_multiplication(missing, 4096)
            assert np.array_equal(ze, oze) and np.array_equal(zp, ozp), cnt
    ze, zp = fs.zero_poly_via_multiplication([3, 4, 5, 9], 16)             # a short domain inside a wide settings object (stride 256)
    oze, ozp = ofs.zero_poly_via_multiplication([3, 4, 5, 9], 16)
    assert np.array_equal(ze, oze) and np.array_equal(zp, ozp)
    fs.close()


def test_zero_poly_pipelines_in_fresh_processes():
    """the vanishing polynomial is evaluated directly for small erasure sets and built as a product tree for large ones (chosen by size, once per
    process): the KAT, the ragged sets, every oracle comparison and the full DAS flow re-run with each pipeline forced at every size"""
    import subprocess
    import sys
    if os.environ.get("KZG_HIP_ZERO_POLY"):
        pytest.skip("already a forced child")
    for mode in ("tree", "direct"):
        res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "zero_poly or recover or full_das_flow"],
                             env=dict(os.environ, KZG_HIP_ZERO_POLY=mode), capture_output=True, text=True, timeout=1200)
        assert res.returncode == 0, (mode, res.stdout[-1500:])


@pytest.mark.parametrize("scale,frac", [(5, 2), (8, 2), (10, 3), (12, 2), (15, 2)])
def test_zero_poly_and_recover_match_oracle(kz, scale, frac):
    fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
    n = 1 << scale
    rng = np.random.default_rng(scale)
    missing = sorted(rng.choice(n, size=n // frac, replace=False).tolist())
    ze, zp = fs.zero_poly_via_multiplication(missing, n)
    oze, ozp = ofs.zero_poly_via_multiplication(missing, n)
    assert np.array_equal(ze, oze) and np.array_equal(zp, ozp)
    # recover_from_samples_test.go:62-137: half the coefficients zero, drop the `missing` samples, recover the data
    poly = np.concatenate([rand_fr(rng, n // 2) if n <= 4096 else ko.synthetic_blob(scale, n // 2), ko.fr_empty(n // 2)])
    data = ofs.fft(poly)
    present = np.ones(n, dtype=np.uint8)
    present[missing] = 0
    samples = np.where(present[:, None].astype(bool), data, 0)
    rec = fs.recover_poly_from_samples(samples, present)
    assert np.array_equal(rec, data)
    if scale <= 12:
        assert np.array_equal(rec, ofs.recover_poly_from_samples(samples, present))
    if scale == 5:   # a full-degree polynomial with half the samples missing: whatever the reference's algorithm yields, both agree
        full = np.where(present[:, None].astype(bool), ofs.fft(rand_fr(rng, n)), 0)
        try:
            want = ofs.recover_poly_from_samples(full, present)
        except ko.OracleError:
            with pytest.raises(kz.KzgError, match="failed to reconstruct"):
                fs.recover_poly_from_samples(full, present)
        else:
            assert np.array_equal(fs.recover_poly_from_samples(full, present), want)
    fs.close()


# ------------------------------------------------------------------ end-to-end: the flow of TestFullDAS (integration_test.go:18-159)
def test_full_das_flow(kz):
    """random 31-byte data -> reverse-bit order -> DAS extension -> commitment -> FK20Multi coset proofs (l = 128) ->
    every coset proof verified (pairing-free form of CheckProofMulti) -> up to half of the samples dropped -> recovery
    -> original bytes.  Everything between the byte arrays runs on the device."""
    scale, l = 10, 128
    points = 1 << scale
    rng = np.random.default_rng(1234)
    data = rng.integers(0, 256, size=points * 31, dtype=np.uint8)
    data[:100] = 0
    even_i = [int.from_bytes(data[i * 31:(i + 1) * 31].tobytes() + b"\x00", "little") for i in range(points)]
    even = ko.reverse_bit_order(ko.fr_from_ints(even_i))
    fs = kz.FFTSettings(scale + 1)
    odd = fs.das_fft_extension(even)                                       # integration_test.go:42
    extended = np.empty((2 * points, 4), dtype=np.uint64)
    extended[0::2], extended[1::2] = even, odd
    setup = fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), 2 * points)
    ks = kz.KZGSettings(fs, setup)
    coeffs = fs.fft(extended, inv=True)
    assert not coeffs[points:].any()                                       # the extension property
    commit = ks.commit_to_poly(coeffs[:points])
    coeffs_i = ko.fr_to_ints(coeffs[:points])
    gen = ko.g1_generator()
    assert ko.g1_equal(commit, ko.g1_mul(gen, ko.fr_from_ints([pyref.eval_poly(coeffs_i, S_TEST)])[0]))
    fk = kz.FK20MultiSettings(ks, 2 * points, l)
    proofs = fk.fk20_multi_da_optimized(coeffs)                            # integration_test.go:77 (natural order)
    sample_count = 2 * points // l
    ext_bro = ko.reverse_bit_order(extended)                               # integration_test.go:71
    w = pyref.root_of_unity(scale + 1)
    for i in range(sample_count):                                          # integration_test.go:98-111
        pos = pyref.rev_bits(i, 4)
        x = pow(w, pos, ko.R_MOD)                                          # domainStride = 1
        d = pyref.coset_proof_dlog(coeffs_i, S_TEST, x, l)
        assert ko.g1_equal(proofs[pos], ko.g1_mul(gen, ko.fr_from_ints([d])[0])), i
        sub = pyref.bitrev(ko.fr_to_ints(ext_bro[i * l:(i + 1) * l]))      # sample i really is the coset's evaluations
        wl = pow(w, (2 * points) // l, ko.R_MOD)
        assert sub[:3] == [pyref.eval_poly(coeffs_i, x * pow(wl, j, ko.R_MOD) % ko.R_MOD) for j in range(3)]
    present_samples = np.ones(sample_count, dtype=bool)
    present_samples[rng.choice(sample_count, size=sample_count // 2, replace=False)] = False
    present = np.repeat(present_samples, l)
    partial = np.where(present[:, None], ext_bro, 0)
    # undo the reverse-bit order (integration_test.go:132), recover on the device, redo it
    present_nat = np.array(pyref.bitrev(present.astype(np.uint8).tolist()), dtype=np.uint8)
    recovered = fs.recover_poly_from_samples(ko.reverse_bit_order(partial), present_nat)
    recovered = ko.reverse_bit_order(recovered)
    assert np.array_equal(recovered, ext_bro)
    back = b"".join(v.to_bytes(32, "little")[:31] for v in ko.fr_to_ints(recovered[:points]))
    assert back == data.tobytes()
    fk.close(); ks.close(); fs.close()


def test_device_resident_batch_transforms(kz, setup_1337):
    """the _dev forms bench.py times (inputs / outputs in HBM) equal the host-buffer forms"""
    import torch
    fs = kz.FFTSettings(12)
    L = kz.lib()
    st = torch.cuda.current_stream().cuda_stream
    blobs = np.stack([ko.synthetic_blob(70 + b) for b in range(3)])
    d_in = torch.from_numpy(blobs.view(np.int64)).cuda()
    d_out = torch.empty_like(d_in)
    assert L.kzg_hip_fft_fr_batch_dev(fs.h, d_in.data_ptr(), 4096, 3, 0, d_out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), fs.fft_batch(blobs))
    d_das = torch.from_numpy(blobs[:, :2048].copy().view(np.int64)).cuda()
    assert L.kzg_hip_das_fft_extension_batch_dev(fs.h, d_das.data_ptr(), 2048, 3, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_das.cpu().numpy().view(np.uint64), fs.das_fft_extension_batch(blobs[:, :2048]))
    pts = np.stack([setup_1337[:64], setup_1337[64:128]])
    d_p = torch.from_numpy(pts.view(np.int64).reshape(2, 64, 18)).cuda()
    d_q = torch.empty_like(d_p)
    for inv in (0, 1):
        assert L.kzg_hip_fft_g1_batch_dev(fs.h, d_p.data_ptr(), 64, 2, inv, d_q.data_ptr(), st) == 0
        torch.cuda.synchronize()
        got = d_q.cpu().numpy().view(np.uint64).reshape(2, 64, 3, 6)
        assert np.array_equal(got[0], fs.fft_g1(pts[0], bool(inv))) and np.array_equal(got[1], fs.fft_g1(pts[1], bool(inv)))
    fs.close()
