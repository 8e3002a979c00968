"""Several devices behind ONE handle of the C ABI (kzg_hip_multi_*, go-kzg_amd/csrc/capi_multi.hip; SURVEY.md 8b threading row, 8e).

The test box has one GPU, so the device lists repeat device 0: every entry still owns its settings, tables and stream, the batches
are divided among the entries and the one-polynomial FK20 calls exchange their slices between the entries' buffers (peer copies; the
RCCL leg is exercised with a single-device communicator).  Results must equal the single-device calls bit for bit, and the oracle's
vectors / byte pins (tests/golden/)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import koracle as ko  # noqa: E402  (the checker)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
DERIVED = json.load(open(os.path.join(GOLDEN, "derived_vectors.json")))
FK20_PINS = json.load(open(os.path.join(GOLDEN, "fk20_pins.json")))
S_TEST = int(KATS["test_secret"]["value"])
TEST_POLY = KATS["test_poly"]["values"]


@pytest.fixture(scope="module")
def kz():
    import gokzg_amd
    assert gokzg_amd.device_count() >= 1, "no gfx950 device: the HIP path is the only path"
    return gokzg_amd


@pytest.fixture(scope="module")
def setup_1337():
    raw = np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    return ko.g1_affine(ko.g1_decompress(raw))


def rand_fr(rng, n):
    return ko.fr_from_ints([int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(n)])


def fk20_multi_test_poly(chunk_count=32):
    """the polynomial of fk20_multi_test.go:21-32"""
    poly = []
    for i in range(chunk_count):
        vals = [1, 2, 3, 4 + i, 7, 8 + i * i, 9, 10, 13, 14, 1, 15, 0, 1000, 0, 33]
        vals[12] = ko.R_MOD - 1
        vals[14] = ko.R_MOD - 134
        poly += vals
    return poly


def sha(fs, proofs):
    return hashlib.sha256(fs.to_compressed_g1(proofs).tobytes()).hexdigest()


def test_handle_shape_transport_and_misuse(kz, setup_1337):
    m = kz.MultiKZGSettings([0, 0], 12, setup_1337)
    L = kz.lib()
    assert L.kzg_hip_multi_device_count(m.h) == 2 and L.kzg_hip_multi_device(m.h, 1) == 0 and L.kzg_hip_multi_device(m.h, 2) == -1
    assert m.transport == "peer-copy" and "repeats" in m.transport_note          # RCCL refuses two ranks on one device
    assert m.transport_self_test.startswith("ok: peer-copy, 2 entries") and m.exchanges == 0   # the constructor proved the exchange
    assert L.kzg_hip_multi_kzg(m.h, 0) != L.kzg_hip_multi_kzg(m.h, 1)             # every entry owns its settings
    assert not L.kzg_hip_multi_kzg(m.h, 2)
    m.close()
    with pytest.raises(kz.NoDeviceError):
        kz.MultiKZGSettings([0, kz.device_count()], 12, setup_1337)              # a device that is not there
    with pytest.raises(kz.KzgPanic) as e:
        kz.MultiKZGSettings([], 12, setup_1337)
    assert e.value.status == kz.ERR_BAD_ARG
    with pytest.raises(kz.KzgPanic) as e:
        kz.MultiKZGSettings([0, 0], 13, setup_1337)                               # kzg.go:25-27: setup shorter than the domain
    assert e.value.status == kz.ERR_LEN_MISMATCH


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_batches_are_divided_among_the_entries(kz, setup_1337, devices):
    """CommitToPoly / ComputeProofSingle on ragged batches: entry i takes a contiguous share, results in input order, equal to one device's"""
    m = kz.MultiKZGSettings(devices, 12, setup_1337)
    m.set_table_budget_gb(10)                                                    # c = 11 tables: three of them stay small
    ks = m.kzg_settings(0)                                                        # the same call on one entry is the single-device reference
    rng = np.random.default_rng(len(devices))
    for batch in (1, 2, 37):
        blobs = np.stack([rand_fr(rng, 4096) for _ in range(min(batch, 5))])
        blobs = np.concatenate([blobs] * (batch // blobs.shape[0] + 1))[:batch].copy()
        blobs[-1, :7] = 0
        got = m.commit_to_poly_batch(blobs)
        assert np.array_equal(got, ks.commit_to_poly_batch(blobs)), batch
        xs = np.arange(5, 5 + batch, dtype=np.uint64)
        assert np.array_equal(m.compute_proof_single_batch(blobs, xs), ks.compute_proof_single_batch(blobs, xs)), batch
    # vector F: the first synthetic blob on the s = 1337 setup
    f = m.commit_to_poly_batch(np.stack([ko.synthetic_blob(1), ko.synthetic_blob(2)]))
    assert ko.g1_compress(f[:1])[0].tobytes().hex() == DERIVED["F_blob_seed1"]["commit_monomial_s1337"]
    m.close()


def test_pinned_input_is_read_in_place(kz, setup_1337):
    """kzg_hip_host_register: a batch whose input lies in pinned memory is walked in place over PCIe (no staged copy) -- same bytes as from pageable
    memory, on one device and through the multi-device handle (each entry reads its share of the one registered range); an offset INTO the range works,
    unregistering restores the staged path, misuse is a status code"""
    m = kz.MultiKZGSettings([0, 0], 12, setup_1337)
    m.set_table_budget_gb(10)
    ks = m.kzg_settings(0)
    rng = np.random.default_rng(77)
    blobs = np.stack([rand_fr(rng, 4096) for _ in range(9)])
    want = ks.commit_to_poly_batch(blobs)
    with kz.pinned(blobs):
        assert np.array_equal(ks.commit_to_poly_batch(blobs), want)
        assert np.array_equal(ks.commit_to_poly_batch(blobs[3:]), want[3:])          # a pointer inside the registered range
        assert np.array_equal(m.commit_to_poly_batch(blobs), want)
        assert np.array_equal(ks.commit_to_poly_batch(blobs[:, :1000].copy()), ks.commit_to_poly_batch(np.ascontiguousarray(blobs[:, :1000])))   # pageable beside it
    assert np.array_equal(ks.commit_to_poly_batch(blobs), want)                     # unregistered again: staged copy
    with kz.pinned(blobs[:4]):                                                      # only the first rows are pinned: a batch that starts inside the
        assert np.array_equal(ks.commit_to_poly_batch(blobs), want)                 # range and runs past its end must take the staged copy (in place
        assert np.array_equal(ks.commit_to_poly_batch(blobs[2:]), want[2:])         # it would fault on the device); a batch inside it is still in place
        assert np.array_equal(ks.commit_to_poly_batch(blobs[1:4]), want[1:4])
        assert np.array_equal(m.commit_to_poly_batch(blobs), want)
    L = kz.lib()
    assert L.kzg_hip_host_register(None, 16) == kz.ERR_BAD_ARG and L.kzg_hip_host_unregister(None) == kz.ERR_BAD_ARG
    assert L.kzg_hip_host_unregister(blobs.ctypes.data) == kz.ERR_HIP              # not registered (any more)
    # a live registration cannot be re-registered or overlapped (its tracked extent would be overwritten): refused before the runtime sees it, the first extent stays
    with kz.pinned(blobs[:4]):
        assert L.kzg_hip_host_register(blobs.ctypes.data, blobs.nbytes) == kz.ERR_BAD_ARG and b"overlaps" in L.kzg_hip_last_error()
        assert L.kzg_hip_host_register(blobs[2:].ctypes.data, blobs[2:].nbytes) == kz.ERR_BAD_ARG
        assert np.array_equal(ks.commit_to_poly_batch(blobs), want)                 # still cut at the boundary of the first four rows
        # an unregister that cannot succeed (an address inside the range is not a registered base) is refused and must not forget the extent
        assert L.kzg_hip_host_unregister(blobs[1:].ctypes.data) == kz.ERR_BAD_ARG
        assert np.array_equal(ks.commit_to_poly_batch(blobs), want) and np.array_equal(ks.commit_to_poly_batch(blobs[1:4]), want[1:4])
    m.close()


def test_eth_and_transform_batches_are_divided_among_the_entries(kz, setup_1337):
    """package eth on every entry (BlobToKZGCommitment / ComputeKZGProof on batches incl. an invalid blob and a z inside the domain) and the F_r transform
    batches (FFT, DASFFTExtension) through the multi-device handle: equal to the single-device calls, vector F in its eth form"""
    raw = np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8)
    lag = ko.g1_affine(ko.g1_decompress(raw))
    m = kz.MultiKZGSettings([0, 0, 0], 12, setup_1337)
    me = kz.MultiEthSettings(m, lag)
    fs0 = m.fft_settings(0)
    e0 = kz.EthSettings(fs0, lag)
    rng = np.random.default_rng(12)
    B = 7
    polys = np.stack([ko.synthetic_blob(1 + b) for b in range(B)])
    blobs = fs0.fr_to_32(polys.reshape(-1, 4)).reshape(B, 4096, 32).copy()
    blobs[5, 100] = 0xff                                       # a field element >= r: that blob is refused, the others are not
    got, ok = me.blob_to_kzg_commitment_batch(blobs)
    want, wok = e0.blob_to_kzg_commitment_batch(blobs)
    assert np.array_equal(got, want) and np.array_equal(ok, wok) and not ok[5] and ok[[0, 1, 2, 3, 4, 6]].all()
    assert got[0].tobytes().hex() == DERIVED["F_blob_seed1"]["commit_eth_bitrev_lagrange"]
    with kz.pinned(blobs):                                     # pinned blobs are converted in place over PCIe: same bytes
        gp, okp = me.blob_to_kzg_commitment_batch(blobs)
        assert np.array_equal(gp, want) and np.array_equal(okp, wok)
    zs = rand_fr(rng, B)
    zs[3] = fs0.expanded_roots_of_unity()[5]                   # z in the domain: "invalid z challenge" for that row only
    pg, yg, okg = me.compute_kzg_proof_batch(polys, zs)
    pw, yw, okw = e0.compute_kzg_proof_batch(polys, zs)
    assert np.array_equal(pg, pw) and np.array_equal(yg, yw) and np.array_equal(okg, okw) and not okg[3] and okg[[0, 1, 2, 4, 5, 6]].all()
    half = np.ascontiguousarray(polys[:, :2048])                # "polynomial has invalid length" (eth/helpers.go:186-188)
    out48, okb = np.zeros((B, 48), dtype=np.uint8), np.zeros(B, dtype=np.uint8)
    assert kz.lib().kzg_hip_multi_eth_compute_kzg_proof_batch(me.h, half.ctypes.data, 2048, B, zs.ctypes.data, out48.ctypes.data, None, okb.ctypes.data) == kz.ERR_LEN_MISMATCH
    rows = np.stack([rand_fr(rng, 4096) for _ in range(5)])
    for inv in (False, True):
        assert np.array_equal(m.fft_batch(rows, inv), fs0.fft_batch(rows, inv))
    assert np.array_equal(m.das_fft_extension_batch(rows[:, :2048]), fs0.das_fft_extension_batch(rows[:, :2048]))
    assert np.array_equal(m.fft_batch(rows[:1, :64]), fs0.fft_batch(rows[:1, :64]))     # one row: two entries get nothing
    g1rows = np.stack([setup_1337[64 * b:64 * b + 64] for b in range(5)])               # FFTG1 on batches: host batch form and the multi-device one
    for inv in (False, True):
        want_g1 = np.stack([fs0.fft_g1(r, inv) for r in g1rows])
        assert np.array_equal(fs0.fft_g1_batch(g1rows, inv), want_g1) and np.array_equal(m.fft_g1_batch(g1rows, inv), want_g1)
    with pytest.raises(kz.KzgError):
        fs0.fft_g1_batch(g1rows[:, :48])                                                 # fft_g1.go:63-65: not a power of two
    e0.close(); me.close(); m.close()


@pytest.mark.parametrize("devices,mode,exchanges", [([0, 0], "gather", 1), ([0, 0], "sharded", 5), ([0, 0, 0, 0], None, 5), ([0, 0, 0], "sharded", 1)])
def test_one_polynomial_fk20_vectors_C_and_E(kz, devices, mode, exchanges, monkeypatch):
    """DAUsingFK20 (fk20_single_test.go:12-22, scale 5) and DAUsingFK20Multi (fk20_multi_test.go, scale 10, chunk 16) of ONE polynomial over the
    entries: vectors C and E of SURVEY.md 8c, both exchange schemes; three entries are not a power of two and fall back to one all-gather"""
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "4")
    # vector C: scale 5, 32 proofs (sharded transforms need 2k >= 8 D: just long enough at 4 entries)
    m = kz.MultiKZGSettings(devices, 5, ko.generate_testing_setup_g1(S_TEST, 33))
    m.set_fft_sharding(mode)
    fk = kz.MultiFK20SingleSettings(m, 32)
    poly = ko.fr_from_ints(TEST_POLY)
    e0 = m.exchanges
    proofs = fk.da_using_fk20(poly)
    c = DERIVED["C_da_using_fk20_scale5"]
    assert hashlib.sha256(ko.g1_compress(proofs).tobytes()).hexdigest() == c["sha256"]
    assert m.exchanges - e0 == exchanges
    batch = fk.da_using_fk20_batch(np.stack([poly] * 5))                         # and the batch form: rows divided among the entries
    assert all(np.array_equal(batch[i], proofs) for i in range(5))
    fk.close(); m.close()
    # vector E: scale 10, chunk length 16 -> 64 coset proofs from 2k = 64 positions: sharded transforms at 2 and 4 entries
    n2, l = 1024, 16
    m = kz.MultiKZGSettings(devices, 10, ko.generate_testing_setup_g1(S_TEST, n2))
    m.set_fft_sharding(mode)
    fkm = kz.MultiFK20MultiSettings(m, n2, l)
    poly = ko.fr_from_ints(fk20_multi_test_poly())
    e0 = m.exchanges
    proofs = fkm.da_using_fk20_multi(poly)
    assert m.exchanges - e0 == exchanges
    e = DERIVED["E_da_using_fk20_multi_scale10_l16"]
    assert hashlib.sha256(ko.g1_compress(proofs).tobytes()).hexdigest() == e["sha256"]
    rng = np.random.default_rng(7)
    polys = np.stack([rand_fr(rng, n2 // 2) for _ in range(3)])
    single = kz.FK20MultiSettings(m.kzg_settings(0), n2, l)
    want = np.stack([single.da_using_fk20_multi(p) for p in polys])
    assert np.array_equal(fkm.da_using_fk20_multi_batch(polys), want)
    assert np.array_equal(single.da_using_fk20_multi_batch(polys), want)          # the single-device host batch form added with it
    for p, w in zip(polys, want):
        assert np.array_equal(fkm.da_using_fk20_multi(p), w)
    with pytest.raises(kz.KzgPanic) as ex:
        fkm.da_using_fk20_multi(polys[0][:256])
    assert ex.value.status == kz.ERR_LEN_MISMATCH
    single.close(); fkm.close(); m.close()


@pytest.mark.parametrize("mode", ["gather", "sharded"])
def test_config4a_one_polynomial_over_two_entries_byte_pin(kz, setup_1337, mode):
    """BASELINE config 4a (DAUsingFK20, scale 12, blob(seed 4)[:2048] -> 4096 proofs) through the multi-device handle on [0, 0]: the oracle's byte pin"""
    m = kz.MultiKZGSettings([0, 0], 12, setup_1337)
    m.set_fft_sharding(mode)
    fk = kz.MultiFK20SingleSettings(m, 4096)
    proofs = fk.da_using_fk20(ko.synthetic_blob(4)[:2048])
    assert sha(m.fft_settings(0), proofs) == FK20_PINS["config4a_da_using_fk20_seed4"]["sha256"]
    fk.close(); m.close()


@pytest.mark.parametrize("devices,mode", [([0, 0], "gather"), ([0, 0], "sharded"), ([0] * 8, "sharded")])
def test_config5_one_polynomial_byte_pin(kz, devices, mode, monkeypatch):
    """BASELINE config 5 (DAUsingFK20Multi, scale 16, chunk 16 -> 4096 coset proofs) of ONE polynomial over 2 and 8 entries, Toeplitz stage by
    output position and, in "sharded", both G1 transforms by decimation (five all-gathers): all proofs hash to the oracle's pin"""
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "24" if len(devices) == 2 else "6")
    l, n = 16, 32768
    fs = kz.FFTSettings(16)
    setup = fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), 65536)
    m = kz.MultiKZGSettings(devices, 16, setup)
    m.set_fft_sharding(mode)
    fk = kz.MultiFK20MultiSettings(m, 2 * n, l)
    e0 = m.exchanges
    proofs = fk.da_using_fk20_multi(ko.synthetic_blob(5, n))
    assert m.exchanges - e0 == (5 if mode == "sharded" else 1)
    pin = FK20_PINS["config5_da_using_fk20_multi_seed5"]
    assert sha(fs, proofs) == pin["sha256"]
    fk.close()
    if len(devices) == 2:
        # config 5's second variant (integration_test.go:74): chunk length 128 -> 512 coset proofs, 256 output positions per entry
        fk2 = kz.MultiFK20MultiSettings(m, 2 * n, 128)
        e0 = m.exchanges
        proofs2 = fk2.da_using_fk20_multi(ko.synthetic_blob(5, n))
        assert proofs2.shape[0] == 512 and m.exchanges - e0 == (5 if mode == "sharded" else 1)
        assert sha(fs, proofs2) == FK20_PINS["config5_l128_da_using_fk20_multi_seed5"]["sha256"]
        fk2.close()
    m.close(); fs.close()


def test_rccl_leg_on_a_single_device_communicator(kz, monkeypatch):
    """the RCCL binding itself (librccl bound at run time, ncclCommInitAll, grouped ncclAllGather on ncclUint8, ncclCommDestroy) with the one
    device a test box has: a one-rank all-gather in place, results unchanged"""
    monkeypatch.setenv("KZG_HIP_MULTI_TRANSPORT", "rccl")
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "4")
    m = kz.MultiKZGSettings([0], 5, ko.generate_testing_setup_g1(S_TEST, 33))
    assert m.transport == "rccl", m.transport_note
    fk = kz.MultiFK20SingleSettings(m, 32)
    proofs = fk.da_using_fk20(ko.fr_from_ints(TEST_POLY))
    assert m.exchanges == 1
    assert hashlib.sha256(ko.g1_compress(proofs).tobytes()).hexdigest() == DERIVED["C_da_using_fk20_scale5"]["sha256"]
    fk.close(); m.close()


def _vector_C_through(kz, m):
    """DAUsingFK20 of the reference's test polynomial over the entries of m: the exchange is on the path of every proof"""
    fk = kz.MultiFK20SingleSettings(m, 32)
    m.set_fft_sharding("sharded" if len(m.devices) in (2, 4) else "gather")
    e0 = m.exchanges
    proofs = fk.da_using_fk20(ko.fr_from_ints(TEST_POLY))
    assert hashlib.sha256(ko.g1_compress(proofs).tobytes()).hexdigest() == DERIVED["C_da_using_fk20_scale5"]["sha256"]
    n = m.exchanges - e0
    fk.close()
    return n


@pytest.mark.parametrize("devices,force,fault,transport,why", [
    ([0, 0], None, "peer", "host-staged", "peer-copy failed its self-test (peer copy failed: injected fault"),
    ([0, 0], None, "peer-corrupt", "host-staged", "holds wrong bytes after the all-gather (first at slice 0, offset 0)"),
    ([0, 0, 0, 0], "host", None, "host-staged", "repeats"),
    ([0], "rccl", "rccl", "peer-copy", "rccl failed its self-test (ncclAllGather failed: injected fault"),
    ([0], "rccl", "rccl-corrupt", "peer-copy", "rccl failed its self-test (self-test: entry 0 holds wrong bytes"),
    ([0], "rccl", "rccl,peer", "host-staged", "peer-copy failed its self-test"),
    # a transport that HANGS instead of failing (a kernel spinning on a flag nobody sets sits in front of the exchange): the probe's deadline fires, the communicator
    # is aborted resp. the stuck streams and arenas are abandoned, and the next transport is proven on fresh ones
    ([0], "rccl", "rccl-hang", "peer-copy", "rccl failed its self-test (self-test: the all-gather did not complete within 400 ms (KZG_HIP_MULTI_PROBE_TIMEOUT_MS): timeout)"),
    # ... and RCCL blocking on the HOST side: the probe's ncclAllGather group runs on a helper thread against the same deadline; ncclCommInitAll against six of them
    ([0], "rccl", "rccl-block", "peer-copy", "rccl failed its self-test (self-test: the RCCL calls did not return within 400 ms (KZG_HIP_MULTI_PROBE_TIMEOUT_MS): timeout on the host side)"),
    ([0], "rccl", "rccl-init-block", "peer-copy", "ncclCommInitAll did not return within 2400 ms: timeout"),
    ([0, 0], None, "peer-hang", "host-staged", "peer-copy failed its self-test (self-test: the all-gather did not complete within 400 ms (KZG_HIP_MULTI_PROBE_TIMEOUT_MS): timeout); streams and arenas of the hung exchange abandoned"),
])
def test_transport_self_test_steps_down_on_an_injected_fault(kz, monkeypatch, devices, force, fault, transport, why):
    """kzg_hip_multi_settings_new proves its exchange before returning (pattern -> all-gather -> every byte verified on every entry) and replaces
    a transport that errs or delivers wrong bytes: rccl -> peer-copy -> host-staged.  KZG_HIP_MULTI_FAULT makes the named leg fail on the one
    GPU of a test box; the reason lands in transport_note and the proofs that then travel over the fall-back are still vector C."""
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "4")
    if force:
        monkeypatch.setenv("KZG_HIP_MULTI_TRANSPORT", force)
    if fault:
        monkeypatch.setenv("KZG_HIP_MULTI_FAULT", fault)
    hang = bool(fault) and ("hang" in fault or "block" in fault)
    if hang:
        monkeypatch.setenv("KZG_HIP_MULTI_PROBE_TIMEOUT_MS", "400")
    import time
    setup33 = ko.generate_testing_setup_g1(S_TEST, 33)
    t0 = time.time()
    m = kz.MultiKZGSettings(devices, 5, setup33)
    if hang:
        # the constructor returns within a few deadlines (probe + abort / drain + the next probes), not when the spinning kernel's own clock runs out (2 x deadline + 0.5 s
        # AFTER which a blocking synchronise would have returned) and never "never"
        assert time.time() - t0 < 10.0, time.time() - t0
        assert "timeout" in m.transport_note
        if fault in ("rccl-hang", "rccl-block"):
            assert "communicators aborted" in m.transport_note or "ncclCommAbort" in m.transport_note, m.transport_note
    assert m.transport == transport, (m.transport, m.transport_note)
    assert why in m.transport_note, m.transport_note
    assert m.transport_self_test.startswith("ok: %s, %d entries" % (transport, len(devices))), m.transport_self_test
    n = _vector_C_through(kz, m)
    assert n == (0 if len(devices) == 1 else 5)          # one entry without RCCL: the single-device call, nothing to exchange
    m.close()


def test_a_transport_that_stays_stuck_costs_bounded_time(kz, monkeypatch):
    """`peer-stuck`: the injected hang is NOT released after its streams were abandoned (a copy that never completes).  Fresh streams may share a hardware queue with
    the stuck one -- then the next probes time out too.  Either way the constructor must come back within a few deadlines: with host-staged proven, or with an error
    whose text names the timeouts.  Nothing may block for good."""
    import time
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "4")
    monkeypatch.setenv("KZG_HIP_MULTI_FAULT", "peer-stuck")
    monkeypatch.setenv("KZG_HIP_MULTI_PROBE_TIMEOUT_MS", "300")
    setup33 = ko.generate_testing_setup_g1(S_TEST, 33)
    t0 = time.time()
    try:
        m = kz.MultiKZGSettings([0, 0, 0, 0], 5, setup33)
    except kz.KzgPanic as e:
        assert time.time() - t0 < 10.0
        assert "timeout" in str(e) and "peer-copy failed its self-test" in str(e), str(e)
        return
    assert time.time() - t0 < 10.0
    assert m.transport == "host-staged" and "timeout" in m.transport_note, (m.transport, m.transport_note)
    _vector_C_through(kz, m)
    m.close()


def test_forced_host_staged_and_out_of_range_ordinal(kz, monkeypatch):
    """host-staged as the FIRST choice (KZG_HIP_MULTI_TRANSPORT=host) never touches the broken peer leg; ordinals are bounded by what the runtime
    enumerates (hipGetDeviceCount), not by the number of gfx950 devices"""
    monkeypatch.setenv("KZG_HIP_MULTI_FAULT", "peer")
    monkeypatch.setenv("KZG_HIP_MULTI_TRANSPORT", "host")
    m = kz.MultiKZGSettings([0, 0], 5, ko.generate_testing_setup_g1(S_TEST, 33))
    assert m.transport == "host-staged" and m.transport_self_test.startswith("ok: host-staged")
    m.close()
    n_ord = kz.lib().kzg_hip_device_count()
    with pytest.raises(kz.NoDeviceError):
        kz.MultiKZGSettings([0, n_ord + 7], 5, ko.generate_testing_setup_g1(S_TEST, 33))


def test_batch_calls_from_many_threads_share_the_workers(kz, setup_1337):
    """the per-entry worker threads of a handle serve concurrent batch calls from several host threads (one queue per entry): every caller gets
    its own rows back, equal to the single-device results"""
    import threading
    m = kz.MultiKZGSettings([0, 0, 0], 12, setup_1337)
    m.set_table_budget_gb(10)
    ks = m.kzg_settings(0)
    rng = np.random.default_rng(5)
    blobs = [np.stack([rand_fr(rng, 4096) for _ in range(3 + t)]) for t in range(6)]
    want = [ks.commit_to_poly_batch(b) for b in blobs]
    got, errs = [None] * 6, []

    def run(t):
        try:
            for _ in range(3):
                got[t] = m.commit_to_poly_batch(blobs[t])
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    m.close()


def _distinct_devices(kz):
    return list(range(kz.device_count()))


@pytest.mark.parametrize("force", [None, "peer", "host"])
@pytest.mark.parametrize("mode", ["gather", "sharded"])
def test_distinct_devices_byte_pins(kz, monkeypatch, force, mode):
    """What a one-GPU box cannot run: the exchange between DISTINCT devices -- ncclAllGather over xGMI (default), hipMemcpyPeerAsync ("peer"),
    host-staged -- under vector C, vector E and the config-5 byte pin.  Skipped below two devices; the first multi-GPU box that runs the suite runs it."""
    devs = _distinct_devices(kz)
    if len(devs) < 2:
        pytest.skip("needs two or more gfx950 devices")
    devs = devs[:1 << (len(devs).bit_length() - 1)]        # a power of two: the sharded transforms
    monkeypatch.setenv("KZG_HIP_FK20_FB_BUDGET_GB", "8")
    if force:
        monkeypatch.setenv("KZG_HIP_MULTI_TRANSPORT", force)
    m = kz.MultiKZGSettings(devs, 5, ko.generate_testing_setup_g1(S_TEST, 33))
    assert m.transport == {None: "rccl", "peer": "peer-copy", "host": "host-staged"}[force], m.transport_note
    assert m.transport_self_test.startswith("ok: "), m.transport_self_test
    if len(devs) <= 4:
        _vector_C_through(kz, m)
    m.close()
    l, n = 16, 32768
    fs = kz.FFTSettings(16)
    setup = fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), 65536)
    m = kz.MultiKZGSettings(devs, 16, setup)
    m.set_fft_sharding(mode)
    fk = kz.MultiFK20MultiSettings(m, 2 * n, l)
    e0 = m.exchanges
    proofs = fk.da_using_fk20_multi(ko.synthetic_blob(5, n))
    assert m.exchanges - e0 == (5 if mode == "sharded" else 1)
    assert sha(fs, proofs) == FK20_PINS["config5_da_using_fk20_multi_seed5"]["sha256"]
    fk.close()
    fk2 = kz.MultiFK20MultiSettings(m, 2 * n, 128)                                  # the l = 128 variant: 512 coset proofs
    assert sha(fs, fk2.da_using_fk20_multi(ko.synthetic_blob(5, n))) == FK20_PINS["config5_l128_da_using_fk20_multi_seed5"]["sha256"]
    fk2.close(); m.close(); fs.close()
