"""The exceptional cases of the group law INSIDE the headline kernel (k_fb_accumulate: the fixed-base table walk behind
KZGSettings.CommitToPoly) and its reduction trees, forced with crafted blobs.

bls.AddG1 (bls/bls_kilic.go:47-53) is a complete addition.  The walk uses incomplete lazy XYZZ formulas on its straight line and hands P = +-Q
to out-of-line generic code, which random scalars never reach.  With the known secret of eth/trusted_setup.json (s = 1337: S_i = [1337^i]G) a
blob can be built whose partial sums collide:

    k_i = +-d * 1337^off,  k_(i+off) = d        =>   k_i S_i = +-d S_(i+off)

* d a single window digit (1, or a digit at a window boundary of every table in use): a lane that walks point i and then point i + off holds
  exactly the table entry it adds next (doubling) or its negative (the sum is the point at infinity, and the walk continues from there);
* other offsets put the two equal / opposite partial sums in two lanes of one wavefront column, two wavefronts of a workgroup, two
  workgroups of a blob (the cooperative trees of k_fb_accumulate and k_fb_finish / k_fb_finish_lanes), whatever the launch shape.

Since round 5 the walk splits every scalar, k = +-|k1| +- |k2| lambda (glv_split_signed), and a lane sums the phi halves of its points un-mapped, maps
the sum (X <- beta X) and continues with the plain halves.  Rows for that order: pairs whose PHI halves collide before the map (k = lambda m for small
m splits into (0, m)), pairs where a phi-only lane meets a mixed lane with the same sum, and rows built backwards from the kernel's iteration order
(lane t of the unsplit walk owns points t + 256 j; its LAST addition is the top non-zero window of the plain half of its last point): the scalars of a
lane are chosen so that everything summed before that addition equals the entry it adds (doubling) or its negative (infinity), for the window sizes
of every table in use and for the plain layout (KZG_HIP_FB_GLV=0), which test_plain_layout_in_a_fresh_process re-runs.

Expected values do not come from any MSM code: the commitment of a blob is [sum k_i 1337^i mod r]G, one scalar multiplication of the
oracle.  All-zero rows sit at the first, middle and last position of every batch.  Runs at the table sizes bench.py uses (signed 11-, 14-
and 16-bit windows) and at batch sizes 1 / 16 / 32 / 130 / 4096 (window-split walk, plain walk with 16 / 3 / 1 workgroups per blob).
"""
import os

import numpy as np
import pytest

from oracle import koracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R = ko.R_MOD
S = 1337
N = 4096
LAMBDA = 0xac45a4010001a40200000000ffffffff
GLV_WALK = os.environ.get("KZG_HIP_FB_GLV") != "0"


def glv_split(k):
    """glv_split_signed (go-kzg_amd/csrc/g1.hpp) on Python integers: (|k1|, k2, neg1, neg2)"""
    sg = k > (R - 1) // 2
    a = R - k if sg else k
    q = (a + LAMBDA // 2) // LAMBDA
    k1 = a - q * LAMBDA
    return abs(k1), q, sg ^ (k1 < 0), sg


def signed_digits(mag, c, nwin):
    """the walk's recoding: digits in [-2^(c-1), 2^(c-1)] with sum d_w 2^(c w) == mag"""
    out, carry = [], 0
    for w in range(nwin):
        raw = ((mag >> (c * w)) & ((1 << c) - 1)) + carry
        if raw > (1 << (c - 1)):
            out.append(raw - (1 << c)); carry = 1
        else:
            out.append(raw); carry = 0
    assert carry == 0 and sum(d << (c * w) for w, d in enumerate(out)) == mag
    return out


def plain_windows(c):
    """fb_windows (capi_kzg.hip): signed c-bit windows of a canonical scalar"""
    nw = (255 + c - 1) // c
    top = ((R - 1) >> (c * (nw - 1))) + 1
    return nw + 1 if top > (1 << (c - 1)) else nw


def last_addition_rows(rng):
    """rows built backwards from the unsplit walk's order at one workgroup per blob (batches >= 512): lane t owns the points t + 256 j, j = 0 .. 15; with
    the endomorphism it sums their phi halves, maps, then their plain halves, windows ascending -- so its last addition is the top non-zero window of the
    plain half (of the whole scalar in the plain layout) of point t + 3840.  k_(t) is solved so that the lane's sum before that addition equals the entry
    (doubling) or its negative (the lane ends at infinity).  In other launch shapes these are ordinary rows."""
    out = []
    for glv, cs in ((True, (16, 15, 12)), (False, (16, 14, 11))):
        for c in cs:
            for t, sign in ((5, 1), (77, -1), (255, 1)):
                pts = [t + 256 * j for j in range(16)]
                ks = {p: int.from_bytes(rng.bytes(32), "little") % R for p in pts}
                last = pts[-1]
                if glv:
                    k1, _, neg1, _ = glv_split(ks[last])
                    dig = signed_digits(k1, c, (128 + c - 1) // c)
                    flip = -1 if neg1 else 1
                else:
                    dig = signed_digits(ks[last], c, plain_windows(c))
                    flip = 1
                w = max(i for i, d in enumerate(dig) if d)
                e = flip * dig[w] << (c * w)                               # the last entry added is e S_last
                total = (2 * e if sign > 0 else 0) * pow(S, last, R) % R    # what the lane must sum to: e + e, or -e + e
                rest = sum(ks[p] * pow(S, p, R) for p in pts[1:]) % R
                ks[pts[0]] = (total - rest) * pow(S, -pts[0], R) % R
                row, dlog = sparse_row(ks)
                assert dlog == total
                out.append(("last addition of lane %d %s (%s layout, c = %d)" % (t, "doubles" if sign > 0 else "cancels", "glv" if glv else "plain", c), row, dlog))
    return out


@pytest.fixture(scope="module")
def kz():
    import gokzg_amd
    assert gokzg_amd.device_count() >= 1, "no gfx950 device: the HIP path is the only path"
    return gokzg_amd


@pytest.fixture(scope="module")
def setup_1337():
    raw = np.frombuffer(open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    return ko.g1_decompress(raw)


def sparse_row(entries):
    """entries: {index: int} -> (4096 Montgomery images, dlog of the commitment)"""
    row = np.zeros((N, 4), dtype=np.uint64)
    idx = sorted(entries)
    row[idx] = ko.fr_from_ints([entries[i] % R for i in idx])
    return row, sum(entries[i] * pow(S, i, R) for i in idx) % R


def crafted_rows():
    """(description, row, dlog) for every collision pattern"""
    rng = np.random.default_rng(1337)
    out = []
    digits = (1, 1 << 154, 1 << 176, 3 << 224)          # window 0 of every table; window 14 / 11 (c = 11 / 14), 16 / 11 (c = 11 / 16), 16 / 14 (c = 14 / 16)
    for off in (1, 32, 64, 128, 256, 768, 2048):
        for i in (0, 37, 255):
            for d in digits:
                for sign in (1, -1):
                    e = {i: sign * d * pow(S, off, R) % R, i + off: d}
                    if i + 2 * off < N:                   # the walk continues after the doubling / from the point at infinity (below 2^125: no phi half,
                        e[i + 2 * off] = int.from_bytes(rng.bytes(15), "little")   # which the GLV walk would sum BEFORE the pair meets)
                    row, dlog = sparse_row(e)
                    out.append(("off=%d i=%d d=%#x sign=%+d" % (off, i, d, sign), row, dlog))
    # the bare pairs: the commitment itself is 2 d S_(i+off) resp. the point at infinity
    for off, i in ((256, 3), (1, 100), (64, 64), (2048, 2047)):
        for sign in (1, -1):
            row, dlog = sparse_row({i: sign * pow(S, off, R) % R, i + off: 1})
            out.append(("bare off=%d i=%d sign=%+d" % (off, i, sign), row, dlog))
    # the same pairs in the PHI halves: k = lambda m (m < 2^125) splits into (0, m), so both entries are summed un-mapped and meet before / at the map
    for off in (1, 64, 256, 2048):
        for i in (0, 255):
            for m in (1, 1 << 100):
                for sign in (1, -1):
                    e = {i: sign * LAMBDA * m * pow(S, off, R) % R, i + off: LAMBDA * m % R}   # a mixed lane (both halves) against a phi-only lane
                    row, dlog = sparse_row(e)
                    out.append(("phi off=%d i=%d m=%#x sign=%+d" % (off, i, m, sign), row, dlog))
    for c, w in ((16, 0), (16, 7), (15, 3), (12, 0), (12, 9)):                                  # single digits on both sides: 1337 d and d in window w, phi halves only
        for sign in (1, -1):
            row, dlog = sparse_row({40: sign * LAMBDA * (S << (c * w)) % R, 41: LAMBDA * (1 << (c * w)) % R})
            out.append(("phi digits c=%d w=%d sign=%+d" % (c, w, sign), row, dlog))
    out += last_addition_rows(rng)
    # three equal partial sums in lanes of one column, and a full workgroup of them
    row, dlog = sparse_row({i: pow(S, 192 - i, R) for i in (0, 64, 128, 192)})
    out.append(("four equal sums, one per wavefront of a workgroup", row, dlog))
    row, dlog = sparse_row({i: pow(S, 255 - i, R) for i in range(256)})
    out.append(("256 equal sums", row, dlog))
    row, dlog = sparse_row({i: (1 if i % 2 else -1) * pow(S, 255 - i, R) % R for i in range(256)})
    out.append(("256 sums cancelling in pairs", row, dlog))
    return out


def want_point(dlog):
    return ko.g1_mul(ko.g1_generator(), ko.fr_from_ints([dlog])[0]) if dlog else ko.g1_zero(1)


def run_batches(ks, crafted, want_crafted, fillers, want_fill, batch):
    """every crafted row at least once in batches of `batch` rows with all-zero rows first, in the middle and last"""
    zero_at = {0, batch // 2, batch - 1} if batch >= 3 else set()
    inf = ko.g1_affine(ko.g1_zero(1)).reshape(3, 6)
    pool = [("crafted", j) for j in range(len(crafted))] + [("filler", j) for j in range(len(fillers))]
    cursor = 0
    calls = 0
    while cursor < len(crafted):
        rows = np.zeros((batch, N, 4), dtype=np.uint64)
        expect = []
        for b in range(batch):
            if b in zero_at:
                expect.append(("zero row", inf))
                continue
            kind, j = pool[cursor % len(pool)]
            cursor += 1
            if kind == "crafted":
                rows[b] = crafted[j][1]
                expect.append((crafted[j][0], want_crafted[j]))
            else:
                rows[b] = fillers[j]
                expect.append(("filler %d" % j, want_fill[j]))
        got = ks.commit_to_poly_batch(rows) if batch > 1 else ks.commit_to_poly(rows[0]).reshape(1, 3, 6)
        got = np.asarray(got).reshape(batch, 3, 6)
        for b in range(batch):
            assert np.array_equal(got[b], expect[b][1]), "batch %d, row %d (%s), call %d" % (batch, b, expect[b][0], calls)
        calls += 1
    if batch == 1:                                           # a lone all-zero blob
        z = np.asarray(ks.commit_to_poly(np.zeros((N, 4), dtype=np.uint64))).reshape(3, 6)
        assert np.array_equal(z, inf)


@pytest.fixture(scope="module")
def material(setup_1337):
    crafted = crafted_rows()
    want_crafted = [ko.g1_affine(want_point(d)).reshape(3, 6) for _, _, d in crafted]
    # dense filler rows: 4 random blobs and small multiples of them (one oracle MSM + one scalar multiplication each)
    base = [ko.synthetic_blob(900 + b) for b in range(4)]
    base_c = [ko.lincomb_g1(setup_1337, b) for b in base]
    fillers, want_fill = [], []
    for m in range(1, 17):
        for b in range(4):
            ints = ko.fr_to_ints(base[b])
            fillers.append(ko.fr_from_ints([v * m % R for v in ints]))
            want_fill.append(ko.g1_affine(ko.g1_mul(base_c[b], ko.fr_from_ints([m])[0])).reshape(3, 6))
    return crafted, want_crafted, fillers, want_fill


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("budget_gb,want_c,want_c_plain", [(210.0, 16, 16), (64.0, 15, 14), (10.0, 12, 11)])
def test_walk_exceptional_cases_crafted_blobs(kz, setup_1337, material, budget_gb, want_c, want_c_plain):
    crafted, want_crafted, fillers, want_fill = material
    want_c = want_c if GLV_WALK else want_c_plain
    fs = kz.FFTSettings(12)
    ks = kz.KZGSettings(fs, setup_1337)
    try:
        ks.set_table_budget_gb(budget_gb)
        ks.commit_to_poly(fillers[0])
        assert ks.table_info()[0] == want_c
        for batch in (1, 16, 32, 130, 4096):
            run_batches(ks, crafted, want_crafted, fillers, want_fill, batch)
    finally:
        ks.close()
        fs.close()


def test_crafted_rows_really_collide():
    """host-side sanity of the construction (CPU, oracle only): the two partial sums of a pair row are equal or opposite points"""
    g = ko.g1_generator()
    off, i, d = 256, 37, 3 << 224
    a = ko.g1_mul(g, ko.fr_from_ints([d * pow(S, off, R) % R * pow(S, i, R) % R])[0])
    b = ko.g1_mul(g, ko.fr_from_ints([d * pow(S, i + off, R) % R])[0])
    assert ko.g1_equal(a, b)
    assert ko.g1_equal(ko.g1_add(a, ko.g1_mul(b, ko.fr_from_ints([R - 1])[0])), ko.g1_zero(1))


@pytest.mark.gpu
@pytest.mark.timeout(2400)
def test_plain_layout_in_a_fresh_process():
    """KZG_HIP_FB_GLV=0: the table layout of rounds 1-4 (one window per c bits of the whole scalar: k_fb_accumulate and the <false> instantiations of the FK20
    Toeplitz kernels) stays selectable for A/B runs; the crafted blobs, the table-size tests and the FK20 vectors / byte pins run against it in a child process
    (the library reads the variable once)"""
    import subprocess
    import sys
    if not GLV_WALK:
        pytest.skip("already the plain-layout child")
    here = os.path.dirname(os.path.abspath(__file__))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_cold_paths.py"), os.path.join(here, "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k",
                          "crafted_blobs or every_table_size or table_budget_setter or c16_table or vector_F or batch_shapes or vector_C or vectors_D_E or config4a or "
                          "file_accumulation or config5 or config4b or da_using_fk20_batch_host"],
                         env=dict(os.environ, KZG_HIP_FB_GLV="0"), capture_output=True, text=True, timeout=2300)
    assert res.returncode == 0, res.stdout[-2000:]


def test_python_split_matches_the_composition():
    """CPU: the Python restatement of the split used to build the rows: k == +-|k1| +- k2 lambda mod r, both halves below 2^126.5"""
    rng = np.random.default_rng(3)
    for _ in range(2000):
        k = int.from_bytes(rng.bytes(32), "little") % R
        k1, k2, n1, n2 = glv_split(k)
        assert ((-k1 if n1 else k1) + (-k2 if n2 else k2) * LAMBDA) % R == k and k1 < 2 ** 126.5 and k2 < 2 ** 126.5
    assert glv_split(LAMBDA * 12345)[:2] == (0, 12345) and glv_split(R - LAMBDA * 12345) == (0, 12345, True, True)
    assert len(last_addition_rows(np.random.default_rng(1))) == 18
