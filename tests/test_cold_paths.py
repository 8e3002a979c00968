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
                    if i + 2 * off < N:                   # the walk continues after the doubling / from the point at infinity
                        e[i + 2 * off] = int.from_bytes(rng.bytes(32), "little") % R
                    row, dlog = sparse_row(e)
                    out.append(("off=%d i=%d d=%#x sign=%+d" % (off, i, d, sign), row, dlog))
    # the bare pairs: the commitment itself is 2 d S_(i+off) resp. the point at infinity
    for off, i in ((256, 3), (1, 100), (64, 64), (2048, 2047)):
        for sign in (1, -1):
            row, dlog = sparse_row({i: sign * pow(S, off, R) % R, i + off: 1})
            out.append(("bare off=%d i=%d sign=%+d" % (off, i, sign), row, dlog))
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
@pytest.mark.parametrize("budget_gb,want_c", [(210.0, 16), (64.0, 14), (10.0, 11)])
def test_walk_exceptional_cases_crafted_blobs(kz, setup_1337, material, budget_gb, want_c):
    crafted, want_crafted, fillers, want_fill = material
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
