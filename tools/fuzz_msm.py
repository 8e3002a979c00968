#!/usr/bin/env python3
"""Randomised cross-check of the bucket MSM (bls.LinCombG1 on caller-supplied points and on cached sets with the table budget at 0): point counts 1 ... 5000,
batches on both sides of the balanced-accumulate threshold, scalar distributions that stress the sort and the segment logic (uniform, tiny, one scalar
repeated, a few distinct scalars, sparse windows, lambda multiples), infinities / repeated / opposite points -- against the oracle's MultiExp (small
cases) and against the fixed-base walk of the same points (all cases).  Not part of the suite; run with KZG_HIP_MSM_SEG=0 / 1 forced as well.
usage: python tools/fuzz_msm.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402
from oracle import koracle as ko  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
R = ko.R_MOD
LAMBDA = 0xac45a4010001a40200000000ffffffff
fs = kz.FFTSettings(13)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
base = ko.g1_affine(ko.g1_decompress(raw))
more = fs.mul_g1_vec(base[:1024], ko.fr_from_ints([int(x) for x in rng.integers(2, 1 << 62, size=1024)]))
base = np.concatenate([base, more])                            # 5120 distinct points


def scalars(kind, n):
    if kind == 0:
        v = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    elif kind == 1:
        v = [int(x) for x in rng.integers(0, 300, size=n)]
    elif kind == 2:
        v = [int.from_bytes(rng.bytes(32), "little") % R] * n
    elif kind == 3:
        pool = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(3)] + [0, 1, R - 1]
        v = [pool[int(i)] for i in rng.integers(0, len(pool), size=n)]
    elif kind == 4:
        v = [(int(rng.integers(1, 256)) << (8 * int(rng.integers(0, 31)))) % R for _ in range(n)]
    else:
        v = [(LAMBDA * int(rng.integers(0, 1 << 20)) + int(rng.integers(0, 3))) % R for _ in range(n)]
    return ko.fr_from_ints(v)


bad = 0
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 300, 1000, 2047, 4096, 5000])) if c % 2 else int(rng.integers(1, 5001))
    B = int(rng.choice([1, 2, 7, 33, 64, 70, 96]))
    if n * B > 400000:
        B = max(1, 400000 // n)
    pts = base[rng.permutation(5120)[:n]].copy()
    for k in range(min(n // 4, 3)):
        i = int(rng.integers(0, n))
        pts[i] = [ko.g1_zero()[0], pts[(i + 1) % n], ko.g1_sub(ko.g1_zero()[0], pts[(i + 1) % n])][k]
    rows = np.stack([scalars(int(rng.integers(0, 6)), n) for _ in range(min(B, 6))])
    rows = np.concatenate([rows] * (B // rows.shape[0] + 1))[:B].copy()
    cached = kz.G1Points(fs, pts)
    want = cached.lin_comb_batch(rows) if n >= 64 else None    # the set's own fixed-base table (sets below 64 points have none)
    cached.set_table_budget_gb(0)
    got = cached.lin_comb_batch(rows)
    if want is not None and not np.array_equal(got, want):
        bad += 1; print("bucket pipeline != table walk", n, B)
    one = fs.lin_comb_g1(pts, rows[0])                         # caller-supplied points, one shot
    if not np.array_equal(one, got[0]):
        bad += 1; print("one-shot LinCombG1 != cached set", n, B)
    if n <= 1000:
        b = int(rng.integers(0, B))
        if ko.g1_compress(got[b:b + 1]).tobytes() != ko.g1_compress(ko.lincomb_g1(pts, rows[b])[None]).tobytes():
            bad += 1; print("oracle mismatch", n, B, b)
    cached.close()
print("fuzz_msm: %d cases, %d mismatches (KZG_HIP_MSM_SEG=%s)" % (cases, bad, os.environ.get("KZG_HIP_MSM_SEG", "default")))
sys.exit(1 if bad else 0)
