#!/usr/bin/env python3
"""Randomised cross-check of the F_r entry points against the oracle: transform sizes / widths / batches / padding, the DAS extension, the vanishing
polynomial with random erasure counts.  Not part of the suite (minutes of oracle time); usage: python tools/fuzz_fr.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402
from oracle import koracle as ko  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rand_fr(n):
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


bad = 0
for c in range(cases):
    max_scale = int(rng.integers(2, 17))
    fs, ofs = kz.FFTSettings(max_scale), ko.FFTSettings(max_scale)
    logn = int(rng.integers(0, max_scale + 1))
    n = 1 << logn
    batch = int(rng.choice([1, 2, 3, 7, 33, 257, max(1, (1 << 20) // n + 1)]))
    if batch * n > (1 << 21):
        batch = max(1, (1 << 21) // n)
    inv = bool(rng.integers(0, 2))
    rows = rand_fr(batch * n).reshape(batch, n, 4)
    got = fs.fft_batch(rows, inv=inv)
    for b in {0, batch // 2, batch - 1}:
        if not np.array_equal(got[b], ofs.fft(rows[b], inv)):
            bad += 1; print("FFT mismatch", max_scale, n, batch, inv, b)
    if n > 2:
        short = rows[0, : int(rng.integers(n // 2 + 1, n))]
        if not np.array_equal(fs.fft(short, inv), ofs.fft(short, inv)):
            bad += 1; print("padded FFT mismatch", max_scale, n, len(short), inv)
    if max_scale >= 2:
        m = 1 << int(rng.integers(1, max_scale))
        db = int(rng.choice([1, 3, max(1, (1 << 20) // m + 1)]))
        d = rand_fr(db * m).reshape(db, m, 4)
        gd = fs.das_fft_extension_batch(d.copy())
        for b in {0, db - 1}:
            if not np.array_equal(gd[b], ofs.das_fft_extension(d[b].copy())):
                bad += 1; print("DAS mismatch", max_scale, m, db, b)
    if max_scale >= 3 and max_scale <= 14:
        length = 1 << int(rng.integers(3, max_scale + 1))
        cnt = int(rng.integers(1, min(length - 1, 63 * (length // 64) if length >= 128 else length - 1) + 1))
        miss = sorted(rng.choice(length, size=cnt, replace=False).tolist())
        ze, zp = fs.zero_poly_via_multiplication(miss, length)
        oze, ozp = ofs.zero_poly_via_multiplication(miss, length)
        if not (np.array_equal(ze, oze) and np.array_equal(zp, ozp)):
            bad += 1; print("zero poly mismatch", max_scale, length, cnt)
    fs.close()
print("cases", cases, "mismatches", bad, "mode", os.environ.get("KZG_HIP_FR_FFT"), os.environ.get("KZG_HIP_ZERO_POLY"))
sys.exit(1 if bad else 0)
