#!/usr/bin/env python3
"""Randomised cross-check of the G1 transform paths (direct passes on 1 / 2 / 4 lanes, stage network on 1 / 2 / 4 lanes, ragged FK20 batches) against
the oracle and against one-at-a-time calls.  Not part of the suite; usage: python tools/fuzz_g1.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402
from oracle import koracle as ko  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
S = 1927409816240961209460912649124
gen = ko.g1_generator()
bad = 0
base = ko.generate_testing_setup_g1(S, 512)


def rand_fr(n):
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


for c in range(cases):
    scale = int(rng.integers(1, 10))
    n = 1 << int(rng.integers(0, scale + 1))
    fs, ofs = kz.FFTSettings(scale), ko.FFTSettings(scale)
    vals = np.stack([base[int(i)] for i in rng.integers(0, 512, size=n)])
    for k in range(min(n, 3)):
        i = int(rng.integers(0, n))
        vals[i] = [ko.g1_zero()[0], vals[(i + 1) % n], ko.g1_sub(ko.g1_zero()[0], vals[(i + 1) % n])][k]
    inv = bool(rng.integers(0, 2))
    got = fs.fft_g1(vals, inv)
    want = ofs.fft_g1(vals, inv)
    if ko.g1_compress(got).tobytes() != ko.g1_compress(want).tobytes():
        bad += 1; print("FFTG1 mismatch", scale, n, inv)
    fs.close()
# ragged FK20 batches against one-at-a-time calls (scale 8: 128 coefficients -> 256 proofs)
fs = kz.FFTSettings(8)
setup = fs.generate_testing_setup_g1(fs.fr_from_32(np.frombuffer(S.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))[0][0], 256)
ks = kz.KZGSettings(fs, setup)
fk = kz.FK20SingleSettings(ks, 256)
polys = rand_fr(70 * 128).reshape(70, 128, 4)
singles = {}
for nb in [int(x) for x in rng.choice(np.arange(1, 71), size=12, replace=False)]:
    got = fk.da_using_fk20_batch(polys[:nb])
    for b in {0, nb // 2, nb - 1}:
        if b not in singles:
            singles[b] = fk.da_using_fk20(polys[b])
        if not np.array_equal(got[b], singles[b]):
            bad += 1; print("FK20 batch mismatch", nb, b)
fk.close(); ks.close(); fs.close()
print("cases", cases, "mismatches", bad, {k: os.environ.get(k) for k in ("KZG_HIP_G1_QUAD", "KZG_HIP_G1_FFT", "KZG_HIP_G1_MUL", "KZG_HIP_G1_DIRECT_COOP")})
sys.exit(1 if bad else 0)
