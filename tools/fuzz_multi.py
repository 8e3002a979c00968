#!/usr/bin/env python3
"""Randomised cross-check of the multi-device handle (kzg_hip_multi_*) on lists that repeat device 0: 1 ... 8 entries, scales 5 ... 11, chunk lengths 1 ... 32,
both exchange schemes and the default policy, batches of ragged sizes -- against the single-device calls on entry 0.  Not part of the suite.
usage: python tools/fuzz_multi.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KZG_HIP_FK20_FB_BUDGET_GB", "1")
os.environ.setdefault("KZG_HIP_FB_BUDGET_GB", "2")
import gokzg_amd as kz  # noqa: E402
from oracle import koracle as ko  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
S = 1927409816240961209460912649124
fs0 = kz.FFTSettings(11)
full = fs0.generate_testing_setup_g1(ko.fr_from_ints([S]), 2048)


def rand_fr(n):
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


bad = 0
for c in range(cases):
    D = int(rng.integers(1, 9)); scale = int(rng.integers(5, 12)); n2 = 1 << scale; n = n2 // 2
    l = 1 << int(rng.integers(0, min(6, scale - 2)))
    mode = [None, "gather", "sharded"][int(rng.integers(0, 3))]
    m = kz.MultiKZGSettings([0] * D, scale, full[:n2 + 1] if n2 < 2048 else full)
    m.set_fft_sharding(mode)
    ks0 = m.kzg_settings(0)
    B = int(rng.integers(1, 3 * D + 2))
    polys = rand_fr(B * n).reshape(B, n, 4)
    polys[B - 1, n // 2:] = 0
    if not np.array_equal(m.commit_to_poly_batch(polys), ks0.commit_to_poly_batch(polys)):
        bad += 1; print("commit batch", D, scale, B)
    mfk, fk = kz.MultiFK20MultiSettings(m, n2, l), kz.FK20MultiSettings(ks0, n2, l)
    want = np.stack([fk.da_using_fk20_multi(p) for p in polys])
    if not np.array_equal(mfk.da_using_fk20_multi_batch(polys), want):
        bad += 1; print("fk20 multi batch", D, scale, l, B)
    for b in {0, B - 1}:
        if not np.array_equal(mfk.da_using_fk20_multi(polys[b]), want[b]):
            bad += 1; print("fk20 multi one polynomial", D, scale, l, mode, b)
    if l == 1:
        mfs, fks = kz.MultiFK20SingleSettings(m, n2), kz.FK20SingleSettings(ks0, n2)
        if not np.array_equal(mfs.da_using_fk20(polys[0]), fks.da_using_fk20(polys[0])):
            bad += 1; print("fk20 single one polynomial", D, scale, mode)
        mfs.close(); fks.close()
    fk.close(); mfk.close(); m.close()
print("fuzz_multi: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
