#!/usr/bin/env python3
"""Fuzz of kzg_hip_lincomb_g1's promotion of repeated point sets: a random schedule of calls over a few point sets (repeats, in-place mutations of one point, prefixes,
fresh sets that push old ones out of the handle's memory), every result against the oracle.  usage: python tools/fuzz_promote.py [calls [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402
from oracle import koracle as ko  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 66)
fs = kz.FFTSettings(4)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
base = ko.g1_decompress(raw)
n = 128
sets = [base[k * n:(k + 1) * n].copy() for k in range(10)]
bad = 0
for c in range(calls):
    k = int(rng.integers(0, 4)) if rng.random() < 0.8 else int(rng.integers(4, 10))      # four hot sets, six cold ones
    pts = sets[k]
    what = rng.random()
    if what < 0.1:
        pts[int(rng.integers(0, n))] = base[int(rng.integers(0, 4096))]                  # in-place change of one point of a (possibly promoted) set
    m = n if what > 0.2 else int(rng.integers(64, n + 1))                                # sometimes a prefix
    sc = ko.fr_from_ints([int.from_bytes(rng.bytes(32), "little") % ko.R_MOD for _ in range(m)])
    got = fs.lin_comb_g1(pts[:m], sc)
    want = ko.g1_affine(ko.lincomb_g1(pts[:m], sc))[0]
    bad += int(not np.array_equal(got, want))
print("fuzz_promote: %d calls, %d mismatches, (promoted sets, calls served by one) = %s" % (calls, bad, fs.lincomb_promotions()))
sys.exit(1 if bad else 0)
