#!/usr/bin/env python3
"""A loop of LONE DAUsingFK20 calls (2048 coefficients -> 4096 proofs, host buffers) for a kernel / HIP-API trace (see tools/lone_commit_trace.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402

fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
fk = kz.FK20SingleSettings(ks, 4096)
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(4, 1, 4096).reshape(-1, 32))
poly = blob[:2048].copy()
for _ in range(4):
    fk.da_using_fk20(poly)
ts = []
for _ in range(40):
    t0 = time.perf_counter()
    fk.da_using_fk20(poly)
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("DAUsingFK20 alone: median %.3f ms, min %.3f ms, p90 %.3f over 40 calls" % (ts[20], ts[0], ts[36]))
