#!/usr/bin/env python3
"""Commitment rate through the blocking HOST-buffer entry point (kzg_hip_commit_to_poly_batch: upload, walk, download per call),
for the PCIe-inclusive figure quoted in DESIGN.md 5."""
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
ks.set_table_budget_gb(float(os.environ.get("TABLE_GB", "110")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blobs, ok = fs.fr_from_32(bench.splitmix_blobs_le32(1, B, 4096).reshape(-1, 32))
blobs = blobs.reshape(B, 4096, 4)
ks.commit_to_poly_batch(blobs)          # builds the table
reps = 10
t0 = time.time()
for _ in range(reps):
    out = ks.commit_to_poly_batch(blobs)
dt = (time.time() - t0) / reps
print("host-buffer commit_to_poly_batch: %.2f ms per %d blobs = %.0f commitments/s (pageable host memory)" % (dt * 1e3, B, B / dt))
with kz.pinned(blobs):                   # kzg_hip_host_register: the walk reads the coefficients in place over PCIe
    same = np.array_equal(ks.commit_to_poly_batch(blobs), out)
    t0 = time.time()
    for _ in range(reps):
        ks.commit_to_poly_batch(blobs)
    dt = (time.time() - t0) / reps
print("host-buffer commit_to_poly_batch: %.2f ms per %d blobs = %.0f commitments/s (input pinned by kzg_hip_host_register; same results: %s)" % (dt * 1e3, B, B / dt, same))
