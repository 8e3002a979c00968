#!/usr/bin/env python3
"""A loop of LONE eth.ComputeKZGProof calls (host buffers) for a kernel trace (see tools/lone_commit_trace.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402

fs = kz.FFTSettings(12)
lag = fs.from_compressed_g1(np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
eth = kz.EthSettings(fs, lag)
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 1, 4096).reshape(-1, 32))
z, _ = fs.fr_from_32(np.frombuffer((0x1234567890abcdef1234567890abcdef).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
fn = lambda: eth.compute_kzg_proof(blob, z)
for _ in range(5):
    fn()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    fn()
    ts.append((time.perf_counter() - t0) * 1e3)
print("eth.ComputeKZGProof alone: median %.3f ms, min %.3f ms over 200 calls" % (float(np.median(ts)), min(ts)))
ts.sort()
print("percentiles (ms): p5 %.3f p25 %.3f p50 %.3f p75 %.3f p95 %.3f" % (ts[10], ts[50], ts[100], ts[150], ts[190]))
