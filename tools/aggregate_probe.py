#!/usr/bin/env python3
"""eth.ComputeAggregateKZGProof on blocks of a few blobs: ms per call from host buffers (transcript hashed on the host while the device commits)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8)
eth = kz.EthSettings(fs, fs.from_compressed_g1(raw))
sizes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 64]
blobs = bench.splitmix_blobs_le32(9, max(sizes), 4096)
out = []
for b in sizes:
    eth.compute_aggregate_kzg_proof(blobs[:b])
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        eth.compute_aggregate_kzg_proof(blobs[:b])
    out.append("%d blobs: %.2f ms" % (b, (time.perf_counter() - t0) / reps * 1e3))
print(os.environ.get("KZG_HIP_SHA256", "sha-ni if present"), " | ".join(out))
