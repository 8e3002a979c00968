#!/usr/bin/env python3
"""DAUsingFK20 (2048 coefficients -> 4096 proofs) on host-buffer batches of a few polynomials: ms per call.  usage: python tools/fk20_batch_probe.py [sizes...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402
sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
fk = kz.FK20SingleSettings(ks, 4096)
polys, _ = fs.fr_from_32(bench.splitmix_blobs_le32(4, max(sizes), 2048).reshape(-1, 32))
polys = polys.reshape(max(sizes), 2048, 4)
out = []
for b in sizes:
    fk.da_using_fk20_batch(polys[:b])
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        fk.da_using_fk20_batch(polys[:b])
    out.append("%d: %.1f ms" % (b, (time.perf_counter() - t0) / reps * 1e3))
print(os.environ.get("KZG_HIP_G1_MUL", "default"), " | ".join(out))
