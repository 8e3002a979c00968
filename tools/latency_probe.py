#!/usr/bin/env python3
"""Single-call latencies of the host-buffer entry points (one blob per call), for the table in DESIGN.md 5."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402


def timeit(fn, reps):
    for _ in range(4):                       # the first calls after an idle period run at a lower clock (a lone DAUsingFK20: 17 ms, then 11.3)
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
setup = fs.from_compressed_g1(raw)
ks = kz.KZGSettings(fs, setup)
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 1, 4096).reshape(-1, 32))
print("CommitToPoly(4096)            %.3f ms" % timeit(lambda: ks.commit_to_poly(blob), 50))
print("ComputeProofSingle(4096)      %.3f ms" % timeit(lambda: ks.compute_proof_single(blob, 17), 50))
print("FFT_Fr(4096)                  %.3f ms" % timeit(lambda: fs.fft(blob, False), 100))
print("DASFFTExtension(2048)         %.3f ms" % timeit(lambda: fs.das_fft_extension(blob[:2048].copy()), 100))
print("FFTG1(4096)                   %.3f ms" % timeit(lambda: fs.fft_g1(setup, False), 5))
print("LinCombG1(4096 caller points) %.3f ms" % timeit(lambda: fs.lin_comb_g1(setup, blob), 20))
cached = kz.G1Points(fs, setup)
print("LinCombG1(4096 cached points) %.3f ms" % timeit(lambda: cached.lin_comb(blob), 20))
blobs64, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 64, 4096).reshape(-1, 32))
blobs64 = blobs64.reshape(64, 4096, 4)
t = timeit(lambda: cached.lin_comb_batch(blobs64), 5)
print("LinCombG1 batch 64 (cached)   %.3f ms = %.0f MSM/s (host buffers)" % (t, 64 / t * 1e3))
fk = kz.FK20SingleSettings(ks, 4096)
print("DAUsingFK20(2048 -> 4096)     %.3f ms" % timeit(lambda: fk.da_using_fk20(blob[:2048].copy()), 5))
