#!/usr/bin/env python3
"""Lone DAUsingFK20Multi (scale 16, chunk 16) through the one-polynomial entry point, the host batch form with batch 1 and the multi-device handle with one entry: ms per call (median).
usage: python tools/fkm_lone_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402


def med(fn, reps=7, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


fs = kz.FFTSettings(16)
sec = np.frombuffer((bench.S_TEST * ((1 << 256) % bench.R_MOD) % bench.R_MOD).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4)
setup = fs.generate_testing_setup_g1(sec, 65536)
poly, _ = fs.fr_from_32(bench.splitmix_blobs_le32(5, 1, 32768).reshape(-1, 32))
m = kz.MultiKZGSettings([0], 16, setup)
mf = kz.MultiFK20MultiSettings(m, 65536, 16)
fk = kz.FK20MultiSettings(m.kzg_settings(0), 65536, 16)
a = fk.da_using_fk20_multi(poly)
assert np.array_equal(a, mf.da_using_fk20_multi(poly)) and np.array_equal(a, fk.da_using_fk20_multi_batch(poly.reshape(1, -1, 4))[0])
print("coalescing", os.environ.get("KZG_HIP_COALESCE", "default"),
      "| one-polynomial entry %.2f ms | host batch of 1 %.2f ms | multi handle, one entry %.2f ms" % (
          med(lambda: fk.da_using_fk20_multi(poly)), med(lambda: fk.da_using_fk20_multi_batch(poly.reshape(1, -1, 4))), med(lambda: mf.da_using_fk20_multi(poly))))
