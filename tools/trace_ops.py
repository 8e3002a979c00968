#!/usr/bin/env python3
"""runs a few single-call operations so that `rocprofv3 --kernel-trace` shows their kernel timelines:
   tools/trace_ops.py lincomb|lincomb_cached|fftg1|fk20|commit"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gokzg_amd as kz

op = sys.argv[1]
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(bench.ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
setup = fs.from_compressed_g1(raw)
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 1, 4096).reshape(-1, 32))
if op == "lincomb":
    for _ in range(3): fs.lin_comb_g1(setup, blob)
elif op == "lincomb_cached":
    c = kz.G1Points(fs, setup)
    for _ in range(3): c.lin_comb(blob)
elif op == "fftg1":
    for _ in range(2): fs.fft_g1(setup, False)
elif op in ("fk20", "commit"):
    ks = kz.KZGSettings(fs, setup)
    ks.set_table_budget_gb(10)
    if op == "commit":
        for _ in range(3): ks.commit_to_poly(blob)
    else:
        os.environ.setdefault("KZG_HIP_FK20_FB_BUDGET_GB", "8")
        fk = kz.FK20SingleSettings(ks, 4096)
        for _ in range(2): fk.da_using_fk20(blob[:2048].copy())
