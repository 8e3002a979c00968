#!/usr/bin/env python3
"""DAUsingFK20 (2048 coefficients -> 4096 proofs), ONE polynomial per call from T Python threads (ctypes releases the GIL for the tens of milliseconds a call
blocks): what the coalescer makes of the reference's API shape for FK20 (fk20_single.go:176-196).  usage: python tools/fk20_threads_probe.py [threads [calls per thread]]
A/B the batching policy with KZG_HIP_COALESCE_PER_BATCH / _EXEC."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
fk = kz.FK20SingleSettings(ks, 4096)
polys, _ = fs.fr_from_32(bench.splitmix_blobs_le32(4, 64, 2048).reshape(-1, 32))
polys = polys.reshape(64, 2048, 4)
for _ in range(4):
    fk.da_using_fk20(polys[0])
for rep in range(2):
    gate = threading.Barrier(T + 1)
    def work(i):
        gate.wait()
        for r in range(per):
            fk.da_using_fk20(polys[(i + r) % 64])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in ths]
    gate.wait()
    t0 = time.perf_counter()
    [t.join() for t in ths]
    print("per_batch=%s exec=%s | %d threads x %d calls: %.0f DAUsingFK20/s" % (os.environ.get("KZG_HIP_COALESCE_PER_BATCH", "default"), os.environ.get("KZG_HIP_COALESCE_EXEC", "default"),
                                                                          T, per, T * per / (time.perf_counter() - t0)))
fk.close(); ks.close(); fs.close()
