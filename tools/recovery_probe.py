import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
import gokzg_amd as kz
for scale in ([int(a) for a in sys.argv[1:]] or (12, 16)):
    n = 1 << scale
    fs = kz.FFTSettings(scale)
    poly, _ = fs.fr_from_32(bench.splitmix_blobs_le32(3, 1, n).reshape(-1, 32))
    poly[n // 2:] = 0
    data = fs.fft(poly, False)
    rng = np.random.default_rng(1)
    present = np.ones(n, dtype=np.uint8); present[rng.permutation(n)[: n // 2]] = 0
    samples = data.copy(); samples[present == 0] = 0
    missing = np.nonzero(present == 0)[0]
    fs.zero_poly_via_multiplication(missing, n)
    t0 = time.time(); fs.zero_poly_via_multiplication(missing, n); t1 = time.time()
    out = fs.recover_poly_from_samples(samples, present); t2 = time.time()
    out = fs.recover_poly_from_samples(samples, present); t3 = time.time()
    print("scale %d: zero_poly %.2f ms, recover %.2f ms, exact %s" % (scale, (t1 - t0) * 1e3, (t3 - t2) * 1e3, bool(np.array_equal(out, data))))
    fs.close()
