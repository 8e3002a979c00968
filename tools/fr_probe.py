#!/usr/bin/env python3
"""F_r transform rates at scale 12, device-resident batches (what bench.py's reference_benchmarks times), plus a bit comparison of the
radix-4 kernel against the radix-2 one (KZG_HIP_FR_FFT=radix2 in a child process).  usage: python tools/fr_probe.py [batch]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gokzg_amd as kz

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = kz.lib()
fs = kz.FFTSettings(12)
rng = np.random.default_rng(7)
R = 52435875175126190479447740508185965837690552500527637822603658699938581184513
raw = rng.integers(0, 2**63, size=(B * 4096, 4), dtype=np.uint64)
raw[:, 3] &= np.uint64((1 << 62) - 1)                       # < r: valid Montgomery images (any value < r is one)
d_in = torch.from_numpy(raw.view(np.int64).reshape(B, 4096, 4)).cuda()
d_out = torch.empty_like(d_in)
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

res = {}
for inv in (0, 1):
    dt = timeit(lambda: lib.kzg_hip_fft_fr_batch_dev(fs.h, d_in.data_ptr(), 4096, B, inv, d_out.data_ptr(), stream))
    res["fft_fr inv=%d" % inv] = B / dt
d_das = d_in[:, :2048, :].contiguous()
dt = timeit(lambda: lib.kzg_hip_das_fft_extension_batch_dev(fs.h, d_das.data_ptr(), 2048, B, stream))
res["das_ext"] = B / dt
tag = os.environ.get("KZG_HIP_FR_FFT", "radix4")
for k, v in res.items():
    print("%s  batch %d  %-14s %.3f M/s" % (tag, B, k, v / 1e6))
lib.kzg_hip_fft_fr_batch_dev(fs.h, d_in.data_ptr(), 4096, B, 0, d_out.data_ptr(), stream)
torch.cuda.synchronize()
h = d_out.cpu().numpy()
lib.kzg_hip_fft_fr_batch_dev(fs.h, d_in.data_ptr(), 4096, B, 1, d_out.data_ptr(), stream)
torch.cuda.synchronize()
hi = d_out.cpu().numpy()
import hashlib
print(tag, "sha256 fwd", hashlib.sha256(h.tobytes()).hexdigest()[:16], "inv", hashlib.sha256(hi.tobytes()).hexdigest()[:16])
if "KZG_HIP_FR_FFT" not in os.environ and not os.environ.get("KZG_FR_PROBE_NO_AB"):
    for form in ("radix2",):          # the radix-2 family (bit comparison); the 256-lane form of the 4096-point kernel has its own harness since round 6: tools/ab_fr_r16/
        subprocess.call([sys.executable, __file__, str(B)], env=dict(os.environ, KZG_HIP_FR_FFT=form))
