#!/usr/bin/env python3
"""Runs a few batched FFT_G1 (scale 12) on the device and prints the batch rate; used under rocprofv3 for per-stage kernel times."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402

GB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = kz.lib()
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
setup = fs.from_compressed_g1(raw)
d_g1 = torch.from_numpy(setup.view(np.int64).reshape(1, 4096, 18)).cuda().repeat(GB, 1, 1).contiguous()
d_out = torch.empty_like(d_g1)
stream = torch.cuda.current_stream().cuda_stream
for i in range(reps + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.time()
    st = lib.kzg_hip_fft_g1_batch_dev(fs.h, d_g1.data_ptr(), 4096, GB, 0, d_out.data_ptr(), stream)
    assert st == 0
torch.cuda.synchronize()
print("fft_g1 scale 12 batch %d: %.1f /s" % (GB, GB * reps / (time.time() - t0)))
