#!/usr/bin/env python3
"""Time of the commitment step alone (kzg_hip_commit_to_poly_batch_dev on device-resident blobs): ms per step, median of the timed steps.
For A/B builds selected with KZG_HIP_LIB -- including timing-only variants whose results are wrong on purpose (bench.py would refuse them).
usage: python tools/walk_probe.py [batch [table_gb [steps]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 210.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib = kz.lib()
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
ks.set_table_budget_gb(gb)
blobs, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, B, 4096).reshape(-1, 32))
d_in = torch.from_numpy(blobs.reshape(B, 4096, 4).view(np.int64)).cuda()
d_out = torch.zeros((B, 18), dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ts = []
for i in range(steps + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lib.kzg_hip_commit_to_poly_batch_dev(ks.h, d_in.data_ptr(), 4096, B, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    assert st == 0, st
    if i >= 3:
        ts.append((time.perf_counter() - t0) * 1e3)
c, w, b = ks.table_info()
print("%s | batch %d, c = %d (%.0f GB): median %.3f ms, min %.3f ms -> %.0f commitments/s" % (os.environ.get("KZG_HIP_LIB", "stock"), B, c, b / 1e9, float(np.median(ts)), min(ts), B / np.median(ts) * 1e3))
