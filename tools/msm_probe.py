#!/usr/bin/env python3
"""The bucket pipeline (what caller-supplied points take): bls.LinCombG1 on 4096 fresh points alone (host buffers) and device-resident batches on a cached set
whose table budget is 0 (forced onto k_msm_sort / accumulate / reduce / combine).  usage: python tools/msm_probe.py [batches...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402
sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 512]
lib = kz.lib()
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
setup = fs.from_compressed_g1(raw)
B = max(sizes)
blobs, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, B, 4096).reshape(-1, 32))
blobs = blobs.reshape(B, 4096, 4)
ts = []
for i in range(23):
    t0 = time.perf_counter(); fs.lin_comb_g1(setup, blobs[i % B]); ts.append((time.perf_counter() - t0) * 1e3)
out = ["alone (host buffers) %.3f ms" % float(np.median(ts[3:]))]
pts = kz.G1Points(fs, setup)
pts.set_table_budget_gb(0)
d_in = torch.from_numpy(blobs.view(np.int64)).cuda()
d_out = torch.zeros((B, 18), dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ref = None
for bs in sizes:
    reps = 10 if bs < 512 else 4
    for i in range(reps + 2):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        st = lib.kzg_hip_lincomb_points_batch_dev(pts.h, d_in.data_ptr(), 4096, bs, d_out.data_ptr(), s)
        assert st == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out.append("batch %d: %.0f MSM/s (%.3f ms)" % (bs, bs / dt, dt * 1e3))
ks = kz.KZGSettings(fs, setup); ks.set_table_budget_gb(10)
want = ks.commit_to_poly_batch(blobs[:min(B, 64)])
got = d_out[:min(B, 64)].cpu().numpy().view(np.uint64).reshape(-1, 3, 6)
print(os.environ.get("KZG_HIP_MSM_SEG", "default"), "|", " | ".join(out), "| equals the table walk:", bool(np.array_equal(got, want)))
