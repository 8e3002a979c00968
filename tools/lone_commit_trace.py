#!/usr/bin/env python3
"""A loop of LONE CommitToPoly / ComputeProofSingle calls (host buffers, one 4096-coefficient blob per call) for a kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o lone -- python tools/lone_commit_trace.py
tools/rocprof_summary.py then gives the per-kernel times of ONE call (every kernel is launched once per call)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402

fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 1, 4096).reshape(-1, 32))
which = sys.argv[1] if len(sys.argv) > 1 else "commit"
fn = (lambda: ks.commit_to_poly(blob)) if which == "commit" else (lambda: ks.compute_proof_single(blob, 17))
for _ in range(5):
    fn()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    fn()
    ts.append((time.perf_counter() - t0) * 1e3)
print("%s alone: median %.3f ms, min %.3f ms over 200 calls" % (which, float(np.median(ts)), min(ts)))
