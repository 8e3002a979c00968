import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import bench
import gokzg_amd as kz
fs = kz.FFTSettings(12)
raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
setup = fs.from_compressed_g1(raw)
blob, _ = fs.fr_from_32(bench.splitmix_blobs_le32(1, 1, 4096).reshape(-1, 32))
for _ in range(3):
    fs.lin_comb_g1(setup, blob)
