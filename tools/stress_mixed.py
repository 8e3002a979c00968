#!/usr/bin/env python3
"""Mixed-call stress of the host side added in round 6 (stream-ordered block cache, promotion of repeated point sets, packed eth result rows): T threads make random
one-polynomial calls (CommitToPoly, ComputeProofSingle, bls.LinCombG1 on the same caller-supplied points, eth.ComputeKZGProof) for a while, every result compared with the
one computed single-threaded beforehand; then handles are created, used and freed in a loop (the cache's blocks leave with their streams).
usage: python tools/stress_mixed.py [threads [seconds [seed]]]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gokzg_amd as kz  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 6
golden = os.path.join(ROOT, "tests", "golden")
fs = kz.FFTSettings(12)
setup = fs.from_compressed_g1(np.frombuffer(open(os.path.join(golden, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8))
lag = fs.from_compressed_g1(np.frombuffer(open(os.path.join(golden, "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
ks = kz.KZGSettings(fs, setup)
ks.set_table_budget_gb(9)
eth = kz.EthSettings(fs, lag)
NB = 12
blobs, _ = fs.fr_from_32(bench.splitmix_blobs_le32(seed, NB, 4096).reshape(-1, 32))
blobs = blobs.reshape(NB, 4096, 4)
z, _ = fs.fr_from_32(np.frombuffer((0x1234567890abcdef1234567890abcdef).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
pts = np.ascontiguousarray(setup[:512])
want = {"c": [ks.commit_to_poly(b) for b in blobs], "p": [ks.compute_proof_single(b, 17 + i) for i, b in enumerate(blobs)],
        "l": [fs.lin_comb_g1(pts, b[:512]) for b in blobs], "e": [eth.compute_kzg_proof(b, z) for b in blobs]}
bad, calls = [0], [0]
stop = time.time() + secs


def worker(t):
    rng = np.random.default_rng(1000 * seed + t)
    n = 0
    while time.time() < stop:
        i = int(rng.integers(0, NB)); op = "cple"[int(rng.integers(0, 4))]
        if op == "c":
            ok = np.array_equal(ks.commit_to_poly(blobs[i]), want["c"][i])
        elif op == "p":
            ok = np.array_equal(ks.compute_proof_single(blobs[i], 17 + i), want["p"][i])
        elif op == "l":
            ok = np.array_equal(fs.lin_comb_g1(pts, blobs[i][:512]), want["l"][i])
        else:
            g = eth.compute_kzg_proof(blobs[i], z)
            ok = np.array_equal(g[0], want["e"][i][0]) and np.array_equal(g[1], want["e"][i][1])
        n += 1
        if not ok:
            bad[0] += 1
    calls[0] += n


ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print("stress_mixed: %d threads, %d calls in %.0f s, %d mismatches, promotions %s" % (T, calls[0], secs, bad[0], fs.lincomb_promotions()))
# handles come and go: every cycle creates the streams / coalescers of a settings object, uses them, frees them
cyc_bad = 0
for c in range(12):
    f2 = kz.FFTSettings(12)
    k2 = kz.KZGSettings(f2, setup)
    k2.set_table_budget_gb(5)
    e2 = kz.EthSettings(f2, lag)
    i = c % NB
    cyc_bad += not np.array_equal(k2.commit_to_poly(blobs[i]), want["c"][i])
    cyc_bad += not np.array_equal(k2.compute_proof_single(blobs[i], 17 + i), want["p"][i])
    g = e2.compute_kzg_proof(blobs[i], z)
    cyc_bad += not (np.array_equal(g[0], want["e"][i][0]) and np.array_equal(g[1], want["e"][i][1]))
    cyc_bad += not np.array_equal(f2.fft(blobs[i], False), fs.fft(blobs[i], False))
    e2.close(); k2.close(); f2.close()
print("handle cycles: 12, mismatches %d" % cyc_bad)
sys.exit(1 if bad[0] or cyc_bad else 0)
