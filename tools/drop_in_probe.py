#!/usr/bin/env python3
"""drop-in throughput: T host threads each calling the reference-shaped ONE-blob entry points (host buffers) concurrently;
prints commitments/s, the batch-resident rate beside it and single-call latencies.  `python tools/drop_in_probe.py [threads]`"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gokzg_amd as kz


def run(ks, blobs, T, per_thread):
    start = threading.Barrier(T + 1)
    def work(i):
        start.wait()
        for r in range(per_thread):
            ks.commit_to_poly(blobs[(i + r) % len(blobs)])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in ts]
    start.wait()
    t0 = time.perf_counter()
    [t.join() for t in ts]
    return T * per_thread / (time.perf_counter() - t0)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    fs = kz.FFTSettings(12)
    raw = np.frombuffer(open(os.path.join(bench.ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    ks = kz.KZGSettings(fs, fs.from_compressed_g1(raw))
    ks.set_table_budget_gb(float(os.environ.get("TABLE_GB", "110")))
    std = bench.splitmix_blobs_le32(1, 64)
    blobs, ok = fs.fr_from_32(std.reshape(-1, 32))
    blobs = blobs.reshape(64, 4096, 4)
    ks.commit_to_poly(blobs[0])
    for t in ((T,) if os.environ.get("ONLY") else (1, 8, 32, T, 2 * T, 4 * T)):
        ks.bench_drop_in(blobs, t, 4)
        rate, out = ks.bench_drop_in(blobs, t, 100)
        print("native threads %3d: %8.0f commitments/s" % (t, rate))
    if os.environ.get("ONLY"):
        ks.close(); fs.close()
        return
    print("python threads %3d: %8.0f commitments/s (GIL-bound harness)" % (T, run(ks, blobs, T, 20)))
    prate, _ = ks.bench_drop_in(blobs, T, 50, op=1)
    print("native threads %3d: %8.0f single proofs/s" % (T, prate))
    t0 = time.perf_counter()
    for _ in range(100):
        ks.commit_to_poly(blobs[1])
    print("single-call latency %.3f ms" % ((time.perf_counter() - t0) * 10))
    ks.close(); fs.close()


if __name__ == "__main__":
    main()
