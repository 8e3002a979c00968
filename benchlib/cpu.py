"""cpu_baseline: the oracle's restatement of bls.LinCombG1 timed on the host (kind "port": Go and the Kilic module are absent), one core and all usable
cores, calibrated against the numbers BENCH.md publishes.  The ONLY part of the bench that touches oracle/."""
import shutil
import subprocess
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from .workload import N_COEFF, R_MOD, S_TEST  # noqa: F401


def _cpu_worker(seconds_budget):
    """one host core: as many oracle LinCombG1(4096) as fit the budget; returns (count, seconds)"""
    from oracle import koracle as ko
    raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    setup = ko.g1_decompress(raw)
    blobs = [ko.synthetic_blob(1 + b) for b in range(4)]
    ko.lincomb_g1(setup, blobs[0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_budget:
        ko.lincomb_g1(setup, blobs[n % 4])
        n += 1
    return n, time.perf_counter() - t0


def _port_vs_published(min_s=1.0):
    """the oracle port timed on the three transforms the reference PUBLISHES numbers for (BENCH.md, Kilic column, Ryzen 9 5950X, 1 thread),
    so that a reader can rescale the port's commitments/s: ratio = port time / published time (> 1: the port is slower than Go + Kilic's
    assembly on that CPU).  FFT over G1 is estimated from the oracle's scalar multiplication: 12 x 2048 butterflies, each one MulG1
    (fft_g1.go:44-55 multiplies every butterfly) -- timing a whole transform would take a minute of the bench."""
    from oracle import koracle as ko
    fs = ko.FFTSettings(12)
    blob = ko.synthetic_blob(12)

    def per_call(fn, min_s=min_s):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min_s:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n
    t_fft = per_call(lambda: fs.fft(blob))
    half = blob[:2048].copy()
    t_das = per_call(lambda: fs.das_fft_extension(half.copy()))
    gen = ko.g1_generator()
    ks_ = [ko.fr_from_ints([int.from_bytes(os.urandom(32), "little") % R_MOD])[0] for _ in range(8)]
    it = iter(range(1 << 30))
    t_mul = per_call(lambda: ko.g1_mul(gen, ks_[next(it) % 8]), 1.5 * min_s)
    pub = {"fft_fr_scale12_ns": 1911871, "das_fft_extension_scale12_ns": 1169011, "fft_g1_scale12_ns": 3745748396}
    mine = {"fft_fr_scale12_ns": t_fft * 1e9, "das_fft_extension_scale12_ns": t_das * 1e9, "fft_g1_scale12_ns": 12 * 2048 * t_mul * 1e9}
    return {"port_ns": mine, "published_ns": pub, "port_over_published": {k: mine[k] / pub[k] for k in pub},
            "sources": "BENCH.md:43 (FFT over F_r), :31 (DAS FFT extension), :55 (FFT over G1), scale 12",
            "mul_g1_port_us": t_mul * 1e6, "fft_g1_is_estimate": "12 x 2048 x MulG1 of the port (additions not counted)"}


def fft_g1_muls(m, full_width_only=False):
    """bls.MulG1 calls of one reference FFTG1 of size m >= 4 (fft_g1.go:33-56: naive 4-point leaves, 16 products each, then one product per butterfly).
    full_width_only: leave out the products by roots[0] = 1 (7 of a leaf's 16, one butterfly per sub-transform), which a double-and-add MulG1 finishes at once."""
    lg = m.bit_length() - 1
    if not full_width_only:
        return (m // 4) * 16 + (lg - 2) * (m // 2)
    return (m // 4) * 9 + sum(m // 2 - m // (1 << k) for k in range(3, lg + 1))


def fk20_single_muls(n_coeff):
    """bls.MulG1 calls of one FK20Single on n_coeff coefficients (fk20_single.go:122-134): the Toeplitz product over 2 n points (:63-70), FFTG1(2 n, inv), FFTG1(n)"""
    return 2 * n_coeff + (fft_g1_muls(2 * n_coeff) + 2 * n_coeff) + fft_g1_muls(n_coeff)   # the inverse transform scales every output by 1 / len (fft_g1.go:85-93)


def fk20_single_full_width_muls(n_coeff):
    return 2 * n_coeff + (fft_g1_muls(2 * n_coeff, True) + 2 * n_coeff) + fft_g1_muls(n_coeff, True)


def cpu_baseline_fk20(n_sample=512, full=False):
    """The FK20 half of the metric on the host: the oracle's FK20Single (kind 'port'; restates kzg.go:43-64, fk20_single.go:122-134), ONE thread.
    A whole 4096-coefficient run is ~1 minute of port time (131 072 MulG1 + 8192-point settings), too long for the default bench: the bounded sample is ONE
    FK20Single on `n_sample` coefficients (scale log2(2 n_sample), GenerateTestingSetup with the test secret), and the 4096-coefficient figure is that time scaled by
    the reference's MulG1 count (the run is > 99 % MulG1) -- labelled an estimate.  KZG_BENCH_FK20_CPU_FULL=1 (or full=True) times the real thing instead."""
    from oracle import koracle as ko

    def one(n):
        scale = (2 * n).bit_length() - 1
        ks = ko.KZGSettings(ko.FFTSettings(scale), ko.generate_testing_setup_g1(S_TEST, 2 * n))
        fk = ko.FK20SingleSettings(ks, 2 * n)
        blob = ko.synthetic_blob(1, n)
        t0 = time.perf_counter()
        fk.fk20_single(blob)
        return time.perf_counter() - t0
    full = full or os.environ.get("KZG_BENCH_FK20_CPU_FULL") == "1"
    n = N_COEFF if full else n_sample
    dt = one(n)
    muls_n, muls_full = fk20_single_muls(n), fk20_single_muls(N_COEFF)
    est = dt * fk20_single_full_width_muls(N_COEFF) / fk20_single_full_width_muls(n)    # the products by roots[0] = 1 cost nothing in a double-and-add MulG1
    return {"value": 1.0 / est, "unit": "FK20Single all-proofs/s on 4096-element blobs", "cores": 1, "kind": "port",
            "is_estimate": not full, "seconds_per_fk20_4096": est,
            "sample": ("1 x FK20Single(n = %d coefficients) of the oracle in %.2f s" % (n, dt)) + ("" if full else
                      ", scaled by the reference's count of full-width MulG1 %d / %d (all MulG1: %d / %d; fk20_single.go:122-134, fft_g1.go:33-56)" % (
                          fk20_single_full_width_muls(N_COEFF), fk20_single_full_width_muls(n), muls_full, muls_n)),
            "mul_g1_in_sample": muls_n, "mul_g1_per_fk20_4096": muls_full, "us_per_mul_g1_in_sample": dt / muls_n * 1e6}


def cpu_baseline(seconds_budget=6.0):
    """oracle (kind 'port'): Kilic-style bls.LinCombG1 on 4096 points.  `value` = ONE thread (the reference is single-threaded);
    `all_cores` = one blob per core on every host core (the metric is a per-second throughput).  Must run before the
    process initialises HIP (the all-cores leg forks)."""
    import multiprocessing as mp
    nproc = os.cpu_count() or 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    try:   # a container's CPU quota (cgroup v2) bounds what "all cores" can mean here
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.999)))
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    go = shutil.which("go")
    go_version = None
    if go:
        try:
            go_version = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
        except (OSError, subprocess.SubprocessError):
            go_version = "present, `go version` failed"
    n1, dt1 = _cpu_worker(seconds_budget)
    try:
        calib = _port_vs_published(min(1.0, seconds_budget / 6.0))
    except Exception as e:                                  # noqa: BLE001
        calib = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        fk20_cpu = cpu_baseline_fk20(512 if seconds_budget >= 6.0 else 64)
    except Exception as e:                                  # noqa: BLE001
        fk20_cpu = {"error": "%s: %s" % (type(e).__name__, e)}
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [seconds_budget] * cores)
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    return {"value": n1 / dt1, "unit": "commitments/s", "cores": 1, "kind": "port",
            "sample": "%d x LinCombG1(n=4096) in %.1f s, oracle/kzg_oracle.c (Kilic-style Pippenger c=9), 1 thread" % (n1, dt1),
            "cpu_model": model, "nproc": nproc, "usable_cores": cores,
            "all_cores": {"value": sum(r[0] / r[1] for r in res), "unit": "commitments/s", "cores": cores,
                          "sample": "%d x LinCombG1(n=4096), one blob per core on %d processes, %.1f s wall" % (total, cores, wall)},
            "go_toolchain": go_version or "absent (`go`: command not found): the Go/Kilic reference cannot be timed on this host (BASELINE.md 3)",
            "port_vs_published": calib, "fk20_4096": fk20_cpu,
            "reference_published": "BENCH.md Kilic column, Ryzen 9 5950X, 1 thread: see reference_benchmarks and port_vs_published"}
