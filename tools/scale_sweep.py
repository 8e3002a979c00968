#!/usr/bin/env python3
"""The reference's published benchmarks (BENCH.md: FFT, FFTExtension, FFTG1, RecoverPolyFromSamples, ZeroPolyViaMultiplication at scales 4-15,
Kilic backend, one Ryzen 9 5950X thread) against this library at every scale: one blocking call on host buffers (the reference's own shape: one
operation per call) and, for the F_r transforms, a device-resident batch.  Writes a markdown table.  usage: python tools/scale_sweep.py [out.md]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gokzg_amd as kz  # noqa: E402

KILIC_NS = {   # BENCH.md, "Kilic BLS" column, ns/op
    "FFTExtension": [6436, 9426, 14649, 27099, 50896, 108707, 231713, 516664, 1169011, 2569475, 5421951, 11377382],
    "FFT": [3991, 8383, 19445, 40446, 87280, 197884, 417888, 881049, 1911871, 3891241, 8331212, 15442864],
    "FFTG1": [4592074, 11496429, 28781767, 64030299, 148535217, 305014341, 760028615, 1684792328, 3745748396, 7411640027, 12106541411, 21183031053],
    "RecoverPolyFromSamples": [None, 255459, 571952, 1299551, 2815613, 5835441, 12809586, 25176204, 50779730, 114070337, 199684810, 425497194],
    "ZeroPolyViaMultiplication": [None, 12411, 34445, 222416, 564170, 1418044, 3394786, 7670790, 18257011, 41847868, 83304452, 172534656],
}
R = 52435875175126190479447740508185965837690552500527637822603658699938581184513
lib = kz.lib()
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)


def rand_fr(n):
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


def best(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def dev_rate(call, n, reps=5):
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


rows = []
fs = kz.FFTSettings(16)
setup = None
for scale in range(4, 16):
    n = 1 << scale
    i = scale - 4
    reps = 20 if scale < 13 else 5
    vals = rand_fr(n)
    fse = kz.FFTSettings(scale)                  # like the reference's benchmarks: a settings object of exactly this scale (FFTExtension's recursion walks the full-width tables)
    t_fft = best(lambda: fse.fft(vals), reps)
    half = rand_fr(n // 2)
    t_ext = best(lambda: fse.das_fft_extension(half.copy()), reps)
    B = max(1, (1 << 22) // n)
    d_in = torch.from_numpy(rand_fr(B * n).view(np.int64).reshape(B, n, 4)).cuda()
    d_out = torch.empty_like(d_in)
    t_fft_b = dev_rate(lambda: lib.kzg_hip_fft_fr_batch_dev(fse.h, d_in.data_ptr(), n, B, 0, d_out.data_ptr(), stream), n) / B
    d_h = d_in[:, : n // 2, :].contiguous()
    t_ext_b = dev_rate(lambda: lib.kzg_hip_das_fft_extension_batch_dev(fse.h, d_h.data_ptr(), n // 2, B, stream), n) / B
    fse.close()
    del d_in, d_out, d_h
    # FFTG1 on [s^i] G
    if setup is None or setup.shape[0] < n:
        setup = fs.generate_testing_setup_g1(fs.fr_from_32(np.frombuffer((1927409816240961209460912649124).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))[0][0], 1 << 15)
    pts = np.ascontiguousarray(setup[:n])
    t_g1 = best(lambda: fs.fft_g1(pts), 3 if scale < 13 else 1)
    # recovery: half of the samples missing
    poly = rand_fr(n); poly[n // 2:] = 0
    data = fs.fft(poly)
    present = np.ones(n, dtype=np.uint8); present[rng.permutation(n)[: n // 2]] = 0
    samples = data.copy(); samples[present == 0] = 0
    missing = np.nonzero(present == 0)[0].astype(np.uint64)
    t_zero = best(lambda: fs.zero_poly_via_multiplication(missing, n), 5) if scale >= 5 else None
    if scale >= 5:
        rec = fs.recover_poly_from_samples(samples, present)
        assert np.array_equal(rec, data)
        t_rec = best(lambda: fs.recover_poly_from_samples(samples, present), 5)
    else:
        t_rec = None
    rows.append((scale, t_fft, t_fft_b, t_ext, t_ext_b, t_g1, t_rec, t_zero))
    print("scale", scale, ["%.3g" % (x * 1e3) if x else None for x in rows[-1][1:]], flush=True)
fs.close()


def cell(t, ref_ns):
    if t is None:
        return "—"
    return "%.3g ms (×%.0f)" % (t * 1e3, ref_ns / 1e9 / t) if ref_ns else "%.3g ms" % (t * 1e3)


out = ["| scale | FFT: one call | FFT: resident batch, per transform | FFTExtension: one call | FFTExtension: resident batch | FFTG1: one call | RecoverPolyFromSamples | ZeroPolyViaMultiplication |",
       "|---|---|---|---|---|---|---|---|"]
for (scale, t_fft, t_fft_b, t_ext, t_ext_b, t_g1, t_rec, t_zero) in rows:
    i = scale - 4
    out.append("| %d | %s | %s | %s | %s | %s | %s | %s |" % (scale, cell(t_fft, KILIC_NS["FFT"][i]), cell(t_fft_b, KILIC_NS["FFT"][i]), cell(t_ext, KILIC_NS["FFTExtension"][i]),
                                                           cell(t_ext_b, KILIC_NS["FFTExtension"][i]), cell(t_g1, KILIC_NS["FFTG1"][i]), cell(t_rec, KILIC_NS["RecoverPolyFromSamples"][i]),
                                                           cell(t_zero, KILIC_NS["ZeroPolyViaMultiplication"][i])))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
