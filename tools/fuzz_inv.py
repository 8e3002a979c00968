#!/usr/bin/env python3
"""Fuzz of the F_p inversion on the device: wave-cooperative form (coop_inv.hpp) == one-lane form (field.hpp) == pow(x, -1, p), on structured and random elements.
usage: python tools/fuzz_inv.py [rounds [seed]]   (each round: 65 536 elements; every 64th is also checked against Python)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokzg_amd as kz  # noqa: E402

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 1 << 390
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
fs = kz.FFTSettings(4)
N = 65536
bad = 0
for r in range(rounds):
    w = rng.integers(0, 1 << 32, size=(N, 12), dtype=np.uint64).astype(np.uint32)
    mode = rng.integers(0, 5, size=N)
    w[mode == 1] &= rng.integers(0, 1 << 32, size=(int((mode == 1).sum()), 12), dtype=np.uint64).astype(np.uint32)          # sparse bit patterns
    w[mode == 2] |= rng.integers(0, 1 << 32, size=(int((mode == 2).sum()), 12), dtype=np.uint64).astype(np.uint32)          # dense ones
    short = np.nonzero(mode == 3)[0]
    for i in short:
        w[i, int(rng.integers(0, 12)):] = 0                                                                                  # short values
    w[:, 11] &= (1 << 28) - 1                                                                                                # below p
    if r == 0:
        for i, x in enumerate([0, 1, 2, P - 1, P - 2, (P + 1) // 2, 1 << 380, 1 << 30, (1 << 30) - 1, (1 << 360) + 1]):
            w[i] = np.frombuffer((x * R % P).to_bytes(48, "little"), dtype=np.uint32)
    img = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
    a, b = np.zeros_like(img), np.zeros_like(img)
    st = kz.lib().kzg_hip_test_fp_inv(fs.h, img.ctypes.data, N, a.ctypes.data, b.ctypes.data, None, None)
    assert st == 0, kz.lib().kzg_hip_last_error()
    bad += int((a.reshape(N, 48) != b.reshape(N, 48)).any(axis=1).sum())
    for i in range(0, N, 64):
        x = int.from_bytes(img[48 * i:48 * i + 48].tobytes(), "little")
        y = int.from_bytes(a[48 * i:48 * i + 48].tobytes(), "little")
        if x % P == 0:
            bad += y != 0
        else:
            bad += (x * y - R * R) % P != 0          # (x R)(x^-1 R) = R^2
print("fuzz_inv: %d elements, %d mismatches" % (rounds * N, bad))
sys.exit(1 if bad else 0)
