"""Constants of the workload, synthetic blobs (splitmix64 streams), rank environment, the timed-steps harness and unit sharding."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R_MOD = 52435875175126190479447740508185965837690552500527637822603658699938581184513
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
N_COEFF = 4096
# algorithmic bytes of one commitment with the setup resident (scalars 131072 + 96 out), and the
# setup itself (4096 affine points = 393216 B) counted once per launch.
BYTES_PER_COMMIT = 131072 + 96
BYTES_SETUP = 393216
FK20_BYTES = 851968
FK20_4096_BYTES = 1310720                  # 131 072 poly + 786 432 xExtFFT + 393 216 proofs
S_TEST = 1927409816240961209460912649124   # the reference's test secret (kzg_single_proofs_test.go:15): setups longer than eth/trusted_setup.json


def splitmix_blobs(base_seed, batch, n=N_COEFF):
    """SURVEY.md 8(d) synthetic scalars -> Montgomery images, shape (batch, n, 4) uint64 (host-side input synthesis)."""
    out = np.empty((batch, n, 4), dtype=np.uint64)
    mask = (1 << 64) - 1
    rmont = (1 << 256) % R_MOD
    for b in range(batch):
        idx = np.arange(1, 4 * n + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = np.uint64((base_seed + b) & mask) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        limbs = z.reshape(n, 4)
        raw = limbs.tobytes()
        row = bytearray(n * 32)
        for i in range(n):
            v = int.from_bytes(raw[32 * i:32 * i + 32], "little") % R_MOD
            row[32 * i:32 * i + 32] = (v * rmont % R_MOD).to_bytes(32, "little")
        out[b] = np.frombuffer(bytes(row), dtype=np.uint64).reshape(n, 4)
    return out


_R_LIMBS = [(R_MOD >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def splitmix_blobs_le32(base_seed, batch, n=N_COEFF):
    """The same scalars as splitmix_blobs, in STANDARD form as 32 little-endian bytes each (batch, n, 32) uint8, fully
    vectorised (the 256-bit value is < 2^256 < 3 r: at most two conditional subtractions of r).  The Montgomery conversion
    is then done on the device with kzg_hip_fr_from_le32 (bls.FrFrom32 over a slice)."""
    seeds = (np.uint64(base_seed & ((1 << 64) - 1)) + np.arange(batch, dtype=np.uint64))[:, None]
    idx = np.arange(1, 4 * n + 1, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        z = seeds + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        v = z.reshape(batch, n, 4)
        r = np.array(_R_LIMBS, dtype=np.uint64)
        for _ in range(2):
            ge = np.ones(v.shape[:2], dtype=bool)          # v >= r, lexicographic from the top limb
            decided = np.zeros(v.shape[:2], dtype=bool)
            for k in (3, 2, 1, 0):
                gt, lt = v[..., k] > r[k], v[..., k] < r[k]
                ge = np.where(~decided & lt, False, ge)
                decided |= gt | lt
            borrow = np.zeros(v.shape[:2], dtype=np.uint64)
            out = v.copy()
            for k in range(4):
                d = v[..., k] - r[k] - borrow
                borrow = ((v[..., k] < r[k] + borrow) | ((r[k] + borrow) < r[k])).astype(np.uint64)
                out[..., k] = d
            v = np.where(ge[..., None], out, v)
    return np.ascontiguousarray(v).view(np.uint8).reshape(batch, n, 32)


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def timed_steps(step_fn, steps, warmup, sync_fn, barrier_fn, max_over_ranks_fn, before_timed=None):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync; returns max-over-ranks seconds.
    Backend-agnostic so that tests/test_bench_dist.py can drive it with gloo on CPU.  `before_timed` runs after the warm-up
    has drained and before the clock starts (bench.py switches the per-kernel HIP-event records on there, so that the
    roofline's launch durations are those of the TIMED steps and nothing of the warm-up)."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    barrier_fn()
    sync_fn()
    if before_timed is not None:
        before_timed()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    barrier_fn()
    t1 = time.perf_counter()
    return max_over_ranks_fn(t1 - t0)


def shard_units(total_units, world, rank):
    """contiguous shard [lo, hi) of `total_units` for `rank` (used for strong-scaling workloads and FK20Multi positions)"""
    base, rem = divmod(total_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
