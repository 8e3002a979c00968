#!/usr/bin/env python3
"""bench.py -- headline benchmark of the commitment / proof hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...: RANK /
    WORLD_SIZE come from the environment), or bare -- `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N
    ranks (one per GPU, torch.distributed.run on 127.0.0.1 with a free port) and prints their ONE JSON line.

Workload (BASELINE.json configs[1]): KZGSettings.CommitToPoly on 4096-coefficient blobs against the 4096-point
monomial setup of eth/trusted_setup.json (s = 1337; rebuilt from tests/golden/trusted_setup_g1.bin through the
library's own FromCompressedG1).  One step = one pass of the hot path over one batch of `--batch` synthetic blobs
(splitmix64 stream, blob b uses seed base + b) that are ALREADY resident in HBM; the step ends with the
`batch` normalised commitments resident in HBM.  Blobs are independent, so ranks shard them with no data-path
collective ("scaling": "weak"); value = commitments of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      -- dominant kernel of the commitment step (k_fb_accumulate, the fixed-base table walk): algorithmic bytes per launch /
                   HIP-event launch time OF THE TIMED STEPS (launches_timed, launch_over_step <= 1.002 asserted) vs 8 TB/s, the PMC traffic of the committed
                   rocprofv3 passes (profiles/), and `mac`: the kernel's v_mad_u64_u32 rate against the rate MEASURED on this GPU in this run (kzg_hip_calibrate).
  roofline.secondary.fk20 -- the same for the FK20 half of the metric (dominant kernel k_g1_fft_stage).
  cpu_baseline  -- the oracle's restatement of bls.LinCombG1 (Kilic-style Pippenger) on the host: one core and all cores (one blob
                   per core), CPU model and core count stated; the Go toolchain probe; port_vs_published (the port against BENCH.md's transforms) (rank 0, N = 1).
  cpu_baseline_fk20_4096 -- the FK20 half on the host: one oracle FK20Single on 512 coefficients, scaled to 4096 by the reference's MulG1 count (estimate, labelled).
  table_sweep   -- commitments/s against the HBM budget of the fixed-base table (5 / 9 / 17 / 60 / 110 GB).
  drop_in       -- the reference's ONE-blob-per-call API from 1 .. 256 native host threads (host buffers, coalesced in the library): three runs per thread
                   count (min / median / max) with the coalescer's own counters (batches, rows per batch, ms per batch by phase); coalescer_256 = the 256-caller run's.
  lincomb       -- variable-base bls.LinCombG1 on a cached point set (GLV bucket MSM), batch 1 / 64 / 512.
  latency       -- single-call latencies of the reference-shaped entry points; LinCombG1 on NEW caller-supplied points (one-shot) and on the SAME points again (promoted).
  fk20          -- DAUsingFK20 (2048 coefficients -> 4096 proofs) all-proofs/s, own timed loop, self-checked against the byte pin.
  fk20_4096     -- the metric's literal input, a 4096-ELEMENT blob (BASELINE config 4b): FK20Single 4096 coefficients -> 4096 proofs and
                   DAUsingFK20 4096 -> 8192 at scale 13 on the 8192-point setup of the reference's test secret: batch rates, lone latencies,
                   roofline on SURVEY.md 8(d)'s 1 310 720 B per unit, self-checked against the config-4b byte pins.
  self_check    -- EVERY output of the timed steps, not one: sum_b rho_b C_b == CommitToPoly(sum_b rho_b blob_b) for random rho (one B-term
                   LinCombG1 over the step's outputs as caller-supplied points + one commitment), and for the FK20 blocks
                   sum_b rho_b sum_j sigma_j proof_b[j] == sum_j sigma_j FK20(sum_b rho_b p_b)[j] (the right side through the one-polynomial path).
  in_process    -- the multi-device handle of the C ABI (kzg_hip_multi_*, what a Go caller sees) in a child process after this one has released
                   its tables: host-buffer batches divided among the devices, ONE polynomial sharded inside the library (all-gather schemes).
  roofline.secondary -- the rooflines of the FK20 / F_r kernels; roofline.profile_avg_ms -- the same launch shape in the committed kernel trace.
  fk20_multi    -- BASELINE config 5: DAUsingFK20Multi at scale 16, chunk 16 (32768 coefficients -> 4096 coset proofs).
  reference_benchmarks -- FFT_Fr / DAS extension / FFT_G1 at scale 12 beside the reference's published BENCH.md numbers.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.workload import *  # noqa: E402,F401,F403  (constants, splitmix_blobs[_le32], dist_env, timed_steps, shard_units: tests and tools use them as bench.X)
from benchlib.workload import R_MOD, HBM_PEAK_GBS, N_COEFF, BYTES_PER_COMMIT, BYTES_SETUP, FK20_BYTES, FK20_4096_BYTES, S_TEST  # noqa: E402,F401
from benchlib.cpu import cpu_baseline  # noqa: E402
from benchlib.roofline import walk_roofline, walk_mac, issue_model  # noqa: E402
from benchlib.launch import self_launch  # noqa: E402
from benchlib.in_process import run_in_process_child, in_process_child  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="blobs per step per GPU (round 1: 512; see batch_sweep)")
    ap.add_argument("--fk20-batch", type=int, default=1024)
    ap.add_argument("--fk20-multi-batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-fk20-multi", action="store_true", help="also time ONE FK20Multi through the sharded driver at world size 1")
    ap.add_argument("--no-fk20", action="store_true")
    ap.add_argument("--table-gb", type=float, default=110.0, help="HBM budget of the commitment table for the headline (110 = the library default: 8 signed 16-bit windows walked by both GLV halves, 103 GB)")
    ap.add_argument("--no-extras", action="store_true", help="skip table_sweep / drop_in / lincomb / latency (profiling runs)")
    ap.add_argument("--fk20-4096-batch", type=int, default=512, help="polynomials per step of the fk20_4096 block (0: skip)")
    ap.add_argument("--no-in-process", action="store_true", help="skip the multi-device-handle leg (a child process at the end)")
    ap.add_argument("--in-process", action="store_true", help="(default) also time the multi-device handle of the C ABI over the job's devices: reported under in_process and rccl")
    ap.add_argument("--in-process-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--devices", type=str, default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.in_process_child:
        sys.exit(in_process_child([int(x) for x in args.devices.split(",") if x != ""]))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank, world, local = dist_env()
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            base = cpu_baseline(float(os.environ.get("KZG_BENCH_CPU_BUDGET_S", "6")))   # first: the all-cores leg forks, which must precede HIP initialisation
        except Exception as e:                              # noqa: BLE001  (a missing / unbuildable oracle must not cost the GPU measurement)
            base = {"value": None, "unit": "commitments/s", "cores": 0, "kind": "port", "sample": "failed: %s: %s" % (type(e).__name__, e)}

    import torch
    import gokzg_amd as kz

    if not torch.cuda.is_available() or kz.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path")
    local = local % max(1, torch.cuda.device_count())        # (test hook: several ranks on one GPU with KZG_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("KZG_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only to smoke-test N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = kz.lib()
    fs = kz.FFTSettings(12, device=local)
    raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    setup = fs.from_compressed_g1(raw)                      # 4096 x [1337^i]G1, decompressed on the device
    ks = kz.KZGSettings(fs, setup)
    ks.set_table_budget_gb(args.table_gb)                   # 110 GB = the library's default budget (c = 16 on 8 windows, 103 GB); table_sweep has the smaller ones
    cal_mad, cal_add, cal_fpmul = fs.calibrate()            # measured on THIS GPU: v_mad_u64_u32 / v_add_u32 lane-ops/s, lazy F_p products/s
    golden = os.path.join(ROOT, "tests", "golden")
    pins = json.load(open(os.path.join(golden, "fk20_pins.json")))
    pmc = {}
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc_traffic.json"):   # counters of the committed rocprofv3 passes (tools/profile_round.sh)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            pmc["_file"] = "profiles/" + name
            break
        except (OSError, ValueError):
            continue

    shapes, shapes_file = [], None
    for name in ("r06_kernel_shapes.json", "r05_kernel_shapes.json", "r04_kernel_shapes.json"):                 # (kernel, grid, workgroup) rows of the committed rocprofv3 kernel trace (tools/rocprof_summary.py)
        try:
            shapes = json.load(open(os.path.join(ROOT, "profiles", name)))["rows"]
            shapes_file = "profiles/" + name
            break
        except (OSError, ValueError, KeyError):
            continue

    def profile_avg_ms(prefix, grid, wg):
        """average duration of the launches of that shape in the committed kernel trace (all kernels whose name starts with `prefix`)"""
        rows_ = [r for r in shapes if r["kernel"].startswith(prefix) and r["grid"] == grid and r["workgroup"] == wg]
        calls = sum(r["calls"] for r in rows_)
        if not calls:
            return None, None
        return sum(r["avg_us"] * r["calls"] for r in rows_) / calls * 1e-3, "%s (rows: %s): %s, grid %d, workgroup %d (%d launches)" % (
            shapes_file.replace("kernel_shapes.json", "kernel_stats.md"), shapes_file, " + ".join(sorted(set(r["kernel"] for r in rows_))), grid, wg, calls)

    def mont_blobs(seed, batch, n=N_COEFF):
        """synthetic scalars as Montgomery images: vectorised splitmix + mod r on the host, FrFrom32 on the device"""
        std = splitmix_blobs_le32(seed, batch, n)
        out, ok = fs.fr_from_32(std.reshape(-1, 32))
        assert ok
        return out.reshape(batch, n, 4)

    B = args.batch
    blobs_h = mont_blobs(1 + rank * B, B)                   # rank r owns blobs [r B, (r + 1) B)
    d_blobs = torch.from_numpy(blobs_h.view(np.int64)).cuda()
    d_out = torch.zeros((B, 18), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        st = lib.kzg_hip_commit_to_poly_batch_dev(ks.h, d_blobs.data_ptr(), N_COEFF, B, d_out.data_ptr(), stream)
        if st:
            raise RuntimeError("commit_to_poly_batch_dev status %d %s" % (st, lib.kzg_hip_last_error().decode()))

    def barrier():
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # self-check of the timed path before timing it: blob 0 of rank 0 is SURVEY.md vector F
    step()
    torch.cuda.synchronize()
    if rank == 0:
        c0 = d_out[0].cpu().numpy().view(np.uint64).reshape(1, 3, 6)
        hx = fs.to_compressed_g1(c0)[0].tobytes().hex()
        exp = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")))["F_blob_seed1"]["commit_monomial_s1337"]
        if hx != exp:
            raise SystemExit("bench self-check failed: commitment of blob(seed 1) = %s, expected %s" % (hx, exp))

    tc_, tw_, tb_ = ks.table_info()   # every rank builds its own table; an allocation failure degrades to a smaller window (capi.hip)
    sys.stderr.write("[rank %d/%d, device %d] commitment table: %d-bit windows x %d, %.1f GB\n" % (rank, world, local, tc_, tw_, tb_ / 1e9))
    rccl = None
    if use_dist:
        # who is in the job: every rank contributes (rank, device index, table window bits, windows, table bytes) through the SAME
        # collective library the data path would use (one all-gather of 5 x int64); rank 0 reports what it saw
        mine = torch.tensor([rank, local, tc_, tw_, tb_], dtype=torch.int64, device="cuda")
        seen = torch.empty(world * 5, dtype=torch.int64, device="cuda")     # flat: gloo's all-gather wants output == world x input, 1-D
        dist.all_gather_into_tensor(seen, mine)
        seen = seen.view(world, 5).cpu().tolist()
        rccl = {"backend": dist.get_backend(), "library": "RCCL (torch.distributed 'nccl' on ROCm)" if dist.get_backend() == "nccl" else "gloo (test transport: several ranks on one GPU)",
                "world_size": world, "ranks_seen": [r[0] for r in seen], "devices": [r[1] for r in seen],
                "self_launched": bool(os.environ.get("KZG_BENCH_SELF_LAUNCHED")),
                "tables_per_rank": [{"rank": r[0], "window_bits": r[2], "windows": r[3], "GB": r[4] / 1e9} for r in seen],
                "all_gather_proofs_ms": None, "sharded_one_polynomial_ms": None}
    # HIP events around every kernel of the TIMED steps (on the launch stream): the roofline's launch duration is taken from the very launches `value` is made of
    secs = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, barrier, max_over_ranks, before_timed=lambda: lib.kzg_hip_prof_reset(fs.h, 1))
    tot, cnt = C.c_double(0), C.c_uint64(0)
    dominant = b"fb_accumulate"
    lib.kzg_hip_prof_read(fs.h, dominant, C.byref(tot), C.byref(cnt))
    if not cnt.value:
        dominant = b"msm_accumulate"
        lib.kzg_hip_prof_read(fs.h, dominant, C.byref(tot), C.byref(cnt))
    lib.kzg_hip_prof_reset(fs.h, 0)
    value = B * world * args.steps / secs

    def poly_lincomb(d_rows, stride, rho_h, count, n):
        """sum_c rho_c row_c over device-resident rows (bls.PolyLinComb's kernel), returned as host images"""
        d_rho = torch.from_numpy(np.ascontiguousarray(rho_h).view(np.int64)).cuda()
        d_comb = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        st = lib.kzg_hip_bench_poly_lincomb_dev(fs.h, d_rows.data_ptr(), stride, d_rho.data_ptr(), count, n, d_comb.data_ptr(), stream)
        if st:
            raise RuntimeError("bench_poly_lincomb_dev status %d" % st)
        torch.cuda.synchronize()
        return d_comb.cpu().numpy().view(np.uint64).reshape(n, 4)

    def all_ranks_ok(ok):
        if not use_dist:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def check_proof_rows(fsx, d_proofs_, rows, per_row, d_polys_, n_coeff, one_polynomial, seed):
        """every proof of a timed FK20 step: FK20 is linear in the polynomial, so for random rho, sigma
        sum_b rho_b (sum_j sigma_j proof_b[j]) == sum_j sigma_j FK20(sum_b rho_b p_b)[j]; the left side is `rows` LinCombG1 calls over the step's
        outputs as caller-supplied points (bucket pipeline) and one more over their results, the right side the ONE-polynomial path of the
        library (direct passes: other kernels than the batch) on the combined polynomial"""
        sigma, rho = mont_blobs(seed, 1, n=per_row)[0], mont_blobs(seed + 1, 1, n=rows)[0]
        P = d_proofs_.cpu().numpy().view(np.uint64).reshape(rows, per_row, 3, 6)
        R = np.stack([fsx.lin_comb_g1(P[b], sigma) for b in range(rows)])
        lhs = fsx.lin_comb_g1(R, rho)
        q = one_polynomial(poly_lincomb(d_polys_, n_coeff, rho, rows, n_coeff))
        return bool(np.array_equal(lhs, fsx.lin_comb_g1(q, sigma)))

    # ALL outputs of the last timed step (the KAT above pins blob 0 only): sum_b rho_b C_b == CommitToPoly(sum_b rho_b blob_b) for random rho --
    # one B-term LinCombG1 over the step's outputs as caller-supplied points and one lone commitment (other launch shapes than the timed one)
    rho_c = mont_blobs(0xC0FFEE + rank, 1, n=B)[0]
    lhs_c = fs.lin_comb_g1(d_out.cpu().numpy().view(np.uint64).reshape(B, 3, 6), rho_c)
    rhs_c = ks.commit_to_poly(poly_lincomb(d_blobs, N_COEFF, rho_c, B, N_COEFF))
    if not all_ranks_ok(np.array_equal(lhs_c, rhs_c)):
        raise SystemExit("bench self-check failed: sum_b rho_b C_b != CommitToPoly(sum_b rho_b blob_b) over the %d outputs of the timed step" % B)
    self_check = {"rows_checked": B, "per_rank": True, "ranks": world,
                  "method": "sum_b rho_b C_b == CommitToPoly(sum_b rho_b blob_b), rho random in F_r: one %d-term LinCombG1 over the step's outputs + one commitment; "
                            "blob 0 also against SURVEY.md vector F" % B}

    # SURVEY.md 8(d) config 2 also names the batch sizes 1, 64 and 1024: the same step at those sizes and at round 1's 512 (secondary
    # figures; the headline's step is `--batch` blobs: the end of a launch -- block trees, one inversion per blob -- is amortised over more
    # work the larger the step)
    batch_sweep = {}
    if not args.no_fk20:
        big = mont_blobs(1 + rank * 2048, 2048)
        d_big = torch.from_numpy(big.view(np.int64)).cuda()
        d_big_out = torch.zeros((2048, 18), dtype=torch.int64, device="cuda")
        for bs in (1, 64, 512, 1024, 2048):
            def sweep_step(bs=bs):
                st = lib.kzg_hip_commit_to_poly_batch_dev(ks.h, d_big.data_ptr(), N_COEFF, bs, d_big_out.data_ptr(), stream)
                if st:
                    raise RuntimeError("commit_to_poly_batch_dev status %d" % st)
            reps = 20 if bs < 1024 else 8
            ssecs = timed_steps(sweep_step, reps, 2, torch.cuda.synchronize, barrier, max_over_ranks)
            batch_sweep[str(bs)] = {"commitments_per_s": bs * world * reps / ssecs, "ms_per_step": ssecs / reps * 1e3}
        del d_big, d_big_out

    # roofline leg: the HIP-event records of the timed steps above (tot, cnt)
    roofline = None
    if cnt.value:
        avg_s = tot.value / cnt.value * 1e-3
        alg_bytes = B * BYTES_PER_COMMIT + BYTES_SETUP
        tab_c, tab_w, tab_bytes = ks.table_info()
        # HBM bytes per launch come from the committed PMC passes (profiles/) only when the workload matches that measurement (benchlib/roofline.py)
        roofline, pm, pm_sc = walk_roofline("k_" + dominant.decode(), B, avg_s, (tab_c, tab_w, tab_bytes), pmc, pmc.get("_file"))
        pm_ok = pm is not None
        # measured on the args.steps timed launches themselves: a launch cannot be longer than the step it is part of
        roofline["launches_timed"], roofline["timed_in"] = int(cnt.value), "the %d timed steps (HIP events on the launch stream; nothing of the warm-up)" % args.steps
        roofline["launch_over_step"] = avg_s * (cnt.value / args.steps) / (secs / args.steps)
        if roofline["launch_over_step"] > 1.002:                 # (reported, never fatal: the line must reach the driver; tests/test_bench_dist.py asserts the bound)
            roofline["launch_over_step_warning"] = "the dominant kernel's launches (%.3f ms per step by HIP events) exceed the step (%.3f ms by the host clock)" % (
                avg_s * 1e3 * cnt.value / args.steps, secs / args.steps * 1e3)
        if dominant == b"fb_accumulate" and B >= 512:           # one 256-lane workgroup per blob from 512 blobs on: the row of this launch shape
            pa_, ps_ = profile_avg_ms("k_fb_accumulate", B * 256, 256)
            roofline["profile_avg_ms"], roofline["profile_source"] = pa_, ps_
            if pa_:
                roofline["profile_frac"] = alg_bytes / (pa_ * 1e-3) * 1e-9 / HBM_PEAK_GBS
        if tab_w:
            # What bounds the walk is integer issue, not HBM: `mac` sets the multiply-adds against the v_mad_u64_u32 rate MEASURED in this run, `issue` adds the
            # non-multiply instructions (SQ_INSTS_VALU of the committed counter pass) at the measured v_add_u32 rate
            adds_pt = ks.table_additions()                       # 2 x windows: both GLV halves of a scalar walk the same rows
            roofline["table"]["additions_per_coefficient"] = adds_pt
            roofline["mac"] = walk_mac(B, adds_pt, avg_s, cal_mad, cal_add, cal_fpmul)
            if pm_ok and "valu_insts_per_launch" in pm:
                roofline["issue"] = issue_model(roofline["mac"]["mads_per_launch"], pm["valu_insts_per_launch"] * pm_sc, avg_s, cal_mad, cal_add)

    # Everything below is secondary to the headline measured above.  On one GPU a failure in a secondary leg is recorded in
    # `secondary_error` and the line is still printed; with several ranks it is raised (a rank that skipped ahead would leave the others
    # in a barrier).
    table_sweep = drop_in = lincomb = latency = fk20 = roofline_fk20 = fk20m = ref_benches = fk20_4096 = roofline_fk20_4096 = roofline_fft_fr = roofline_das = None
    secondary_error = None
    try:
        if os.environ.get("KZG_BENCH_FAIL_SECONDARY"):           # test hook (tests/test_bench_dist.py)
            raise RuntimeError("KZG_BENCH_FAIL_SECONDARY is set")
        if not args.no_extras and not args.no_fk20 and world == 1:   # single-GPU characterisations: not repeated by every rank of an N > 1 run
            # --- commitments/s against the HBM budget of the fixed-base table (library default: 110 GB -> c = 16 on 8 windows = 103 GB, the headline)
            table_sweep = {}
            for gb in (5.0, 9.0, 17.0, 60.0):
                ks.set_table_budget_gb(gb)
                step()
                torch.cuda.synchronize()
                tsecs = timed_steps(step, 5, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                c_, w_, b_ = ks.table_info()
                table_sweep["%g" % gb] = {"commitments_per_s": B * world * 5 / tsecs, "window_bits": c_, "windows": w_, "table_GB": b_ / 1e9}
            # --- eth.ComputeKZGProof (eth/helpers.go:179-203) while the monomial settings hold a 58 GB table; the eth settings build their own
            # default table (103 GB) beside it
            eth_proof = None
            try:
                lag_raw = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8)
                eth = kz.EthSettings(fs, fs.from_compressed_g1(lag_raw))
                EB = min(512, B)
                d_z = torch.from_numpy(fs.fr_from_32(splitmix_blobs_le32(77, 1, EB).reshape(-1, 32))[0].view(np.int64)).cuda()
                d_p48 = torch.zeros((EB, 48), dtype=torch.uint8, device="cuda")
                d_bad = torch.zeros(EB, dtype=torch.int32, device="cuda")

                def eth_step():
                    st = lib.kzg_hip_eth_compute_kzg_proof_batch_dev(eth.h, d_blobs.data_ptr(), N_COEFF, EB, d_z.data_ptr(), d_p48.data_ptr(), None, d_bad.data_ptr(), stream)
                    if st:
                        raise RuntimeError("eth_compute_kzg_proof_batch_dev status %d %s" % (st, lib.kzg_hip_last_error().decode()))
                eth_step()
                torch.cuda.synchronize()
                esecs = timed_steps(eth_step, 5, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                eth.bench_drop_in_proof(blobs_h[:64], 8, 4)
                erates = {}
                for T in (1, 8, 64):
                    erates[str(T)] = eth.bench_drop_in_proof(blobs_h[:64], T, 100 if T == 1 else 40)[0]
                eth_proof = {"entry": "kzg_hip_eth_compute_kzg_proof (host buffers, blocking, one polynomial per call; coalesced in the library)",
                             "device_resident_batch_%d_per_s" % EB: EB * 5 / esecs, "invalid_rows_in_batch": int((d_bad != 0).sum().item()),
                             "threads_per_s": erates, "monomial_table_GB_beside_it": table_sweep["60"]["table_GB"]}
                # eth.ComputeAggregateKZGProof (eth/eth.go:175-182): the blobs of one block from host buffers -> commitments + aggregated proof,
                # the SHA-256 transcript hashed on the host while the device commits
                agg = {}
                blob_bytes = splitmix_blobs_le32(99, 16, N_COEFF)
                for nb in (1, 4, 16):
                    eth.compute_aggregate_kzg_proof(blob_bytes[:nb])
                    t0_ = time.perf_counter()
                    for _ in range(10):
                        eth.compute_aggregate_kzg_proof(blob_bytes[:nb])
                    agg["%d_blobs_ms" % nb] = (time.perf_counter() - t0_) / 10 * 1e3
                eth_proof["compute_aggregate_kzg_proof_host_buffers"] = agg
                # lone calls of the byte-level entry points (median of 30 after 3 warm-up calls, like `latency`)
                def med_ms(fn, reps=30):
                    for _ in range(3):
                        fn()
                    ts_ = []
                    for _ in range(reps):
                        t0_ = time.perf_counter()
                        fn()
                        ts_.append((time.perf_counter() - t0_) * 1e3)
                    return float(np.median(ts_))
                z_one = mont_blobs(4242, 1, 1).reshape(1, 4)
                eth_proof["lone_call_ms"] = {"ComputeKZGProof": med_ms(lambda: eth.compute_kzg_proof(blobs_h[0], z_one)),
                                             "BlobToKZGCommitment": med_ms(lambda: eth.blob_to_kzg_commitment(blob_bytes[0]))}
                eth.close()
            except Exception as e:                              # noqa: BLE001
                eth_proof = {"error": "%s: %s" % (type(e).__name__, e)}
            ks.set_table_budget_gb(args.table_gb)
            step()
            torch.cuda.synchronize()
            c_, w_, b_ = ks.table_info()
            table_sweep["%g" % args.table_gb] = {"commitments_per_s": value, "window_bits": c_, "windows": w_, "table_GB": b_ / 1e9, "headline": True}

            # --- the reference's API is ONE blob per call: T native host threads, each calling kzg_hip_commit_to_poly on host buffers
            host_blobs = blobs_h[:64].copy()
            drop_in = {"entry": "kzg_hip_commit_to_poly (host buffers, blocking, one 4096-coefficient blob per call)", "threads": {}}
            ks.bench_drop_in(host_blobs, 8, 4)
            # every thread count three times (min / median / max: a cold box and a warm one have differed by 30 % at 256 callers); `commitments_per_s` is the median.
            # The coalescer's own counters over the three 256-caller runs (batches, rows per batch, where a batch's time goes) make a discrepancy diagnosable from the line alone
            for T in (1, 8, 64, 256):
                before = ks.coalesce_stats(0)
                runs = []
                for _rep in range(3):
                    rate_, outs = ks.bench_drop_in(host_blobs, T, 200 if T == 1 else 60)
                    runs.append(rate_)
                runs.sort()
                rate_ = runs[1]
                drop_in["threads"][str(T)] = {"commitments_per_s": rate_, "min": runs[0], "median": runs[1], "max": runs[2], "runs": 3,
                                              "frac_of_device_resident_batch": rate_ / (value / world),
                                              "frac_of_512_blob_resident_rate": rate_ / (batch_sweep["512"]["commitments_per_s"] / world)}
                after = ks.coalesce_stats(0)
                nb = after["batches"] - before["batches"]
                if nb > 0:
                    ms = {k: (after["ms_per_batch"][k] * after["batches"] - before["ms_per_batch"][k] * before["batches"]) / nb for k in after["ms_per_batch"]}
                    drop_in["threads"][str(T)]["coalescer"] = {"batches": nb, "avg_batch": (after["requests"] - before["requests"]) / nb, "ms_per_batch": ms,
                                                               "largest_concurrency_estimate": after["largest_concurrency_estimate"],
                                                               "batches_in_flight_limit": after["batches_in_flight_limit"]}
            drop_in["coalescer_256"] = drop_in["threads"]["256"].get("coalescer")
            drop_in["host_cores"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
            want0 = d_out[(T - 1 + 59) % 64].cpu().numpy().view(np.uint64).reshape(3, 6)      # thread T-1's last call used blob (T-1 + 59) % 64
            drop_in["bit_exact_vs_batched_path"] = bool(np.array_equal(outs[T - 1], want0))
            prate_, _ = ks.bench_drop_in(host_blobs, 64, 40, op=1)
            drop_in["compute_proof_single_64_threads_per_s"] = prate_
            drop_in["eth_compute_kzg_proof"] = eth_proof

            # --- variable-base bls.LinCombG1 on a cached point set (the seam eth/helpers.go:99,159,199 and CommitToEvalPoly go through)
            pts = kz.G1Points(fs, setup)
            d_lc_out = torch.zeros((512, 18), dtype=torch.int64, device="cuda")
            lincomb = {"n": N_COEFF, "batch": {}, "bucket_pipeline_batch": {},
                       "kernel_chain": "a cached set walks its own fixed-base table (k_fb_accumulate; default budget min(32 GB, free HBM - 24 GB): 13-bit windows for 4096 points); "
                                       "bucket_pipeline_batch = the same set with the table budget at 0: k_msm_sort / accumulate / reduce / combine (GLV halves, signed 8-bit "
                                       "windows, 2^64 rows cached), which is also what caller-supplied points take (latency.LinCombG1_4096_one_shot_ms)"}

            def lc_rates(key):
                for bs in (1, 64, 512):
                    def lc_step(bs=bs):
                        st = lib.kzg_hip_lincomb_points_batch_dev(pts.h, d_blobs.data_ptr(), N_COEFF, bs, d_lc_out.data_ptr(), stream)
                        if st:
                            raise RuntimeError("lincomb_points_batch_dev status %d" % st)
                    lc_step()
                    torch.cuda.synchronize()
                    reps = 10 if bs < 512 else 3
                    lsecs = timed_steps(lc_step, reps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                    lincomb[key][str(bs)] = {"msm_per_s": bs * world * reps / lsecs, "ms_per_step": lsecs / reps * 1e3}
            pts.set_table_budget_gb(0)
            lc_rates("bucket_pipeline_batch")
            torch.cuda.synchronize()
            nchk = min(B, 512)                                   # the last step of each form ran on the first 512 blobs
            d_lc_bucket = d_lc_out[:nchk].clone()                # the bucket pipeline's own output, before the table walk overwrites the buffer
            pts.set_table_budget_gb(32)
            lc_rates("batch")
            lincomb["table"] = "budget 32 GB"
            torch.cuda.synchronize()
            lincomb["bucket_pipeline_matches_fixed_base_commitments"] = bool(torch.equal(d_lc_bucket, d_out[:nchk]))   # k_msm_* (incl. reduce_chunks at 512) vs k_fb_accumulate
            lincomb["matches_fixed_base_commitments"] = bool(torch.equal(d_lc_out[:nchk], d_out[:nchk]))              # the set's 32 GB table vs the settings' table
            # one linear combination per call (bls.LinCombG1's shape) from 64 host threads: coalesced into batched bucket MSMs
            import threading
            lc_T, lc_per = 64, 30
            lc_gate = threading.Barrier(lc_T + 1)
            def lc_worker(i):
                lc_gate.wait()
                for r in range(lc_per):
                    pts.lin_comb(blobs_h[(i + r) % 64])
            for _ in range(5):
                pts.lin_comb(blobs_h[0])
            lc_ths = [threading.Thread(target=lc_worker, args=(i,)) for i in range(lc_T)]
            [t.start() for t in lc_ths]
            lc_gate.wait()
            lc_t0 = time.perf_counter()
            [t.join() for t in lc_ths]
            lincomb["one_call_at_a_time_from_64_threads_per_s"] = lc_T * lc_per / (time.perf_counter() - lc_t0)
            # the same 64 callers through bls.LinCombG1's own signature (caller-supplied points, the SAME slice every time): promoted to a cached set by the third call,
            # every call compares its 590 KB of points with the kept copy (outside any lock) and then joins the set's coalesced batches
            lc_pts = np.ascontiguousarray(np.roll(setup, 5, axis=0))
            for _ in range(5):
                fs.lin_comb_g1(lc_pts, blobs_h[0])
            lc_gate2 = threading.Barrier(lc_T + 1)
            def lc_worker2(i):
                lc_gate2.wait()
                for r in range(lc_per):
                    fs.lin_comb_g1(lc_pts, blobs_h[(i + r) % 64])
            lc_ths = [threading.Thread(target=lc_worker2, args=(i,)) for i in range(lc_T)]
            [t.start() for t in lc_ths]
            lc_gate2.wait()
            lc_t0 = time.perf_counter()
            [t.join() for t in lc_ths]
            lincomb["caller_supplied_same_points_from_64_threads_per_s"] = lc_T * lc_per / (time.perf_counter() - lc_t0)

            # --- single-call latencies through the host-buffer entry points (ms)
            # median of the timed calls (a one-off ~50 ms host / driver hiccup somewhere in this block was seen to land in one of the ten
            # calls of one entry or another and to triple its mean); the slowest call of each entry is reported beside it
            lat_max = {}
            def lat(name, fn, reps):
                for _ in range(3):
                    fn()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append((time.perf_counter() - t0) * 1e3)
                lat_max[name] = max(ts)
                return float(np.median(ts))
            one = blobs_h[0]
            # caller-supplied points (bls.LinCombG1's signature): a NEW set on every call takes the one-shot bucket pipeline; the SAME set again and again (what the
            # reference's callers do with their setup) is promoted to a cached set by its third sighting and then walks a table (capi_core.hip, lincomb_promo)
            fresh_sets = [np.ascontiguousarray(np.roll(setup, k + 1, axis=0)) for k in range(13)]
            fresh_it = iter(range(1 << 30))
            same_set = np.ascontiguousarray(np.roll(setup, 77, axis=0))
            promo_before = fs.lincomb_promotions()
            latency = {"CommitToPoly_4096_ms": lat("CommitToPoly", lambda: ks.commit_to_poly(one), 30),
                       "ComputeProofSingle_4096_ms": lat("ComputeProofSingle", lambda: ks.compute_proof_single(one, 17), 30),
                       "LinCombG1_4096_one_shot_ms": lat("LinCombG1_one_shot", lambda: fs.lin_comb_g1(fresh_sets[next(fresh_it) % 13], one), 10),
                       "LinCombG1_4096_same_points_again_ms": lat("LinCombG1_same_points_again", lambda: fs.lin_comb_g1(same_set, one), 10),
                       "LinCombG1_4096_cached_points_ms": lat("LinCombG1_cached", lambda: pts.lin_comb(one), 10),
                       "FFTG1_4096_ms": lat("FFTG1", lambda: fs.fft_g1(setup, False), 7)}
            latency["statistic"] = "median of the timed calls after 3 warm-up calls"
            promo_after = fs.lincomb_promotions()
            latency["LinCombG1_promotion"] = {"sets_promoted": promo_after[0] - promo_before[0], "calls_served_by_a_promoted_set": promo_after[1] - promo_before[1],
                                              "note": "the one-shot row sees 13 different point sets in turn (none repeats within the handle's memory of 6): never promoted; the same-points row is "
                                                      "promoted during its warm-up calls (third sighting) and every timed call compares its 590 KB of points with the kept copy before it runs"}
            del fresh_sets
            # the reference's functions return Jacobian points with whatever Z the additions left; with kzg_hip_kzg_set_projective_outputs the library does the same
            # (no inversion per result).  Default (above): normalised, Z = one.  Checked here: the projective results are the same group elements.
            want_c, want_p = ks.commit_to_poly(one), ks.compute_proof_single(one, 17)
            ks.set_projective_outputs(True)
            try:
                latency["projective_outputs"] = {
                    "CommitToPoly_4096_ms": lat("CommitToPoly_projective", lambda: ks.commit_to_poly(one), 30),
                    "ComputeProofSingle_4096_ms": lat("ComputeProofSingle_projective", lambda: ks.compute_proof_single(one, 17), 30),
                    "one_blob_per_call_from_64_threads_per_s": ks.bench_drop_in(blobs_h[:64].copy(), 64, 40)[0],
                    "same_group_elements": bool(np.array_equal(fs.to_compressed_g1(np.stack([ks.commit_to_poly(one), ks.compute_proof_single(one, 17)]).reshape(2, 3, 6)),
                                                               fs.to_compressed_g1(np.stack([want_c, want_p]).reshape(2, 3, 6)))),
                    "note": "kzg_hip_kzg_set_projective_outputs(ks, 1): results leave as un-normalised Jacobian images, the reference's own return type"}
            finally:
                ks.set_projective_outputs(False)
            latency["slowest_call_ms"] = lat_max
            pts.close()

        fk20 = None
        roofline_fk20 = None
        if not args.no_fk20:
            fk = kz.FK20SingleSettings(ks, 4096)
            FB = args.fk20_batch
            polys_h = mont_blobs(4 + rank * FB, FB)[:, :2048, :].copy()
            d_polys = torch.from_numpy(polys_h.view(np.int64)).cuda()
            d_proofs = torch.zeros((FB, 4096, 18), dtype=torch.int64, device="cuda")

            def fk_step():
                st = lib.kzg_hip_da_using_fk20_batch_dev(fk.h, d_polys.data_ptr(), 2048, FB, d_proofs.data_ptr(), stream)
                if st:
                    raise RuntimeError("da_using_fk20_batch_dev status %d" % st)

            fk_step()
            torch.cuda.synchronize()
            fk_ok = None
            if rank == 0:   # self-check of the timed path: polynomial 0 is blob(seed 4)[:2048], byte-pinned by the oracle (tests/golden/fk20_pins.json)
                p0 = d_proofs[0].cpu().numpy().view(np.uint64).reshape(4096, 3, 6)
                fk_ok = hashlib.sha256(fs.to_compressed_g1(p0).tobytes()).hexdigest() == pins["config4a_da_using_fk20_seed4"]["sha256"]
                if not fk_ok:
                    raise SystemExit("bench self-check failed: FK20 proofs of blob(seed 4) do not match the byte pin")
            fsteps = max(1, args.steps // 2)
            fsecs = timed_steps(fk_step, fsteps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
            rows_ok = all_ranks_ok(check_proof_rows(fs, d_proofs, FB, 4096, d_polys, 2048, fk.da_using_fk20, 0xFA20 + 2 * rank))
            if not rows_ok:
                raise SystemExit("bench self-check failed: the random linear combination over all %d x 4096 FK20 proofs of the timed step does not match" % FB)
            fk20 = {"metric": "FK20 all-proofs/s (DAUsingFK20, 2048 coeffs -> 4096 proofs, scale 12)",
                    "value": FB * world * fsteps / fsecs, "batch_per_gpu": FB,
                    "ms_per_all_proofs": fsecs / fsteps / FB * 1e3, "self_check_byte_pin": fk_ok,
                    "self_check": {"rows_checked": FB, "proofs_checked": FB * 4096,
                                   "method": "sum_b rho_b sum_j sigma_j proof_b[j] == sum_j sigma_j DAUsingFK20(sum_b rho_b p_b)[j] (right side on the one-polynomial path)"}}
            if not args.no_extras and world == 1:
                for _ in range(3):
                    fk.da_using_fk20(polys_h[0])
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    fk.da_using_fk20(polys_h[0])
                    ts.append((time.perf_counter() - t0) * 1e3)
                fk20["DAUsingFK20_single_call_ms"] = float(np.median(ts))      # median of 5 calls, like the `latency` block
                # a few polynomials per host-buffer call (four lanes per butterfly up to 8 polynomials: go-kzg_amd/csrc/g1_quad.hpp)
                small = {}
                for nb in (2, 8, 16, 32):
                    if nb > polys_h.shape[0]:
                        continue
                    fk.da_using_fk20_batch(polys_h[:nb])
                    t0 = time.perf_counter()
                    for _ in range(3):
                        fk.da_using_fk20_batch(polys_h[:nb])
                    small[str(nb)] = (time.perf_counter() - t0) / 3 * 1e3
                fk20["DAUsingFK20_host_batch_ms"] = small
                # the reference's API is one polynomial per call (fk20_single.go:176-196): 64 host threads calling it concurrently are merged
                # into batched launches by the library (Python threads here: ctypes releases the GIL for the ~50 ms a call blocks)
                import threading
                TT, per = 64, 6
                gate = threading.Barrier(TT + 1)
                def fk_worker(i):
                    gate.wait()
                    for r in range(per):
                        fk.da_using_fk20(polys_h[(i + r) % len(polys_h)])
                for _ in range(4):
                    fk.da_using_fk20(polys_h[0])                               # the staging buffers of the coalescer exist (pinned on first use)
                ths = [threading.Thread(target=fk_worker, args=(i,)) for i in range(TT)]
                [t.start() for t in ths]
                gate.wait()
                t0 = time.perf_counter()
                [t.join() for t in ths]
                fk20["DAUsingFK20_from_64_threads_per_s"] = TT * per / (time.perf_counter() - t0)
            # roofline of the FK20 half: HIP events around every launch of the dominant kernel (k_g1_fft_stage, 24 launches per step:
            # 12 radix-2 stages x 2 transforms), separate un-timed pass
            lib.kzg_hip_prof_reset(fs.h, 1)
            fk_step()
            torch.cuda.synchronize()
            tot2, cnt2 = C.c_double(0), C.c_uint64(0)
            lib.kzg_hip_prof_read(fs.h, b"g1_fft_stage", C.byref(tot2), C.byref(cnt2))
            tot3, cnt3 = C.c_double(0), C.c_uint64(0)
            lib.kzg_hip_prof_read(fs.h, b"fb_mul_vec", C.byref(tot3), C.byref(cnt3))
            lib.kzg_hip_prof_reset(fs.h, 0)
            if cnt2.value:
                kern_s = tot2.value * 1e-3                          # all stage launches of one step (FB polynomials)
                alg = FB * FK20_BYTES                               # 851 968 B per DAUsingFK20 (poly + xExtFFT + proofs)
                # multiply-adds of the stage kernel per polynomial: 2 transforms x 20 481 twiddle multiplications (width-5 NAF GLV with
                # an affine 8-entry table: ~126 doublings x 1963 + ~42.7 mixed additions x 3315 + 33.1k for the co-Z table (78M + 26S incl. its
                # normalisation and the beta x of every entry; the binary-GCD inversion has no multiplies) + 24 576 shared (x + wy, x - wy) x 7384
                # Fused pipeline (22 stage launches per step): the first two stages of the inverse transform are inside the fixed-base Toeplitz
                # stage, 2047 + 2046 twiddle multiplications and 4096 butterflies fewer.
                per_mul = 126 * 1963 + 42.7 * 3315 + 33100
                fused = int(cnt2.value) == 22
                mads_unit = ((2 * 20481 - 4093) * per_mul + (2 * 24576 - 4096) * 7384) if fused else (2 * 20481 * per_mul + 2 * 24576 * 7384)
                pf = pmc.get("k_g1_fft_stage", {})
                # the committed counter passes ran the 512-polynomial step; every launch is lane-per-butterfly with identical work per polynomial,
                # so per-step counts scale with the batch (stated in traffic_source)
                psc = (FB / pf["batch"]) if pf.get("batch") else None
                roofline_fk20 = {"bound": "hbm", "kernel": "k_g1_fft_stage", "achieved": alg / kern_s * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg / kern_s * 1e-9 / HBM_PEAK_GBS, "launches_per_step": int(cnt2.value), "avg_launch_ms": tot2.value / cnt2.value,
                                 "kernel_ms_per_all_proofs": tot2.value / FB, "algorithmic_bytes_per_step": alg,
                                 "traffic": (pf.get("fetch_bytes_per_step", 0) + pf.get("write_bytes_per_step", 0)) * psc if psc else None,
                                 "traffic_source": ((pmc.get("_file") + " (same launch shape)") if psc == 1 else "%s (counters of the %d-polynomial step x %g)" % (pmc.get("_file"), pf["batch"], psc)) if psc else None,
                                 "profile_avg_ms": profile_avg_ms("k_g1_fft_stage", FB * 2048, 256)[0], "profile_source": profile_avg_ms("k_g1_fft_stage", FB * 2048, 256)[1],
                                 "share_of_step": kern_s / (fsecs / fsteps), "table_walk_ms_per_step": tot3.value if cnt3.value else None,
                                 "mac": {"mads_per_all_proofs": mads_unit, "achieved_Tmad_s": FB * mads_unit / kern_s * 1e-12, "measured_peak_Tmad_s": cal_mad * 1e-12,
                                         "frac": FB * mads_unit / kern_s / cal_mad},
                                 "counters": {k: pf[k] for k in ("valu_insts_per_step", "sq_wait_inst_any", "sq_active_inst_any", "sq_busy_cycles", "scratch_bytes_per_lane") if k in pf},
                                 "issue": ({"issue_model_ms_per_step": (FB * mads_unit / cal_mad + max(pf["valu_insts_per_step"] * psc * 64.0 - FB * mads_unit, 0.0) / cal_add) * 1e3,
                                            "frac_of_kernel_time_explained": (FB * mads_unit / cal_mad + max(pf["valu_insts_per_step"] * psc * 64.0 - FB * mads_unit, 0.0) / cal_add) / kern_s}
                                           if psc and "valu_insts_per_step" in pf else None),
                                 "pipeline": "Toeplitz stage fused with two DIF stages of the inverse transform (k_fb_mul_vec_dif2), 10 DIF + 12 DIT stage launches" if fused else "24 stage launches (unfused)",
                                 "note": "one step = %d polynomials; the kernel is launched once per radix-2 stage; integer-issue-bound like the table walk" % FB}
            if use_dist:
                # the north star's "RCCL all-gather of proof points over xGMI": every rank ends up with the proofs of 32 blobs of every
                # rank (up to 32 x 4096 x 144 B = 18.9 MB per rank).  Reported beside the throughput; a failure must not cost the bench line.
                try:
                    from gokzg_amd import multi_gpu as mg
                    part = d_proofs[:32].contiguous()
                    gsecs = timed_steps(lambda: mg.all_gather_proofs(part), 5, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                    gathered = mg.all_gather_proofs(part)
                    cnt_ = part.shape[0]
                    ok_ = bool(gathered.shape[0] == world * cnt_ and torch.equal(gathered[rank * cnt_:(rank + 1) * cnt_], part))
                    fk20["all_gather_proofs"] = {"ms": gsecs / 5 * 1e3, "bytes_per_rank": int(part.numel() * 8), "ranks": world,
                                                 "GB_s_out_per_rank": part.numel() * 8 * (world - 1) / (gsecs / 5) * 1e-9, "own_slice_intact": ok_}
                    rccl["all_gather_proofs_ms"] = gsecs / 5 * 1e3
                    rccl["all_gather_proofs_bytes_per_rank"] = int(part.numel() * 8)
                except Exception as e:                           # noqa: BLE001
                    fk20["all_gather_proofs"] = {"error": "%s: %s" % (type(e).__name__, e)}
            fk.close()

        if not args.no_fk20 and args.fk20_4096_batch > 0:
            # The metric's literal input: a 4096-ELEMENT blob (BASELINE config 4b).  FK20Single (fk20_single.go:122-137) needs a domain of twice
            # the polynomial's length: scale 13, 8192-point setup [s^i]G1 of the reference's test secret (GenerateTestingSetup on the device).
            # Two forms on the same settings: FK20Single 4096 coefficients -> 4096 proofs (131 072 MulG1 in the reference, against 81 920 for
            # config 4a) and DAUsingFK20 4096 -> 8192 proofs.
            fs13 = kz.FFTSettings(13, device=local)
            sec13 = np.frombuffer((S_TEST * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4)
            ks13 = kz.KZGSettings(fs13, fs13.generate_testing_setup_g1(sec13, 8192))
            fk13 = kz.FK20SingleSettings(ks13, 8192)
            QB = args.fk20_4096_batch
            q_h = mont_blobs(1 + rank * QB, QB)                     # row 0 of rank 0 is blob(seed 1): config 4b's pinned polynomial
            d_q = torch.from_numpy(q_h.view(np.int64)).cuda()
            d_qp = torch.zeros((QB, 4096, 18), dtype=torch.int64, device="cuda")
            d_qda = torch.zeros((QB, 8192, 18), dtype=torch.int64, device="cuda")

            def q_step():
                st = lib.kzg_hip_fk20_single_batch_dev(fk13.h, d_q.data_ptr(), 4096, QB, d_qp.data_ptr(), stream)
                if st:
                    raise RuntimeError("fk20_single_batch_dev status %d %s" % (st, lib.kzg_hip_last_error().decode()))

            def qda_step():
                st = lib.kzg_hip_da_using_fk20_batch_dev(fk13.h, d_q.data_ptr(), 4096, QB, d_qda.data_ptr(), stream)
                if st:
                    raise RuntimeError("da_using_fk20_batch_dev (4096) status %d %s" % (st, lib.kzg_hip_last_error().decode()))

            q_step(); qda_step()
            torch.cuda.synchronize()
            q_ok = None
            if rank == 0:
                h1 = hashlib.sha256(fs13.to_compressed_g1(d_qp[0].cpu().numpy().view(np.uint64).reshape(4096, 3, 6)).tobytes()).hexdigest()
                h2 = hashlib.sha256(fs13.to_compressed_g1(d_qda[0].cpu().numpy().view(np.uint64).reshape(8192, 3, 6)).tobytes()).hexdigest()
                q_ok = h1 == pins["config4b_fk20_single_seed1"]["sha256"] and h2 == pins["config4b_da_using_fk20_seed1"]["sha256"]
                if not q_ok:
                    raise SystemExit("bench self-check failed: FK20 proofs of the 4096-element blob(seed 1) do not match the config-4b byte pins")
            qsteps = max(1, args.steps // 2)
            qsecs = timed_steps(q_step, qsteps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
            qdsecs = timed_steps(qda_step, qsteps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
            rows_ok = all_ranks_ok(check_proof_rows(fs13, d_qp, QB, 4096, d_q, 4096, fk13.fk20_single, 0xFB20 + 2 * rank)
                                   and check_proof_rows(fs13, d_qda, QB, 8192, d_q, 4096, fk13.da_using_fk20, 0xFC20 + 2 * rank))
            if not rows_ok:
                raise SystemExit("bench self-check failed: the random linear combination over all FK20 proofs of the 4096-element blobs does not match")
            fk20_4096 = {"metric": "FK20 all-proofs/s on 4096-element blobs (FK20Single, 4096 coeffs -> 4096 proofs, scale 13; BASELINE config 4b)",
                         "value": QB * world * qsteps / qsecs, "batch_per_gpu": QB, "ms_per_all_proofs": qsecs / qsteps / QB * 1e3,
                         "da_using_fk20_4096_to_8192": {"value": QB * world * qsteps / qdsecs, "ms_per_all_proofs": qdsecs / qsteps / QB * 1e3},
                         "rate_over_config_4a": (QB * world * qsteps / qsecs) / fk20["value"] if fk20 else None,
                         "reference_work_ratio_4b_over_4a": 131072 / 81920,
                         "self_check_byte_pins": q_ok,
                         "self_check": {"rows_checked": QB, "proofs_checked": QB * (4096 + 8192),
                                        "method": "random linear combination over every proof of both forms against the one-polynomial path; row 0 against the config-4b pins"}}
            if not args.no_extras and world == 1:
                def med(fn, reps=5):
                    for _ in range(3):
                        fn()
                    ts_ = []
                    for _ in range(reps):
                        t0_ = time.perf_counter()
                        fn()
                        ts_.append((time.perf_counter() - t0_) * 1e3)
                    return float(np.median(ts_))
                fk20_4096["FK20Single_4096_single_call_ms"] = med(lambda: fk13.fk20_single(q_h[0]))
                fk20_4096["DAUsingFK20_4096_single_call_ms"] = med(lambda: fk13.da_using_fk20(q_h[0]))
            # roofline of the stage kernel at this shape: HIP events around every stage launch of one FK20Single step (separate un-timed pass)
            lib.kzg_hip_prof_reset(fs13.h, 1)
            q_step()
            torch.cuda.synchronize()
            tq, cq = C.c_double(0), C.c_uint64(0)
            lib.kzg_hip_prof_read(fs13.h, b"g1_fft_stage", C.byref(tq), C.byref(cq))
            lib.kzg_hip_prof_reset(fs13.h, 0)
            if cq.value:
                kq = tq.value * 1e-3
                per_mul = 126 * 1963 + 42.7 * 3315 + 33100          # multiply-adds of one twiddle multiplication (see the 4a block)
                # a transform of N points multiplies in N/2 log2 N - (N - 1) butterflies (twiddle 1 does not multiply); inverse transform of 8192
                # points with its two widest stages inside the Toeplitz stage (4095 + 4094 multiplications, 8192 butterflies fewer), forward of 4096
                fusedq = int(cq.value) == 23
                muls = (45057 - 8189 + 20481) if fusedq else (45057 + 20481)
                bfly = (13 * 4096 - 8192 + 12 * 2048) if fusedq else (13 * 4096 + 12 * 2048)
                mads_q = muls * per_mul + bfly * 7384
                algq = QB * FK20_4096_BYTES
                roofline_fk20_4096 = {"bound": "hbm", "kernel": "k_g1_fft_stage", "achieved": algq / kq * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": algq / kq * 1e-9 / HBM_PEAK_GBS, "traffic": None, "launches_per_step": int(cq.value), "avg_launch_ms": tq.value / cq.value,
                                      "algorithmic_bytes_per_step": algq, "algorithmic_bytes_per_unit": FK20_4096_BYTES,
                                      "profile_avg_ms": profile_avg_ms("k_g1_fft_stage", QB * 4096, 256)[0], "profile_source": profile_avg_ms("k_g1_fft_stage", QB * 4096, 256)[1],
                                      "kernel_ms_per_all_proofs": tq.value / QB, "share_of_step": kq / (qsecs / qsteps),
                                      "mac": {"mads_per_all_proofs": mads_q, "achieved_Tmad_s": QB * mads_q / kq * 1e-12, "measured_peak_Tmad_s": cal_mad * 1e-12,
                                              "frac": QB * mads_q / kq / cal_mad},
                                      "pipeline": ("Toeplitz stage fused with two DIF stages (k_fb_mul_vec_dif2), 11 DIF launches on 8192 points + 12 DIT launches on the 4096 even positions"
                                                   if fusedq else "%d stage launches (unfused)" % int(cq.value)),
                                      "note": "one step = %d polynomials of 4096 coefficients (FK20Single at scale 13)" % QB}
            del d_q, d_qp, d_qda
            fk13.close(); ks13.close(); fs13.close()

        fk20m = None
        if not args.no_fk20 and args.fk20_multi_batch > 0:
            # BASELINE config 5: FK20Multi, scale 16 (n2 = 65536, 32768 coefficients), chunk length 16 -> 4096 coset proofs per polynomial.
            # Setup [s^i]G1 for the reference's test secret is generated on the device (GenerateTestingSetup).
            s_test = 1927409816240961209460912649124
            fs16 = kz.FFTSettings(16, device=local)
            sec = np.frombuffer((s_test * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4)
            ks16 = kz.KZGSettings(fs16, fs16.generate_testing_setup_g1(sec, 65536))
            fkm = kz.FK20MultiSettings(ks16, 65536, 16)
            MB = args.fk20_multi_batch
            mp_h = mont_blobs(5 + rank * MB, MB, n=32768)
            d_mp = torch.from_numpy(mp_h.view(np.int64)).cuda()
            d_mproofs = torch.zeros((MB, 4096, 18), dtype=torch.int64, device="cuda")

            def fkm_step():
                st = lib.kzg_hip_da_using_fk20_multi_batch_dev(fkm.h, d_mp.data_ptr(), 32768, MB, d_mproofs.data_ptr(), stream)
                if st:
                    raise RuntimeError("da_using_fk20_multi_batch_dev status %d" % st)

            fkm_step()
            torch.cuda.synchronize()
            fkm_ok = None
            if rank == 0:   # polynomial 0 is blob(seed 5, 32768): all 4096 coset proofs byte-pinned by the oracle's full-size run
                p0 = d_mproofs[0].cpu().numpy().view(np.uint64).reshape(4096, 3, 6)
                fkm_ok = hashlib.sha256(fs16.to_compressed_g1(p0).tobytes()).hexdigest() == pins["config5_da_using_fk20_multi_seed5"]["sha256"]
                if not fkm_ok:
                    raise SystemExit("bench self-check failed: FK20Multi proofs of blob(seed 5) do not match the byte pin")
            msteps = max(1, args.steps // 2)
            msecs = timed_steps(fkm_step, msteps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
            if not all_ranks_ok(check_proof_rows(fs16, d_mproofs, MB, 4096, d_mp, 32768, fkm.da_using_fk20_multi, 0xFD20 + 2 * rank)):
                raise SystemExit("bench self-check failed: the random linear combination over all %d x 4096 FK20Multi proofs of the timed step does not match" % MB)
            fk20m = {"metric": "FK20Multi all-coset-proofs/s (DAUsingFK20Multi, scale 16, chunk 16: 32768 coeffs -> 4096 proofs)",
                     "value": MB * world * msteps / msecs, "batch_per_gpu": MB, "ms_per_all_proofs": msecs / msteps / MB * 1e3, "self_check_byte_pin": fkm_ok,
                     "self_check": {"rows_checked": MB, "proofs_checked": MB * 4096, "method": "random linear combination over every coset proof against the one-polynomial path"}}
            if use_dist or args.sharded_fk20_multi:
                # ONE FK20Multi with its Toeplitz stage sharded over the ranks and an RCCL all-gather of the 144-byte point slices
                # (BASELINE config 5, go-kzg_amd/multi_gpu.py).  A latency figure, reported beside the throughput numbers; a failure
                # here must not cost the bench line.
                try:
                    from gokzg_amd import multi_gpu as mg
                    be = mg.HipFK20MultiBackend(fkm)
                    if use_dist:                                 # every rank works on rank 0's polynomial
                        dist.broadcast(d_mp[0], src=0)
                    one = d_mp[0].contiguous()
                    ref = torch.empty((4096, 18), dtype=torch.int64, device="cuda")
                    st = lib.kzg_hip_da_using_fk20_multi_batch_dev(fkm.h, one.data_ptr(), 32768, 1, ref.data_ptr(), stream)
                    got = mg.da_using_fk20_multi_sharded(be, one, 32768, 4096)
                    torch.cuda.synchronize()
                    same = bool(st == 0 and torch.equal(got, ref))
                    ssecs = timed_steps(lambda: mg.da_using_fk20_multi_sharded(be, one, 32768, 4096), 3, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                    fk20m["sharded_one_polynomial"] = {"ms": ssecs / 3 * 1e3, "ranks": world, "matches_unsharded": same,
                                                       "collective": "all_gather of 4096 x 144 B point slices (RCCL)" if use_dist else "none (1 rank)"}
                    if rccl is not None:
                        rccl["sharded_one_polynomial_ms"] = ssecs / 3 * 1e3
                        rccl["sharded_one_polynomial_matches_unsharded"] = same
                except Exception as e:                           # noqa: BLE001
                    fk20m["sharded_one_polynomial"] = {"error": "%s: %s" % (type(e).__name__, e)}
            fkm.close(); ks16.close(); fs16.close()

        ref_benches = None
        if not args.no_fk20:
            # The three transforms the reference publishes numbers for (BENCH.md, Kilic column, Ryzen 9 5950X, 1 thread), scale 12,
            # device-resident batches: FFT over F_r (:43), FFT over G1 (:55), DAS FFT extension (:31).
            def rate(fn, units, reps):
                secs_ = timed_steps(fn, reps, 1, torch.cuda.synchronize, barrier, max_over_ranks)
                return units * world * reps / secs_

            FB = 1024
            d_fr = torch.from_numpy(mont_blobs(12 + rank * FB, FB).view(np.int64)).cuda()
            d_fr_out = torch.empty_like(d_fr)

            def fr_step():
                st = lib.kzg_hip_fft_fr_batch_dev(fs.h, d_fr.data_ptr(), N_COEFF, FB, 0, d_fr_out.data_ptr(), stream)
                if st:
                    raise RuntimeError("fft_fr_batch_dev status %d" % st)

            d_das = d_fr[:, :2048, :].contiguous()

            def das_step():
                st = lib.kzg_hip_das_fft_extension_batch_dev(fs.h, d_das.data_ptr(), 2048, FB, stream)
                if st:
                    raise RuntimeError("das_fft_extension_batch_dev status %d" % st)

            GB = 64
            d_g1 = torch.from_numpy(setup.view(np.int64).reshape(1, 4096, 18)).cuda().repeat(GB, 1, 1).contiguous()
            d_g1_out = torch.empty_like(d_g1)

            def g1_step():
                st = lib.kzg_hip_fft_g1_batch_dev(fs.h, d_g1.data_ptr(), N_COEFF, GB, 0, d_g1_out.data_ptr(), stream)
                if st:
                    raise RuntimeError("fft_g1_batch_dev status %d" % st)

            lib.kzg_hip_prof_reset(fs.h, 1)
            r_fr, r_das, r_g1 = rate(fr_step, FB, 5), rate(das_step, FB, 5), rate(g1_step, GB, 2)
            torch.cuda.synchronize()

            def fr_roofline(prof_name, pmc_key, kernel, alg_bytes_unit, mads_unit, what, lanes_per_unit):
                """HBM roofline of an LDS-resident F_r transform kernel (bytes in + bytes out per transform), its multiply-add rate
                against the v_mad_u64_u32 rate measured in this run, and the committed counter pass of the same launch shape"""
                t_, c_ = C.c_double(0), C.c_uint64(0)
                lib.kzg_hip_prof_read(fs.h, prof_name, C.byref(t_), C.byref(c_))
                if not c_.value:
                    return None
                avg = t_.value / c_.value * 1e-3
                pk = pmc.get(pmc_key, {})
                same = pk.get("batch") == FB
                out = {"bound": "hbm", "kernel": kernel, "achieved": FB * alg_bytes_unit / avg * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": FB * alg_bytes_unit / avg * 1e-9 / HBM_PEAK_GBS, "avg_launch_ms": avg * 1e3, "launches_timed": int(c_.value), "per_launch": what,
                       "algorithmic_bytes_per_launch": FB * alg_bytes_unit,
                       "traffic": (pk["fetch_bytes_per_launch"] + pk["write_bytes_per_launch"]) if same else None,
                       "traffic_source": (pmc.get("_file") + " (same launch shape; FETCH_SIZE x 2 per the guide's correction for wide coalesced reads)") if same else None,
                       "mac": {"mads_per_launch": FB * mads_unit, "achieved_Tmad_s": FB * mads_unit / avg * 1e-12, "measured_peak_Tmad_s": cal_mad * 1e-12,
                               "frac": FB * mads_unit / avg / cal_mad}}
                out["profile_avg_ms"], out["profile_source"] = profile_avg_ms(kernel, FB * lanes_per_unit, lanes_per_unit)
                if same:
                    valu = pk["valu_insts_per_launch"]
                    model = FB * mads_unit / cal_mad + max(valu * 64.0 - FB * mads_unit, 0.0) / cal_add
                    out["issue"] = {"valu_wave_insts_per_launch": valu, "mad_share_of_insts": FB * mads_unit / 64.0 / valu, "issue_model_ms": model * 1e3,
                                    "frac_of_launch_explained": model / avg, "lds_bank_conflict_share": pk["lds_bank_conflict"] / max(pk["lds_idx_active"], 1.0),
                                    "wave_cycles_split": {"active": pk["sq_active_inst_any"] / pk["sq_wave_cycles"], "issue_stall": pk["sq_wait_inst_any"] / pk["sq_wave_cycles"],
                                                          "parked_waitcnt_or_barrier": pk["sq_wait_any"] / pk["sq_wave_cycles"]}}
                return out
            # multiply-adds per transform: 1024 lanes x (21 products of 153 + 4 canonicalisations of 8); per extension: 512 lanes x (46 products of 153)
            roofline_fft_fr = fr_roofline(b"fr_fft4096", "k_fr_fft4096_r4", "k_fr_fft4096_r4", 2 * 4096 * 32, 1024 * (21 * 153 + 4 * 8), "%d forward transforms of 4096 points" % FB, 1024)
            roofline_das = fr_roofline(b"das_ext2048", "k_das_ext2048_r4", "k_das_ext2048_r4", 2 * 2048 * 32, 512 * (46 * 153 + 5 * 9), "%d extensions of 2048 values" % FB, 512)
            lib.kzg_hip_prof_reset(fs.h, 0)
            # the same transform with 4096 rows per launch (16 rounds of workgroups instead of 4: the launch's fixed costs and its last, partly empty round weigh less);
            # the rooflines above stay on the 1024-row launch, the shape of the committed counter pass
            FB4 = 4096
            d_fr4 = d_fr.repeat(FB4 // FB, 1, 1).contiguous()
            d_fr4_out = torch.empty_like(d_fr4)

            def fr4_step():
                st = lib.kzg_hip_fft_fr_batch_dev(fs.h, d_fr4.data_ptr(), N_COEFF, FB4, 0, d_fr4_out.data_ptr(), stream)
                if st:
                    raise RuntimeError("fft_fr_batch_dev status %d" % st)
            r_fr4 = rate(fr4_step, FB4, 5)
            del d_fr4, d_fr4_out
            ref_benches = {
                "fft_fr_scale12_per_s": {"value": r_fr, "reference_published": 1e9 / 1911871, "source": "BENCH.md:43 (Kilic, 5950X, 1 thread)", "batch": FB,
                                         "value_4096_rows_per_launch": r_fr4},
                "das_fft_extension_scale12_per_s": {"value": r_das, "reference_published": 1e9 / 1169011, "source": "BENCH.md:31", "batch": FB},
                "fft_g1_scale12_per_s": {"value": r_g1, "reference_published": 1e9 / 3745748396, "source": "BENCH.md:55", "batch": GB},
            }

    except Exception as e:                                   # noqa: BLE001
        if world > 1:
            raise
        import traceback
        secondary_error = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc().strip().splitlines()[-3:])
    if roofline is not None:                                  # the other kernels' rooflines live under the headline kernel's
        roofline["secondary"] = {"fk20": roofline_fk20, "fk20_4096": roofline_fk20_4096, "fft_fr": roofline_fft_fr, "das_ext": roofline_das}
    # the multi-device handle of the C ABI, in a child process once this one has released its tables (every rank releases; rank 0 runs it)
    in_process = None
    ks.close()
    fs.close()
    del d_blobs, d_out
    torch.cuda.empty_cache()
    if not args.no_in_process and not args.no_fk20 and not os.environ.get("KZG_BENCH_FAIL_SECONDARY"):
        barrier()                                             # every rank has released its tables
        if rank == 0:
            in_process = run_in_process_child(list(range(world)) if (world > 1 and torch.cuda.device_count() >= world) else [local])
        if use_dist:
            # the other ranks wait on the rendezvous store, on the CPU: a collective barrier would keep a spinning kernel on the very GPUs the child is timing
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("kzg_bench_in_process_done", "1")
            else:
                store.wait(["kzg_bench_in_process_done"], datetime.timedelta(seconds=600))
    if rccl is not None and isinstance(in_process, dict) and "error" not in in_process:
        # the same job through ONE process: the multi-device handle of the C ABI over the ranks' devices (full object under `in_process`)
        try:
            c5 = in_process["da_using_fk20_multi_one_polynomial_scale16"]
            rows5 = {k: v for k, v in c5.items() if k.endswith("_entries")}
            rccl["in_process_multi_device_handle"] = {
                "devices": in_process["devices"], "transport": in_process["transport"],
                "commit_to_poly_batch_per_s_host_buffers": in_process["commit_to_poly_batch"]["commitments_per_s"],
                "scaling_vs_one_device": in_process["commit_to_poly_batch"]["scaling_vs_one_device"],
                "da_using_fk20_batch_per_s_host_buffers": in_process["da_using_fk20_batch"]["all_proofs_per_s"],
                "one_fk20_multi_scale16_ms": {k: {m_: v[m_]["ms"] for m_ in ("gather", "sharded") if m_ in v} for k, v in rows5.items()},
                "one_fk20_multi_scale16_unsharded_ms": c5.get("unsharded_one_device_ms")}
        except (KeyError, TypeError):
            pass
    if rank == 0:
        # BASELINE's metric has two halves; `value` is the commitments half, the FK20 half on the metric's literal 4096-element blob (config 4b)
        # rides at top level beside it with its own roofline, and once more as plain numbers inside `roofline` (a reader that keeps only the
        # contract's keys still sees both halves)
        fk_half = fk20_4096["value"] if isinstance(fk20_4096, dict) and "value" in fk20_4096 else None
        if roofline is not None:
            roofline["second_half_of_metric"] = {"metric": "FK20 all-proofs/s, 4096-element blob (FK20Single, 4096 coefficients -> 4096 proofs)", "value": fk_half,
                                                 "unit": "all-proofs/s", "roofline_frac": (roofline_fk20_4096 or {}).get("frac"),
                                                 "mac_frac": ((roofline_fk20_4096 or {}).get("mac") or {}).get("frac"),
                                                 "fk20_2048_to_4096_all_proofs_per_s": fk20["value"] if isinstance(fk20, dict) and "value" in fk20 else None}
        if isinstance(base, dict) and isinstance(base.get("all_cores"), dict):   # flat copies: nested objects of cpu_baseline are dropped by some readers
            base["value_all_cores"], base["cores_all"] = base["all_cores"].get("value"), base["all_cores"].get("cores")
        print(json.dumps({
            "metric": "KZG commitments/sec (CommitToPoly, 4096-element blob); FK20 half: value_fk20_4096", "value": value, "unit": "commitments/s",
            "value_fk20_4096": fk_half, "unit_fk20_4096": "all-proofs/s", "roofline_fk20_4096": roofline_fk20_4096,
            "cpu_baseline_fk20_4096": base.get("fk20_4096") if isinstance(base, dict) else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": secs / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "dtype_note": "30/32-bit limbs in u32 lanes, 64-bit accumulators (v_mad_u64_u32): 381-bit F_p and 255-bit F_r Montgomery arithmetic",
            "data": "synthetic",
            "config": {"workload": "CommitToPoly, 4096-coeff blobs, eth/trusted_setup.json monomial setup (s=1337), %d blobs/step/GPU resident in HBM, fixed-base table budget %g GB (library default 110 GB: signed 16-bit windows, 8 of them walked by both GLV halves of every scalar, 103 GB; see table_sweep)" % (B, args.table_gb),
                       "global_batch": B * world, "parallelism": "dp%d (independent blobs, no data-path collective)" % world},
            "rccl": rccl, "roofline": roofline, "cpu_baseline": base, "self_check": self_check, "batch_sweep": batch_sweep, "table_sweep": table_sweep, "drop_in": drop_in,
            "lincomb": lincomb, "latency": latency, "fk20": fk20, "fk20_4096": fk20_4096, "fk20_multi": fk20m, "reference_benchmarks": ref_benches, "in_process": in_process,
            "secondary_error": secondary_error,
        }))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
