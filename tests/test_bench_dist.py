"""bench.py's host logic on CPU: synthetic input synthesis matches SURVEY.md 8(d) (== the oracle's generator), and the
N > 1 harness (barrier, max-over-ranks timing, unit sharding) runs with world_size 2 over gloo."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)




def _free_port():
    """a TCP port nobody listens on right now (the kernel picks it): fixed or pid-derived ports can collide with another job on the host and
    turn the rendezvous into a 30-minute wait"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _collect(q, procs, n, timeout):
    """n results from the workers' queue; gives up as soon as every worker has exited without delivering (a crashed rank must fail the
    test at once, not after the full timeout)"""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            if not any(p.is_alive() for p in procs):
                try:
                    while len(out) < n:
                        out.append(q.get(timeout=1))             # results written just before the exit
                except queue.Empty:
                    break
    assert len(out) == n, "workers delivered %d of %d results (exit codes %s)" % (len(out), n, [p.exitcode for p in procs])
    return out

def _reap(procs, timeout):
    """workers have already delivered their results through the queue: give them `timeout` seconds to leave on their own, then kill what is
    left (a rank that lingers in device teardown must not outlive the test run: an orphan keeps the caller's stdout pipe open); a worker
    that DID exit has to have exited cleanly"""
    for p in procs:
        p.join(timeout=timeout)
        if p.is_alive():
            p.kill()
            p.join(timeout=30)
        else:
            assert p.exitcode == 0, p.exitcode

def test_splitmix_blobs_match_oracle_generator():
    import bench
    from oracle import koracle as ko
    got = bench.splitmix_blobs(1, 2, n=64)
    assert np.array_equal(got[0], ko.synthetic_blob(1, 64))
    assert np.array_equal(got[1], ko.synthetic_blob(2, 64))


def test_splitmix_blobs_le32_vectorised_matches_oracle_generator():
    import bench
    from oracle import koracle as ko
    got = bench.splitmix_blobs_le32(7, 2, n=128)
    for b in range(2):
        want = ko.fr_to_ints(ko.synthetic_blob(7 + b, 128))
        assert [int.from_bytes(got[b, i].tobytes(), "little") for i in range(128)] == want


def test_shard_units_cover_exactly():
    import bench
    for total in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_units(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import time
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))      # rank 1 is slower: the reported time must be the max over ranks

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    secs = bench.timed_steps(step, 5, 2, lambda: None, dist.barrier, max_over_ranks)
    # all-gather of per-rank byte slices in rank order (the FK20Multi position sharding uses exactly this)
    lo, hi = bench.shard_units(10, world, rank)
    mine = torch.arange(lo, hi, dtype=torch.uint8)
    sizes = [bench.shard_units(10, world, r) for r in range(world)]
    bufs = [torch.zeros(h - l, dtype=torch.uint8) for l, h in sizes]
    dist.all_gather(bufs, mine)
    q.put((rank, len(calls), secs, torch.cat(bufs).tolist()))
    dist.destroy_process_group()


def test_timed_steps_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs, 2, 120))
    _reap(procs, 60)
    assert [r[1] for r in res] == [7, 7]                  # W + K steps on every rank
    assert abs(res[0][2] - res[1][2]) < 1e-9             # both ranks report the same (max) time
    assert res[0][2] >= 5 * 0.02 * 0.9                    # ... which is the slow rank's
    assert res[0][3] == list(range(10)) == res[1][3]


@pytest.mark.gpu
def test_bench_two_ranks_under_torch_distributed_run():
    """bench.py under an external launcher for N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N),
    with two ranks on the ONE GPU of the test box: gloo moves the bytes (RCCL refuses two ranks on a device), everything else -- rank
    sharding of the blobs, barrier + max-over-ranks timing, the all-gather of proofs, the sharded FK20Multi with its byte comparison,
    the single JSON line from rank 0 -- is the N > 1 code path.  Small tables so that two ranks fit one device."""
    import json
    import subprocess
    env = dict(os.environ, KZG_BENCH_BACKEND="gloo", KZG_HIP_FK20_FB_BUDGET_GB="3", MASTER_ADDR="127.0.0.1")
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--fk20-batch", "8", "--fk20-multi-batch", "2",
           "--table-gb", "4", "--no-extras", "--fk20-4096-batch", "4", "--no-in-process"]
    # own session: on a timeout the whole process group (launcher + both ranks) is killed, nothing is left holding the device or the pipes
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        raise AssertionError("bench.py under torch.distributed.run did not finish in 600 s: " + err[-2000:])
    res = subprocess.CompletedProcess(cmd, proc.returncode, out, err)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["global_batch"] == 128
    assert d["cpu_baseline"] is None                                # rank 0 at N = 1 only
    assert d["fk20"]["self_check_byte_pin"] is True and d["fk20"]["all_gather_proofs"]["own_slice_intact"] is True
    assert d["fk20"]["all_gather_proofs"]["ranks"] == 2
    assert d["fk20_multi"]["self_check_byte_pin"] is True and d["fk20_multi"]["self_check"]["rows_checked"] == 2
    assert d["self_check"]["rows_checked"] == 64 and d["fk20"]["self_check"]["rows_checked"] == 8       # every output of the timed steps is checked
    assert d["fk20_4096"]["self_check_byte_pins"] is True and d["fk20_4096"]["value"] > 0              # the 4096-element blob (config 4b) is timed
    assert d["fk20_multi"]["sharded_one_polynomial"]["matches_unsharded"] is True, d["fk20_multi"]["sharded_one_polynomial"]


@pytest.mark.gpu
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it and no WORLD_SIZE in the environment (the shape of the driver's command
    line): bench.py starts its own two ranks, and the ONE line on stdout is rank 0's, carrying the `rccl` object (backend, ranks seen
    through an all-gather, per-rank table shape, the proof all-gather and the sharded polynomial).  Two ranks on the one test GPU, so the
    transport is gloo (KZG_BENCH_BACKEND); on an N-GPU node the same path runs on RCCL."""
    import json
    import signal
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KZG_BENCH_BACKEND="gloo", KZG_HIP_FK20_FB_BUDGET_GB="3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--fk20-batch", "8",
           "--fk20-multi-batch", "2", "--table-gb", "4", "--fk20-4096-batch", "4"]   # (with the in_process leg: rank 0 runs the child, rank 1 waits on the store)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        raise AssertionError("self-launched bench.py did not finish in 600 s: " + err[-2000:])
    assert proc.returncode == 0, err[-3000:]
    lines = out.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out[-2000:]      # stdout is exactly the JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["global_batch"] == 128
    r = d["rccl"]
    assert r["backend"] == "gloo" and r["self_launched"] is True and r["world_size"] == 2 and r["ranks_seen"] == [0, 1]
    assert [t["rank"] for t in r["tables_per_rank"]] == [0, 1] and all(t["windows"] > 0 for t in r["tables_per_rank"])
    assert r["all_gather_proofs_ms"] > 0 and r["sharded_one_polynomial_ms"] > 0 and r["sharded_one_polynomial_matches_unsharded"] is True
    assert d["fk20"]["self_check_byte_pin"] is True and d["fk20_multi"]["self_check_byte_pin"] is True
    ip = d["in_process"]                                       # the multi-device handle, timed in a child process by rank 0 (one GPU here: devices [0])
    assert "error" not in ip, ip
    assert ip["commit_to_poly_batch"]["vector_F"] is True and ip["da_using_fk20_batch"]["byte_pin_row0"] is True
    assert ip["da_using_fk20_multi_one_polynomial_scale16"]["2_entries"]["sharded"]["byte_pin"] is True


def test_self_launch_plumbing_without_a_gpu():
    """CPU box: `bench.py --gpus 2` still goes through its own launcher; both ranks refuse to run without a gfx950 device (no CPU
    fallback), the parent reports the failure with a non-zero exit code and prints NO JSON line."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_self_launches_its_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert res.stdout.strip() == ""
    assert "launching 2 ranks" in res.stderr and "needs an MI355X" in res.stderr
    assert "--nproc-per-node 2" in res.stderr and "127.0.0.1" in res.stderr


def test_bench_single_gpu_invocation_does_not_self_launch():
    """--gpus 1 (the default) never goes through the launcher"""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src
    assert callable(bench.self_launch)


@pytest.mark.gpu
def test_bench_line_survives_a_failing_secondary_leg():
    """the headline (commitments/s with roofline) is measured first; a failure in any later leg (table sweep, one-blob API, FK20 ...) must
    not cost the JSON line the driver reads: it is reported in `secondary_error`"""
    import json
    import subprocess
    env = dict(os.environ, KZG_BENCH_FAIL_SECONDARY="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--table-gb", "4", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["roofline"]["mac"]["frac"] > 0
    assert "KZG_BENCH_FAIL_SECONDARY" in d["secondary_error"] and d["fk20"] is None


def _error_paths(obj, path=""):
    """every place in a bench line where a leg swallowed an exception into {"error": ...}"""
    found = []
    if isinstance(obj, dict):
        if "error" in obj and obj["error"]:
            found.append("%s: %s" % (path or "<line>", obj["error"]))
        for k, v in obj.items():
            found += _error_paths(v, path + "/" + str(k))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            found += _error_paths(v, "%s[%d]" % (path, i))
    return found


def test_port_vs_published_runs_and_carries_the_three_ratios():
    """the calibration of the CPU port against the reference's published transforms (BENCH.md:31,43,55): in round 5 a missing import turned it into a swallowed
    NameError and nothing noticed; it is called here directly (tiny budget), and through cpu_baseline's own try/except the result must carry no `error`"""
    from benchlib import cpu
    r = cpu._port_vs_published(0.05)
    assert set(r["port_over_published"]) == {"fft_fr_scale12_ns", "das_fft_extension_scale12_ns", "fft_g1_scale12_ns"}
    assert all(0.05 < v < 100 for v in r["port_over_published"].values()), r["port_over_published"]
    assert r["mul_g1_port_us"] > 10


def test_cpu_baseline_fk20_estimate():
    """the FK20 half of the metric on the host: one oracle FK20Single on a small polynomial, scaled by the reference's MulG1 count"""
    from benchlib import cpu
    assert cpu.fft_g1_muls(4096) == 36864                                   # 1024 leaves x 16 + 10 levels x 2048
    assert cpu.fk20_single_muls(4096) == 131072                             # FK20Single on 4096 coefficients
    r = cpu.cpu_baseline_fk20(n_sample=16)
    assert r["is_estimate"] and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0
    assert r["mul_g1_per_fk20_4096"] == 131072
    assert not _error_paths(r)


def test_cpu_baseline_carries_no_swallowed_error():
    """cpu_baseline() end to end with a sub-second budget: every nested leg (one core, all cores, port_vs_published, fk20_4096) present, none an {"error": ...}"""
    import subprocess
    code = ("import json, sys; sys.path.insert(0, %r); from benchlib.cpu import cpu_baseline; print(json.dumps(cpu_baseline(0.3)))" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)       # own process: the all-cores leg forks
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert not _error_paths(d), _error_paths(d)
    assert d["value"] > 0 and d["all_cores"]["value"] > 0 and d["fk20_4096"]["value"] > 0 and "port_over_published" in d["port_vs_published"]


@pytest.mark.gpu
def test_bench_line_has_no_failed_leg():
    """the whole default bench (every leg, reduced sizes) on a box where everything should work: no leg may have swallowed an exception into an
    {"error": ...} object, `secondary_error` is null, and the dominant kernel's launches -- timed on the timed steps -- fit inside the steps"""
    import json
    import subprocess
    env = dict(os.environ, KZG_BENCH_CPU_BUDGET_S="0.5", KZG_HIP_FK20_FB_BUDGET_GB="8")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "512", "--table-gb", "9", "--fk20-batch", "16",
           "--fk20-multi-batch", "2", "--fk20-4096-batch", "4"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["secondary_error"] is None, d["secondary_error"]
    assert not _error_paths(d), _error_paths(d)
    r = d["roofline"]
    assert r["launches_timed"] == 3 and r["launch_over_step"] <= 1.002 and r["avg_launch_ms"] <= d["ms_per_step"] * 1.002
    assert d["cpu_baseline"]["port_vs_published"]["port_over_published"]["fft_fr_scale12_ns"] > 0
    assert d["cpu_baseline_fk20_4096"]["value"] > 0 and d["value_fk20_4096"] > 0


def test_roofline_arithmetic_reproduces_the_committed_profile():
    """benchlib/roofline.py on the numbers of profiles/r06_*: the contract's `achieved` = algorithmic bytes / average launch time of the dominant kernel (the row of
    that launch shape in the committed kernel trace), counters matched on kernel, length and table shape, the multiply-add and issue fractions from the same inputs"""
    import json
    from benchlib import roofline as rl
    shapes = json.load(open(os.path.join(ROOT, "profiles", "r06_kernel_shapes.json")))["rows"]
    row, = [r for r in shapes if r["kernel"].startswith("k_fb_accumulate") and r["grid"] == 4096 * 256 and r["workgroup"] == 256]
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc.json")))
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line_unprofiled.json")))
    avg_s = row["avg_us"] * 1e-6
    table = (16, 8, 8 * 4096 * 32768 * 96)
    r, pm, sc = rl.walk_roofline("k_fb_accumulate", 4096, avg_s, table, pmc, "profiles/r06_pmc.json")
    assert r["algorithmic_bytes_per_launch"] == 4096 * 131168 + 393216 == 537657344
    assert abs(r["achieved"] - 0.537657344 / avg_s) < 1e-9 and 0.0015 < r["frac"] < 0.0018           # 13 GB/s of 8 TB/s: the kernel is issue-bound
    assert pm is not None and sc == 1.0 and r["traffic"] == pm["fetch_bytes_per_launch"] + pm["write_bytes_per_launch"] > 50 * r["algorithmic_bytes_per_launch"]
    assert abs(row["avg_us"] * 1e-3 - line["roofline"]["avg_launch_ms"]) / line["roofline"]["avg_launch_ms"] < 0.03     # trace row and HIP events of the un-profiled run agree
    # a plain-layout table (16 windows) must NOT pick up the counters of the 8-window walk; neither may a batch that is no multiple of the profiled one
    assert rl.walk_roofline("k_fb_accumulate", 4096, avg_s, (16, 16, 2 * table[2]), pmc, "x")[1] is None
    assert rl.walk_roofline("k_fb_accumulate", 1000, avg_s, table, pmc, "x")[1] is None
    mac = rl.walk_mac(4096, 16, avg_s, line["roofline"]["mac"]["measured_peak_Tmad_s"] * 1e12, line["roofline"]["mac"]["measured_v_add_u32_Tops_s"] * 1e12, 6.8e10)
    assert mac["mads_per_launch"] == 4096 * 4096 * 16 * 3055 and 0.6 < mac["frac"] < 0.75
    iss = rl.issue_model(mac["mads_per_launch"], pm["valu_insts_per_launch"], avg_s, line["roofline"]["mac"]["measured_peak_Tmad_s"] * 1e12,
                         line["roofline"]["mac"]["measured_v_add_u32_Tops_s"] * 1e12)
    assert 0.5 < iss["mad_share_of_insts"] < 0.65 and 0.8 < iss["frac_of_launch_explained"] < 1.05
