"""The sharded FK20Multi orchestration (go-kzg_amd/multi_gpu.py): world_size 2 over gloo on CPU with the ORACLE standing
in for the two device calls (test infrastructure only), against the unsharded oracle result; plus the real HIP backend
at world size 1 on a GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_TEST = 1927409816240961209460912649124


class OracleBackend:
    """CPU stand-in with the same two-call interface: hExtFFT slice, then IFFT / pad / FFT / bit-reverse"""

    def __init__(self, n2, l):
        from oracle import koracle as ko
        self.ko, self.l, self.n2, self.k2 = ko, l, n2, n2 // l
        self.fs = ko.FFTSettings(n2.bit_length() - 1)
        self.ks = ko.KZGSettings(self.fs, ko.generate_testing_setup_g1(S_TEST, n2))
        self.fk = ko.FK20MultiSettings(self.ks, n2, l)

    def hext_slice(self, poly_t, n, j0, cnt):
        import torch
        ko = self.ko
        poly = poly_t.numpy().view(np.uint64).reshape(n, 4)
        acc = ko.g1_zero(self.k2)
        for f in range(self.l):
            tc = ko.toeplitz_coeffs_step_strided(poly, f, self.l)
            part = self.ks.toeplitz_part2(tc, self.fk.file(f))
            for j in range(j0, j0 + cnt):
                acc[j] = ko.g1_add(acc[j], part[j])
        return torch.from_numpy(acc[j0:j0 + cnt].reshape(cnt, 18).view(np.int64).copy())

    def finish(self, hext_t, bit_reverse=True):
        import torch
        ko = self.ko
        hext = hext_t.numpy().view(np.uint64).reshape(self.k2, 3, 6)
        h = self.fs.fft_g1(hext, inv=True)
        h[self.k2 // 2:] = ko.g1_zero(self.k2 // 2)
        out = self.fs.fft_g1(h)
        if bit_reverse:
            out = ko.reverse_bit_order(out)
        return torch.from_numpy(ko.g1_affine(out).reshape(self.k2, 18).view(np.int64).copy())




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

def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import gokzg_amd  # noqa: F401  (registers the package so the submodule import below resolves)
    from gokzg_amd import multi_gpu
    from oracle import koracle as ko
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    n2, l = 128, 4
    be = OracleBackend(n2, l)
    poly = ko.synthetic_blob(77, n2 // 2)
    got = multi_gpu.da_using_fk20_multi_sharded(be, torch.from_numpy(poly.view(np.int64).copy()), n2 // 2, be.k2)
    want = ko.g1_affine(be.fk.da_using_fk20_multi(poly))
    ok = np.array_equal(got.numpy().view(np.uint64).reshape(-1, 3, 6), want)
    mine = torch.full((3, 2, 18), rank, dtype=torch.int64)
    allp = multi_gpu.all_gather_proofs(mine)
    ok = ok and allp.shape == (6, 2, 18) and bool((allp[:3] == 0).all()) and bool((allp[3:] == 1).all())
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_fk20_multi_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs, 2, 300))
    _reap(procs, 60)
    assert res == [(0, True), (1, True)]


@pytest.mark.gpu
def test_sharded_fk20_multi_hip_backend_world1():
    """the real device calls (slice + finish) compose to the unsharded DAUsingFK20Multi; also in two slices"""
    import torch
    import gokzg_amd as kz
    from gokzg_amd import multi_gpu
    from oracle import koracle as ko
    n2, l = 1024, 16
    fs = kz.FFTSettings(10)
    setup = ko.generate_testing_setup_g1(S_TEST, n2)
    ks = kz.KZGSettings(fs, setup)
    fk = kz.FK20MultiSettings(ks, n2, l)
    poly = ko.synthetic_blob(9, n2 // 2)
    want = fk.da_using_fk20_multi(poly)
    be = multi_gpu.HipFK20MultiBackend(fk)
    d_poly = torch.from_numpy(poly.view(np.int64).copy()).cuda()
    got = multi_gpu.da_using_fk20_multi_sharded(be, d_poly, n2 // 2, be.k2)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64).reshape(-1, 3, 6), want)
    a = be.hext_slice(d_poly, n2 // 2, 0, 20)
    b = be.hext_slice(d_poly, n2 // 2, 20, be.k2 - 20)
    got2 = be.finish(torch.cat([a, b]))
    torch.cuda.synchronize()
    assert np.array_equal(got2.cpu().numpy().view(np.uint64).reshape(-1, 3, 6), want)
    fk.close(); ks.close(); fs.close()


def _gpu_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import gokzg_amd as kz
    from gokzg_amd import multi_gpu
    from oracle import koracle as ko
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KZG_HIP_FB_BUDGET_GB"] = "1"
    os.environ["KZG_HIP_FK20_FB_BUDGET_GB"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))   # both ranks share the one GPU: RCCL refuses that, gloo moves the bytes
    n2, l = 1024, 16
    fs = kz.FFTSettings(10)
    ks = kz.KZGSettings(fs, ko.generate_testing_setup_g1(S_TEST, n2))
    fk = kz.FK20MultiSettings(ks, n2, l)
    poly = ko.synthetic_blob(9, n2 // 2)
    want = fk.da_using_fk20_multi(poly)
    be = multi_gpu.HipFK20MultiBackend(fk)
    d_poly = torch.from_numpy(poly.view(np.int64).copy()).cuda()
    got = multi_gpu.da_using_fk20_multi_sharded(be, d_poly, n2 // 2, be.k2)
    torch.cuda.synchronize()
    ok = np.array_equal(got.cpu().numpy().view(np.uint64).reshape(-1, 3, 6), want)
    mine = torch.full((3, 2, 18), rank, dtype=torch.int64, device="cuda")
    allp = multi_gpu.all_gather_proofs(mine).cpu()
    ok = ok and allp.shape == (6, 2, 18) and bool((allp[:3] == 0).all()) and bool((allp[3:] == 1).all())
    q.put((rank, bool(ok)))
    fk.close(); ks.close(); fs.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_fk20_multi_hip_backend_two_ranks_one_gpu():
    """world size 2 with the REAL device calls on both ranks (each computes half of the Toeplitz stage on the GPU, the 144-byte
    slices cross a real all-gather, each finishes with the two G1 FFTs): bit-identical to the unsharded result on both ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs, 2, 600))
    _reap(procs, 120)
    assert res == [(0, True), (1, True)]


def _gpu_worker_scale16(rank, world, port, q):
    import hashlib
    import json
    import torch
    import torch.distributed as dist
    import gokzg_amd as kz
    from gokzg_amd import multi_gpu
    from oracle import koracle as ko
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KZG_HIP_FB_BUDGET_GB"] = "1"
    os.environ["KZG_HIP_FK20_FB_BUDGET_GB"] = "8"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    n2, l = 65536, 16
    fs = kz.FFTSettings(16)
    ks = kz.KZGSettings(fs, fs.generate_testing_setup_g1(ko.fr_from_ints([S_TEST]), n2))
    fk = kz.FK20MultiSettings(ks, n2, l)
    poly = ko.synthetic_blob(5, n2 // 2)
    be = multi_gpu.HipFK20MultiBackend(fk)
    d_poly = torch.from_numpy(poly.view(np.int64).copy()).cuda()
    got = multi_gpu.da_using_fk20_multi_sharded(be, d_poly, n2 // 2, be.k2)
    torch.cuda.synchronize()
    proofs = got.cpu().numpy().view(np.uint64).reshape(-1, 3, 6)
    pin = json.load(open(os.path.join(ROOT, "tests", "golden", "fk20_pins.json")))["config5_da_using_fk20_multi_seed5"]
    ok = proofs.shape[0] == pin["count"] and hashlib.sha256(fs.to_compressed_g1(proofs).tobytes()).hexdigest() == pin["sha256"]
    q.put((rank, bool(ok)))
    fk.close(); ks.close(); fs.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_fk20_multi_scale16_two_ranks_one_gpu_byte_pin():
    """BASELINE config 5 through the SHARDED driver at full size (n2 = 65536, chunk 16, seed 5): two ranks, real device calls, the
    slices cross a real all-gather; all 4096 compressed proofs hash to the oracle's pin (tests/golden/fk20_pins.json) on both ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker_scale16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs, 2, 900))
    _reap(procs, 120)
    assert res == [(0, True), (1, True)]
