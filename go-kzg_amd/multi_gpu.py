"""One-process-per-GPU drivers (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Two ways the hot path uses several GPUs:

* blobs are independent: `shard_units` gives each rank a contiguous share of a batch, every rank runs the whole
  pipeline on its share, no data-path collective (this is what bench.py measures).
* ONE scale-16 FK20Multi split across ranks (`da_using_fk20_multi_sharded`): the Toeplitz stage
  hExtFFT[j] = sum_f C_f[j] X_f[j] is sharded by output position j; each rank computes its slice with
  kzg_hip_fk20_multi_hext_slice_dev, the slices (144-byte device-internal points, plain bytes) are all-gathered,
  and every rank finishes with the two G1 FFTs (kzg_hip_fk20_multi_finish_dev).  G1 points have no reduce op in
  RCCL, so the exchange is an all-gather of bytes, 4096 x 144 B = 576 KiB in total: latency-bound on xGMI.

`backend` abstracts the two device calls so that the orchestration can be tested with gloo on CPU ranks
(tests/test_multi_gpu.py); the HIP backend is the only product backend.
"""
import torch
import torch.distributed as dist


def shard_units(total_units, world, rank):
    base, rem = divmod(total_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipFK20MultiBackend:
    """the two C-ABI calls on device-resident torch tensors (int64 view of the 144-byte point images)"""

    def __init__(self, fk):
        import gokzg_amd as kz
        self.fk, self.lib = fk, kz.lib()
        self.k2 = fk.n2 // fk.chunk_len

    def hext_slice(self, d_poly, n, j0, cnt):
        out = torch.empty((cnt, 18), dtype=torch.int64, device=d_poly.device)
        st = self.lib.kzg_hip_fk20_multi_hext_slice_dev(self.fk.h, d_poly.data_ptr(), n, j0, cnt, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if st:
            raise RuntimeError("kzg_hip_fk20_multi_hext_slice_dev status %d" % st)
        return out

    def finish(self, d_hext, bit_reverse=True):
        out = torch.empty((self.k2, 18), dtype=torch.int64, device=d_hext.device)
        st = self.lib.kzg_hip_fk20_multi_finish_dev(self.fk.h, d_hext.data_ptr(), int(bit_reverse), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if st:
            raise RuntimeError("kzg_hip_fk20_multi_finish_dev status %d" % st)
        return out


def da_using_fk20_multi_sharded(backend, d_poly, n, k2, group=None):
    """DAUsingFK20Multi (fk20_multi.go:113-133) of ONE polynomial with the Toeplitz stage sharded over the ranks of `group`.
    Returns the 2k proofs (reverse-bit order) on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_units(k2, world, rank)
    mine = backend.hext_slice(d_poly, n, lo, hi - lo)
    if world == 1:
        hext = mine
    else:
        spans = [shard_units(k2, world, r) for r in range(world)]
        if all(h - l == spans[0][1] - spans[0][0] for l, h in spans):
            hext = torch.empty((k2,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(hext, mine.contiguous(), group=group)      # one fused all-gather of bytes
        else:
            bufs = [torch.empty((h - l,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device) for l, h in spans]
            dist.all_gather(bufs, mine.contiguous(), group=group)
            hext = torch.cat(bufs)
    return backend.finish(hext, bit_reverse=True)


def all_gather_proofs(d_proofs, group=None):
    """Data-parallel FK20: every rank holds the proofs of ITS blobs (batch_r x m x 18 int64 = 144-byte points); one RCCL
    all-gather of bytes over xGMI gives every rank the proofs of all blobs, rank-major.  (The "all-gather of proof points"
    of the north star; 48-byte compressed form works the same way on uint8 tensors.)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return d_proofs
    world = dist.get_world_size(group)
    out = torch.empty((world * d_proofs.shape[0],) + tuple(d_proofs.shape[1:]), dtype=d_proofs.dtype, device=d_proofs.device)
    dist.all_gather_into_tensor(out, d_proofs.contiguous(), group=group)
    return out
