"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on MI355X, "gloo" in the CPU tests).

The reference parallelises CalcInverseDynamicsPartialsFiniteDiff over timesteps with
OpenMP (reference optimizer/trajectory_optimizer.cc:476); across GPUs the same loop is
split into contiguous ranges of the tau-index k.  Each rank's fd_kernel writes the slab
records [dtau_k/dq_{k-1} | dtau_k/dq_k | dtau_k/dq_{k+1} | tau_k] of its range; one
all-gather makes the slab complete on every rank, which then assembles and solves
redundantly (identical bits on all ranks, no second collective).  DESIGN.md §7.
"""
from __future__ import annotations

import torch


def shard_len(N: int, world: int) -> int:
    """records per rank; the last ranks' ranges are padded / empty when world does not divide N"""
    return -(-N // world)


def shard_bounds(N: int, world: int, rank: int):
    """[k_begin, k_end) of `rank`: contiguous, equal-sized up to the tail, union = [0, N)"""
    per = shard_len(N, world)
    lo = min(N, rank * per)
    return lo, min(N, lo + per)


class SlabExchange:
    """All-gather of the per-k slab records.  `slab` is a 1-D fp64 tensor of N*stride elements
    (on the GPU: a zero-copy view of the context's resident slab) that holds this rank's
    records at their final position; after `gather()` it holds every rank's."""

    def __init__(self, dist, slab: torch.Tensor, N: int, stride: int, rank: int, world: int):
        assert slab.numel() == N * stride and slab.dtype == torch.float64
        self.dist, self.slab, self.N, self.stride, self.rank, self.world = dist, slab, N, stride, rank, world
        self.per = shard_len(N, world)
        self.lo, self.hi = shard_bounds(N, world, rank)
        self.even = (N % world == 0)
        if not self.even:  # padded staging buffers: equal-sized contributions for the collective
            self.send = torch.zeros(self.per * stride, dtype=slab.dtype, device=slab.device)
            self.recv = torch.zeros(world * self.per * stride, dtype=slab.dtype, device=slab.device)

    @property
    def mine(self) -> torch.Tensor:
        return self.slab[self.lo * self.stride:self.hi * self.stride]

    def gather(self):
        if self.even:  # in place: every rank's range is already at its final offset
            self.dist.all_gather_into_tensor(self.slab, self.mine)
            return
        n = (self.hi - self.lo) * self.stride
        self.send[:n].copy_(self.mine)
        self.dist.all_gather_into_tensor(self.recv, self.send)
        for r in range(self.world):
            lo, hi = shard_bounds(self.N, self.world, r)
            if r != self.rank and hi > lo:
                self.slab[lo * self.stride:hi * self.stride].copy_(
                    self.recv[r * self.per * self.stride:r * self.per * self.stride + (hi - lo) * self.stride])


def device_slab_view(dev, N: int) -> torch.Tensor:
    """zero-copy torch view of a HipPath's resident slab (include/idto_hip.h IDTO_ARR_SLAB)"""
    stride = dev.slab_stride

    class _Ptr:
        __cuda_array_interface__ = {"shape": (N * stride,), "typestr": "<f8",
                                    "data": (dev.device_ptr("slab"), False), "version": 2}
    return torch.as_tensor(_Ptr(), device=f"cuda:{dev.device}")


class RcclShard:
    """The same exchange inside libidto_hip.so (include/idto_hip.h idto_hip_comm_*): the library owns
    an RCCL communicator, `dist` (any torch.distributed backend, gloo is enough) only carries the
    128-byte unique id from rank 0 to the others.  After construction the HipPath's shard is its
    rank's k-range and `dev.gn_step_sharded()` is eval_partials + ncclAllGather (in place, on the
    context's stream) + grad_hess + factor_solve."""

    def __init__(self, dist, dev, rank: int, world: int):
        from . import hip
        box = [hip.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        dev.comm_init(box[0], rank, world)
        self.dev, self.rank, self.world = dev, rank, world
        self.lo, self.hi = shard_bounds(dev.N, world, rank)

    def close(self):
        self.dev.comm_destroy()
