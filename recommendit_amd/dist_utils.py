"""Collectives used by the multi-GPU training step (one process per GPU, torch.distributed).

On MI355X the backend is "nccl" (= RCCL over xGMI) and tensors stay on the device.  With the
"gloo" backend (CPU rehearsal / several ranks sharing one GPU in tests) device tensors are staged
through host memory so the very same call sequence runs."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def direct() -> bool:
    """RIHIP_COLLECTIVES=rccl (default) | direct.  The 8 GPUs of an MI355X node form a full xGMI mesh of point-to-point
    links (7 x ~153 GB/s per GPU): every peer is ONE hop away.  "direct" runs the all-gather of tower outputs and the
    reduce-scatter of partial item gradients as equal-split all-to-alls (7 concurrent one-hop transfers of 1/W of the
    data, then a local fixed-order sum) instead of RCCL's own schedules for ncclAllGather / ncclReduceScatter.  Its
    logic is covered by a CPU (gloo) test; it has never run on a multi-GPU node, which is why the library's own
    collectives stay the default until a scaling run has compared the two."""
    return os.environ.get("RIHIP_COLLECTIVES", "rccl") == "direct"


def _direct_for(t: torch.Tensor, group) -> bool:
    """one-hop forms: device tensors over RCCL when asked for; RIHIP_COLLECTIVES=direct_any also takes them for host
    tensors over gloo (the CPU test of their logic)"""
    mode = os.environ.get("RIHIP_COLLECTIVES", "rccl")
    if dist.get_world_size(group) < 2:
        return False
    if mode == "direct_any":
        return not _staged(t, group)
    return mode == "direct" and t.is_cuda and not _staged(t, group)


def _all_gather_direct(out: torch.Tensor, inp: torch.Tensor, group, async_op: bool):
    """the same block goes to every peer (input replicated W times: 10 us of HBM copies at 4 MiB)"""
    W = dist.get_world_size(group)
    rep = inp.contiguous().unsqueeze(0).expand(W, *inp.shape).contiguous().view(W * inp.shape[0], *inp.shape[1:])
    w = dist.all_to_all_single(out, rep, group=group, async_op=async_op)
    return w if async_op else _Done()


def _reduce_scatter_direct(out: torch.Tensor, inp: torch.Tensor, group, async_op: bool):
    """slice p of every rank's partial goes straight to rank p, which adds the W slices in a fixed order (bitwise
    reproducible, independent of any ring schedule)"""
    W = dist.get_world_size(group)
    recv = torch.empty_like(inp)
    work = dist.all_to_all_single(recv, inp.contiguous(), group=group, async_op=True)
    h = _DirectReduce(work, recv, out, W)
    if async_op:
        return h
    h.wait()
    return _Done()


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    def wait(self) -> None:
        return None


class _DirectReduce:
    """handle of the one-hop reduce-scatter: wait() orders the current stream after the exchange, then sums the W
    received slices into `out` (fixed order)"""

    def __init__(self, work, recv: torch.Tensor, out: torch.Tensor, world: int):
        self.work, self.recv, self.out, self.world = work, recv, out, world

    def wait(self) -> None:
        self.work.wait()
        # one launch; the association order inside is fixed by the kernel, not by arrival order: reproducible
        torch.sum(self.recv.view(self.world, *self.out.shape), dim=0, out=self.out)


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out[rank*n:(rank+1)*n] = inp of that rank (row-major concatenation over ranks).
    async_op=True (RCCL only): returns a work handle; the collective runs on RCCL's stream, ordered after the kernels
    already queued on the current stream, and `handle.wait()` orders later kernels after it -- lets an all-gather
    that is only needed two kernels later overlap with the kernel in between."""
    if _direct_for(inp, group):
        return _all_gather_direct(out, inp, group, async_op)
    if inp.is_cuda and not _staged(inp, group) and async_op:
        return dist.all_gather_into_tensor(out, inp.contiguous(), group=group, async_op=True)
    _all_gather_sync(out, inp, group)
    return _Done()


def _all_gather_sync(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    if _staged(inp, group):
        W = dist.get_world_size(group)
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(W)]
        dist.all_gather(parts, inp.detach().cpu().contiguous(), group=group)
        out.copy_(torch.cat(parts, 0).to(out.device))
    elif inp.is_cuda:
        dist.all_gather_into_tensor(out, inp.contiguous(), group=group)
    else:
        W = dist.get_world_size(group)
        parts = [torch.empty_like(inp) for _ in range(W)]
        dist.all_gather(parts, inp.contiguous(), group=group)
        out.copy_(torch.cat(parts, 0))


def all_reduce_sum_(t: torch.Tensor, group=None) -> None:
    if _staged(t, group):
        c = t.detach().cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c.to(t.device))
    else:
        dist.all_reduce(t, group=group)


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> None:
    if _staged(t, group):
        c = t.detach().cpu()
        dist.broadcast(c, src, group=group)
        t.copy_(c.to(t.device))
    else:
        dist.broadcast(t, src, group=group)


def reduce_scatter_sum(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out = (sum over ranks of inp)[rank*n:(rank+1)*n], n = out.shape[0] (rows).  Same async contract as
    all_gather_into."""
    if _direct_for(inp, group):
        return _reduce_scatter_direct(out, inp, group, async_op)
    if inp.is_cuda and not _staged(inp, group):
        w = dist.reduce_scatter_tensor(out, inp.contiguous(), group=group, async_op=async_op)
        return w if async_op else _Done()
    c = inp.detach().cpu().contiguous()          # gloo has no reduce_scatter: all-reduce on the host, keep own slice
    dist.all_reduce(c, group=group)
    r, n = dist.get_rank(group), out.shape[0]
    out.copy_(c[r * n:(r + 1) * n].to(out.device))
    return _Done()


def all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group=None, async_op: bool = False):
    """Variable all-to-all along dim 0: rank r sends inp rows [sum(in_splits[:p]), +in_splits[p]) to peer p and
    receives out_splits[p] rows from it (torch.distributed.all_to_all_single; RCCL: direct peer-to-peer sends over
    xGMI, one per link).  `out`/`inp` are exactly sum(out_splits)/sum(in_splits) rows long.  Same async contract as
    all_gather_into."""
    out_splits, in_splits = [int(x) for x in out_splits], [int(x) for x in in_splits]
    assert out.shape[0] == sum(out_splits) and inp.shape[0] == sum(in_splits)
    if _staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.detach().cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(o.to(out.device))
        return _Done()
    w = dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group, async_op=async_op)
    return w if async_op else _Done()


# ---- row-sharded tables: global row g >= 1 lives on rank (g-1) % W at local row (g-1) // W + 1; local row 0 is padding
def n_local_rows(n_global: int, rank: int, world: int) -> int:
    """number of global rows 1..n_global owned by `rank` (the local table has this + 1 rows)"""
    return max(0, (n_global - rank + world - 1) // world)


def shard_rows(full, rank: int, world: int):
    """[n_global+1, d] table (row 0 = padding) -> this rank's [n_local+1, d] shard (NumPy array or torch tensor)"""
    own = full[1 + rank::world]
    if isinstance(full, torch.Tensor):
        return torch.cat([full[:1], own], 0)
    import numpy as np
    return np.concatenate([full[:1], own], 0)
