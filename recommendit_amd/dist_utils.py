"""Collectives used by the multi-GPU training step (one process per GPU, torch.distributed).

On MI355X the backend is "nccl" (= RCCL over xGMI) and tensors stay on the device.  With the
"gloo" backend (CPU rehearsal / several ranks sharing one GPU in tests) device tensors are staged
through host memory so the very same call sequence runs."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    def wait(self) -> None:
        return None


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out[rank*n:(rank+1)*n] = inp of that rank (row-major concatenation over ranks).
    async_op=True (RCCL only): returns a work handle; the collective runs on RCCL's stream, ordered after the kernels
    already queued on the current stream, and `handle.wait()` orders later kernels after it -- lets an all-gather
    that is only needed two kernels later overlap with the kernel in between."""
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


def exchange_counts(send_counts: torch.Tensor, group=None):
    """send_counts int64[W] (device or host): rows this rank will send to each peer.  Returns two Python lists
    (send, recv) -- the ONE host synchronisation of a sharded-table step (the split sizes of the all-to-alls)."""
    W = dist.get_world_size(group)
    if _staged(send_counts, group) or not send_counts.is_cuda:
        s = send_counts.detach().cpu().contiguous()
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=group)
        return s.tolist(), r.tolist()
    both = torch.empty((2 * W,), dtype=send_counts.dtype, device=send_counts.device)
    both[:W].copy_(send_counts)
    dist.all_to_all_single(both[W:], both[:W].clone(), group=group)
    h = both.cpu()
    return h[:W].tolist(), h[W:].tolist()


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
