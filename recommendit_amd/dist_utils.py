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
