"""Distributed helpers of the evaluation path with the reference's names (open_clip/utils.py:134-175, 295-330),
restated for RCCL over xGMI: every helper issues ONE collective on ONE contiguous buffer (ring collectives on xGMI
are latency-bound at these sizes, so fewer, fused calls win) instead of one call per tensor."""
import collections.abc
import itertools
from typing import List, Sequence

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def scaled_all_reduce(tensors: Sequence[torch.Tensor], is_scale: bool = True) -> Sequence[torch.Tensor]:
    """In-place sum over ranks of every tensor (scaled by 1/world_size unless is_scale=False): the tensors are packed
    into one flat fp32/own-dtype buffer, reduced with a single all-reduce and unpacked."""
    world = get_world_size()
    if world == 1 or len(tensors) == 0:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if is_scale:
        flat.mul_(1.0 / world)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors


@torch.no_grad()
def concat_all_gather(tensor: torch.Tensor) -> torch.Tensor:
    """Rank-major concatenation of equally shaped tensors (no gradient), one all_gather_into_tensor."""
    if not is_dist_avail_and_initialized():
        return tensor
    world = dist.get_world_size()
    t = tensor.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, t)
    return out


@torch.no_grad()
def all_gather(q: torch.Tensor, exclude_self: bool = False) -> torch.Tensor:
    """Gather tensors whose FIRST dimension differs between ranks: sizes are exchanged, the payload goes out padded to
    the longest in one collective, and the padding is cut off again (rank-major order)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = torch.zeros(world, device=q.device, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, torch.tensor([q.shape[0]], device=q.device, dtype=torch.int64))
    sizes = [int(s) for s in sizes.tolist()]
    longest = max(sizes)
    pad = torch.zeros((longest,) + tuple(q.shape[1:]), device=q.device, dtype=q.dtype)
    pad[:q.shape[0]] = q
    out = torch.empty((world * longest,) + tuple(q.shape[1:]), device=q.device, dtype=q.dtype)
    dist.all_gather_into_tensor(out, pad)
    parts: List[torch.Tensor] = [out[r * longest:r * longest + sizes[r]] for r in range(world) if not (exclude_self and r == rank)]
    return torch.cat(parts, dim=0) if parts else out[:0]


def get_model(model):
    """The module behind a DataParallel / DistributedDataParallel wrapper (utils.py:178-184)."""
    if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        return model.module
    return model


def _ntuple(n):
    def parse(x):
        return x if isinstance(x, collections.abc.Iterable) else tuple(itertools.repeat(x, n))
    return parse


to_1tuple, to_2tuple, to_3tuple, to_4tuple = _ntuple(1), _ntuple(2), _ntuple(3), _ntuple(4)
to_ntuple = lambda n, x: _ntuple(n)(x)
