"""Batch-sharded data parallelism for the frozen forward.

One process per GPU; weights replicated at load; rank r evaluates images [r*B/W, (r+1)*B/W) (or its
own full batch in the weak-scaling bench); the only collective is one all_gather of the fp32 logits
- the analogue of the reference's DataParallel gather (quant_train.py:358).  Backend "nccl" is
RCCL over xGMI on MI355X nodes; "gloo" is used by the CPU tests of the sharding logic."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(batch: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a batch; the first (batch % world) ranks take one extra image."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_logits(local_logits: torch.Tensor, batch: int | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """all_gather of per-rank logits -> [batch, classes] in rank order.  Equal shards use one
    all_gather_into_tensor (into ``out`` if given: [world * n, classes]); ragged shards are padded to the largest
    shard and trimmed.  Without a process group the local logits ARE the batch; an initialised group of ONE rank
    still runs the collective (bench.py and the GPU test exercise the RCCL path that way on a 1-GPU box)."""
    if not dist.is_initialized():
        return local_logits
    world = dist.get_world_size()
    n, c = local_logits.shape
    if batch is None or batch % world == 0:
        if out is None:
            out = torch.empty(world * n, c, dtype=local_logits.dtype, device=local_logits.device)
        dist.all_gather_into_tensor(out, local_logits.contiguous())
        return out
    nmax = (batch + world - 1) // world
    padded = torch.zeros(nmax, c, dtype=local_logits.dtype, device=local_logits.device)
    padded[:n] = local_logits
    full = torch.empty(world * nmax, c, dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(full, padded)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(batch, r, world)
        parts.append(full[r * nmax:r * nmax + (hi - lo)])
    return torch.cat(parts, 0)


def sharded_forward(forward, images: torch.Tensor) -> torch.Tensor:
    """Evaluate ``forward`` on this rank's shard of ``images`` and gather the full-batch logits."""
    if not dist.is_initialized():
        return forward(images)
    lo, hi = shard_bounds(images.shape[0], dist.get_rank(), dist.get_world_size())
    return gather_logits(forward(images[lo:hi]), images.shape[0])
