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


def share_plan(plan, src: int = 0):
    """One plan for every rank: rank ``src`` hands its tuned plan (``IntegerEngine.export_plan()``: a small dict of strings) to all
    ranks, which then REPLAY it (``IntegerEngine(model, plan=...)``) instead of tuning their own.  Independently tuned plans differ
    in a few launches and by +-2 % in speed, and under the MAX-over-ranks clock the slowest of N plans would set the number; it
    also saves N - 1 tuning passes.  The analogue of the reference's single replicated module (quant_train.py:350-358: one
    ``model`` object wrapped by DataParallel).  Returns the shared plan; without a process group, the argument."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return plan
    box = [plan if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def plans_identical(plan) -> bool:
    """True iff every rank holds the same plan (compared through a digest of its canonical JSON)."""
    import hashlib
    import json
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    digest = hashlib.sha256(json.dumps(plan, sort_keys=True).encode()).hexdigest()
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, digest)
    return len(set(seen)) == 1


def gather_logits(local_logits: torch.Tensor, batch: int | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """all_gather of per-rank logits -> [batch, classes] in rank order.  Equal shards use one
    all_gather_into_tensor (into ``out`` if given: [world * n, classes]); ragged shards are padded to the largest
    shard and trimmed (``out``, if given, must then be [batch, classes] and receives the trimmed result).  Without a
    process group the local logits ARE the batch; an initialised group of ONE rank still runs the collective (bench.py
    and the GPU test exercise the RCCL path that way on a 1-GPU box)."""
    if not dist.is_initialized():
        return local_logits
    world = dist.get_world_size()
    n, c = local_logits.shape
    if batch is None or batch % world == 0:
        if out is None:
            out = torch.empty(world * n, c, dtype=local_logits.dtype, device=local_logits.device)
        elif tuple(out.shape) != (world * n, c) or out.dtype != local_logits.dtype or out.device != local_logits.device:
            raise ValueError(f"gather_logits: out must be {(world * n, c)} {local_logits.dtype} on {local_logits.device}, "
                             f"got {tuple(out.shape)} {out.dtype} on {out.device}")
        dist.all_gather_into_tensor(out, local_logits.contiguous())
        return out
    if out is not None and (tuple(out.shape) != (batch, c) or out.dtype != local_logits.dtype or out.device != local_logits.device):
        raise ValueError(f"gather_logits: ragged shards need out of shape {(batch, c)} {local_logits.dtype} on {local_logits.device}, "
                         f"got {tuple(out.shape)} {out.dtype} on {out.device}")
    nmax = (batch + world - 1) // world
    padded = torch.zeros(nmax, c, dtype=local_logits.dtype, device=local_logits.device)
    padded[:n] = local_logits
    full = torch.empty(world * nmax, c, dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(full, padded)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(batch, r, world)
        parts.append(full[r * nmax:r * nmax + (hi - lo)])
    if out is not None:
        torch.cat(parts, 0, out=out)
        return out
    return torch.cat(parts, 0)


def sharded_forward(forward, images: torch.Tensor) -> torch.Tensor:
    """Evaluate ``forward`` on this rank's shard of ``images`` and gather the full-batch logits."""
    if not dist.is_initialized():
        return forward(images)
    lo, hi = shard_bounds(images.shape[0], dist.get_rank(), dist.get_world_size())
    return gather_logits(forward(images[lo:hi]), images.shape[0])
