"""world_size-2 CPU (gloo) test of the batch-sharding + logits all_gather logic used for N > 1.
The per-rank 'forward' here is a deterministic stand-in (the real one needs a GPU); what is tested is
that sharding + gather reproduces the single-process result bit-for-bit, incl. ragged batches."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_forward(x):  # per-image independent, like the frozen network
    g = torch.Generator().manual_seed(0)
    w = torch.randn(3 * 8 * 8, 10, generator=g)
    return x.reshape(x.shape[0], -1) @ w


def _worker(rank, world, port, batch, q):
    import torch.distributed as dist
    from hawq_amd.dist import gather_logits, shard_bounds, sharded_forward
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.randn(batch, 3, 8, 8, generator=torch.Generator().manual_seed(1))
    full = sharded_forward(_fake_forward, x)
    lo, hi = shard_bounds(batch, rank, world)
    again = gather_logits(_fake_forward(x[lo:hi]), batch)
    ok = torch.equal(full, _fake_forward(x)) and torch.equal(full, again)
    q.put((rank, bool(ok), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])
def test_sharded_forward_matches_single_process(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and all(shape == (batch, 10) for _, _, shape in res)


def test_shard_bounds_cover_batch():
    from hawq_amd.dist import shard_bounds
    for batch in (1, 7, 128, 130):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
