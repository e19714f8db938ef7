"""world_size-2 CPU (gloo) test of the batch-sharding + logits all_gather logic used for N > 1.
The per-rank 'forward' here is a deterministic stand-in (the real one needs a GPU); what is tested is
that sharding + gather reproduces the single-process result bit-for-bit, incl. ragged batches."""
import os
import socket

import pytest
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def test_bench_strong_scaling_shards_line_up_with_the_golden_logits():
    """bench.py --scaling strong: rank r evaluates images shard_bounds(128, r, N) of ONE batch and checks its logits
    against the same rows of the oracle fixture; the gathered result is the fixture itself."""
    import numpy as np
    import bench
    from hawq_amd.dist import shard_bounds
    from tests import helpers as H
    fx = H.load("b128_resnet50_uniform8.npz")
    full = torch.from_numpy(fx["logits"])
    for world in (1, 2, 4, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(128, r, world)
            assert hi - lo == 128 // world
            assert bench.golden_parity("resnet50", "uniform8", 128, 1, full[lo:hi], lo) is True
            parts.append(full[lo:hi])
        assert torch.equal(torch.cat(parts), full)
    wrong = full.clone()
    wrong[5, 7] += 1
    assert bench.golden_parity("resnet50", "uniform8", 128, 1, wrong) is False
    assert bench.golden_parity("resnet50", "uniform8", 64, 1, full[:64]) is None   # no fixture for other batches


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (it used to run ONE rank
    silently).  --dry-spawn keeps the whole launch / rendezvous / shard / gather / MAX-over-ranks / one-JSON-line path and
    replaces the network by a CPU stand-in on gloo."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-spawn", "--steps", "3", "--warmup", "1", "--batch", "8"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["weak"]["global_batch"] == 16 and d["weak"]["batch_per_gpu"] == 8 and d["weak"]["every_rank_parity"] is True
    assert d["strong"]["global_batch"] == 8 and d["strong"]["batch_per_gpu"] == 4 and d["strong"]["every_rank_parity"] is True
    assert d["value"] == d["weak"]["value"]
    # rank 0's plan is broadcast and every rank holds it (hawq_amd.dist.share_plan / plans_identical): one plan for the whole job
    assert d["config"]["plan_identical_on_all_ranks"] is True and d["config"]["plan_is_rank0s"] is True
    assert d["config"]["plan"]["tiles"] == "3.4.5.6.7" and d["config"]["plan"]["fused_variants"] == "1.0"
    # a launcher that provides another world size than --gpus asks for is refused, not silently accepted
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-spawn"], cwd=root, env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def _plan_worker(rank, world, port, q):
    import torch.distributed as dist
    from hawq_amd.dist import gather_logits, plans_identical, share_plan
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    own = {"batch": 128, "chains": 2, "tiles": f"{rank}.{rank + 1}", "fused_variants": "7.0"}
    before = plans_identical(own)
    shared = share_plan(own if rank == 0 else None)
    # the ragged gather honours a preallocated output: 7 images over 2 ranks
    lo, hi = (0, 4) if rank == 0 else (4, 7)
    full = torch.arange(70, dtype=torch.float32).reshape(7, 10)
    out = torch.full((7, 10), -1.0)
    res = gather_logits(full[lo:hi], 7, out=out)
    try:
        gather_logits(full[lo:hi], 7, out=torch.empty(8, 10))
        refused = False
    except ValueError:
        refused = True
    q.put((rank, before, plans_identical(shared), shared, bool(res.data_ptr() == out.data_ptr() and torch.equal(out, full)), refused))
    dist.barrier()
    dist.destroy_process_group()


def test_share_plan_gives_every_rank_rank0s_plan_and_ragged_gather_honours_out():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, before, after, shared, out_ok, refused in res:
        assert before is False and after is True and shared["tiles"] == "0.1" and out_ok and refused


def _world1_worker(port, q):
    import torch.distributed as dist
    from hawq_amd.dist import gather_logits
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    x = torch.randn(5, 7)
    y = gather_logits(x)
    q.put(bool(y.data_ptr() != x.data_ptr() and torch.equal(x, y)))
    dist.destroy_process_group()


def test_gather_logits_runs_the_collective_in_a_world_of_one():
    """an initialised group of one rank is NOT short-circuited: the product function itself issues the all_gather"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


@pytest.mark.gpu
def test_rccl_gather_executes_on_the_gpu_at_world_size_one():
    """The collective of the multi-GPU path (one all_gather of the logits over RCCL) executed for real, in a world of
    one rank, through the product's own hawq_amd.dist.gather_logits (bench.rccl_world1_selfcheck)."""
    import bench
    x = torch.randn(128, 1000, device="cuda")
    assert bench.rccl_world1_selfcheck(torch.device("cuda", 0), x) is True
