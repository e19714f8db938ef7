#!/usr/bin/env python
"""Throughput bench of the MI355X integer forward (BASELINE.json metric: images/s, ResNet50
W8A8 at batch 128 per GPU; W4A4 / mixed / ResNet18 reported in ``extra``).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--arch resnet50] [--scheme uniform8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frozen forward of one batch: fp32 images resident in HBM -> fp32 logits in HBM
(one hipGraph launch of the fused integer plan).  With N > 1 the path shards by batch, no data-path
collective, and each step ends with one RCCL all_gather of the logits (the reference's DataParallel
gather, quant_train.py:358) through ``hawq_amd.dist.gather_logits``.  ``--gpus N`` without a launcher
(no WORLD_SIZE in the environment) starts the N ranks itself (``torch.distributed.run`` on 127.0.0.1).
A multi-rank run measures BOTH decompositions and reports ``--scaling`` (default weak) as ``value``:
  weak    every rank processes its own batch of 128; value = N * 128 * K / t
  strong  ONE batch of 128 is sharded 128/N images per rank; value = 128 * K / t
``n_gpus`` is the number of ranks the process group actually has.  Prints ONE JSON line on rank 0.
``--dry-spawn`` runs the same launch / shard / gather / report path on CPU (gloo) with a stand-in forward:
what the CPU test of the N > 1 plumbing uses (there is no CPU path for the network itself).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COPY_CEILING_GBS = 6290.0   # measured float4 copy rate of the part (MI355X_MICROARCH.md): what a plan's byte floor is priced at


class Plans:
    """Recorded plans by workload key (``<arch>_<scheme>_b<images the engine sees>``): ``profiles/plans.json`` by default, so that
    the default bench REPLAYS the committed tiles / fused variants / chain count instead of re-tuning (two tuning passes differ
    by +-2 %, more than most kernel changes - VERDICT r3 #5 / #11) and the committed PMC traffic belongs to the plan that is timed.
    ``--retune`` ignores the file; ``--save-plan`` writes the plans this run used or tuned."""

    def __init__(self, path, retune):
        self.path, self.loaded, self.used = path, {}, {}
        if path and not retune and os.path.isfile(path):
            with open(path) as f:
                self.loaded = json.load(f)

    def get(self, key):
        pl = self.loaded.get(key)
        label = os.environ.get("HAWQ_PLAN_LABEL") or os.path.relpath(self.path, ROOT)   # (the extras child reads a temp copy of the parent's plans)
        return dict(pl, source=f"replayed {label} (tuned at git head {pl.get('git_head', '?')})") if pl else None

    def record(self, key, eng):
        pl = eng.export_plan()
        pl["git_head"] = self.loaded.get(key, {}).get("git_head") if eng.plan_source.startswith("replayed") else os.environ.get("GRAFT_HEAD", "unknown")
        self.used[key] = pl

    def save(self, path):
        merged = dict(self.loaded)
        merged.update(self.used)
        with open(path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)


def setup_workload(arch, scheme, batch, dev, seed, shard=None, plans=None, model=None, share=False):
    """Synthetic weights (seed 0), ranges calibrated on 8 synthetic images, `batch` synthetic images (seed);
    ``shard = (lo, hi)`` keeps only that slice of the batch (strong scaling: this rank's images).  ``plans``: replay the recorded plan
    of this workload if there is one.  ``share`` (N > 1): rank 0's plan - recorded or tuned just now - is broadcast and every other
    rank replays it (hawq_amd.dist.share_plan), so that all ranks run ONE plan and only one rank pays for tuning."""
    import torch.distributed as dist
    from hawq_amd.api import build_quantized_resnet, calibrate
    from hawq_amd.dist import share_plan
    from hawq_amd.engine import IntegerEngine
    from hawq_amd.skeleton import synthetic_images

    if model is None:
        model = build_quantized_resnet(arch, scheme, seed=0).to(dev)
        calibrate(model, synthetic_images(8, seed=0).to(dev))
    x = synthetic_images(batch, seed=seed)
    if shard is not None:
        x = x[shard[0]:shard[1]]
    x = x.to(dev)
    key = f"{arch}_{scheme}_b{x.shape[0]}"
    plan = plans.get(key) if plans is not None else None
    leader = not share or not dist.is_initialized() or dist.get_rank() == 0
    eng = None
    if leader:
        eng = IntegerEngine(model, use_graph=True, plan=plan)
        eng(x)  # allocate, replay the plan or autotune, warm up, capture the hipGraph
        plan = dict(eng.export_plan(), source=eng.plan_source if eng.plan_source.startswith("replayed") else "tuned on rank 0 in this run")
    if share and dist.is_initialized():
        plan = share_plan(plan if leader else None)
        if not leader:
            eng = IntegerEngine(model, use_graph=True, plan=plan)
            eng(x)
    if plans is not None:
        plans.record(key, eng)
    spin_up(eng)
    return model, eng, x


def spin_up(eng, seconds=0.7):
    """Part of the set-up, outside every timed region: replay the captured forward for `seconds` so that the part is at its sustained
    clocks when the W warm-up steps begin.  A tuning pass used to do that as a side effect (seconds of launches); an engine built by
    REPLAYING a recorded plan reaches the timed region after a few milliseconds of GPU work, and the first ~100 ms after idle run at ramping
    clocks (measured: 87.7 k img/s for a replayed plan straight after the build vs 93-94 k for the same plan after a tuned run's launches;
    the driver's --warmup 5 is 7 ms).  Applied to tuned and replayed engines alike."""
    # HAWQ_BENCH_SPIN_UP=<seconds> (tools/profile_round.sh sets 0 for its counter passes: the totals of two runs are subtracted, so
    # both must launch the same number of forwards, which a time-based loop under a profiler does not)
    seconds = float(os.environ.get("HAWQ_BENCH_SPIN_UP", seconds))
    t0 = time.perf_counter()
    with torch.cuda.stream(eng.stream):
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                eng.run_resident()
            torch.cuda.synchronize()


def mobilenet_line(batch, dev, steps):
    import numpy as np
    from hawq_amd import roofline
    from hawq_amd.api import build_quantized_model, calibrate
    from hawq_amd.skeleton import synthetic_images

    model = build_quantized_model("mobilenetv2_w1", "uniform8", seed=0).to(dev)
    calibrate(model, synthetic_images(8, seed=0).to(dev))
    x = synthetic_images(batch, seed=1).to(dev)
    eng = model.engine()
    y = eng(x)
    with torch.no_grad():
        y_mod = model.forward_modules(x[:8])
    s = (model.output.conv_scaling_factor.reshape(1, -1).double() * model.quant_act_output.act_scaling_factor.double()).cpu().numpy()
    same = bool(np.array_equal(np.rint(y[:8].cpu().numpy() / s), np.rint(y_mod.cpu().numpy() / s)))
    wall, gpu_ms, blk = timed_steps(eng, steps, 5, 1)
    return {"images_per_s": round(batch * steps / wall, 1), "gpu_ms": round(gpu_ms, 4), "gpu_ms_std": blk["std_ms"],
            "launches": eng.n_launches, "concurrent_sub_batches": eng.chains, "chain_timing_ms": {str(k): round(v, 4) for k, v in getattr(eng, "chain_timing_ms", {}).items()} or None,
            "fast_requant_launches": eng.fast_requant_launches, "autotuned_tiles": eng.tile_choice,
            "plan_bytes_per_image": int(eng.total_plan_bytes // batch),
            "hbm_frac": round(eng.total_plan_bytes / (gpu_ms * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS, 4),
            # all 128 x 1000 logits against the CPU oracle's fixture (oracle/oracle_mbv2.py through tests/golden/make_b128.py)
            "gpu_logits_bit_equal": golden_parity("mobilenetv2_w1", "uniform8", batch, 1, y),
            "plan_equals_module_path": same}


def golden_parity(arch, scheme, batch, seed, logits, lo=0):
    """Compare logits of the benchmarked workload with the CPU oracle's (tests/golden/b128_*.npz, all 128 images;
    written by tests/golden/make_b128.py).  None if there is no fixture for this workload."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", f"b128_{arch}_{scheme}.npz")
    if batch != 128 or not os.path.isfile(path):
        return None
    fx = np.load(path)
    if int(fx["seed"]) != seed:
        return None
    y = logits.cpu().numpy()
    return bool(np.array_equal(y, fx["logits"][lo:lo + y.shape[0]]))


def timed_steps(eng, steps, warmup, world, batch_total=None):
    """W untimed + K timed steps on the engine stream; returns (wall seconds, GPU ms per step, block stats).
    With an initialised process group every step ends with the logits gather of the multi-GPU path
    (hawq_amd.dist.gather_logits into a preallocated tensor); the wall time is then the MAX over ranks.  Block stats: the
    K steps are cut into >= 10 blocks (HIP events recorded between steps, no synchronisation) so that a run reports mean
    +- std of the per-step time (tvm_benchmark/test_resnet_inference_time.py:257-271 protocol)."""
    import torch.distributed as dist
    from hawq_amd import _lib
    from hawq_amd.dist import gather_logits

    sp = eng.stream.cuda_stream
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    _lib.call("hawq_event_create", C.byref(ev0))
    _lib.call("hawq_event_create", C.byref(ev1))
    blk = max(1, steps // 10)
    marks = []
    gathered = torch.empty(world * eng.logits.shape[0], eng.logits.shape[1], device=eng.logits.device) if world > 1 else None

    def step():
        eng.run_resident()
        if world > 1:
            gather_logits(eng.logits, batch_total, out=gathered)

    with torch.cuda.stream(eng.stream):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.call("hawq_event_record", ev0, sp)
        for i in range(steps):
            step()
            if (i + 1) % blk == 0 and i + 1 < steps:
                e = C.c_void_p()
                _lib.call("hawq_event_create", C.byref(e))
                _lib.call("hawq_event_record", e, sp)
                marks.append((i + 1, e))
        _lib.call("hawq_event_record", ev1, sp)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    ms = C.c_float()
    _lib.call("hawq_event_elapsed_ms", ev0, ev1, C.byref(ms))
    total = ms.value
    per_block, prev_e, prev_i = [], ev0, 0
    for i, e in marks + [(steps, ev1)]:
        _lib.call("hawq_event_elapsed_ms", prev_e, e, C.byref(ms))
        per_block.append(ms.value / (i - prev_i))
        prev_e, prev_i = e, i
    for _, e in marks:
        _lib.call("hawq_event_destroy", e)
    _lib.call("hawq_event_destroy", ev0)
    _lib.call("hawq_event_destroy", ev1)
    n = len(per_block)
    mean = sum(per_block) / n
    std = (sum((v - mean) ** 2 for v in per_block) / max(n - 1, 1)) ** 0.5
    stats = dict(blocks=n, steps_per_block=blk, mean_ms=round(mean, 4), std_ms=round(std, 4),
                 min_ms=round(min(per_block), 4), max_ms=round(max(per_block), 4))
    wall = t1 - t0
    if world > 1:   # the contract's clock: MAX over ranks
        t = torch.tensor([wall], device=eng.logits.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, total / steps, stats


def cpu_baseline(model, x, gpu_logits, sample=16, runs=3):
    """The CPU fake-quant port (oracle/fakequant_port.py: the reference's frozen forward with its cost structure -
    per-forward BN fold + weight re-quantisation + Decimal batch_frexp) on the host cores, SURVEY 8(d) protocol scaled to a
    bounded sample: a quick thread-count sweep (8 / 32 / all cores on 4 images), one warm-up, then `runs` timed forwards of
    the first `sample` images of the SAME batch; its logits double as a parity check of the GPU result.  Secondary baseline
    (SURVEY 8(d)): the same port with its tensors on the MI355X (`.cuda()`, fp32 MIOpen / rocBLAS path)."""
    from oracle import fakequant_port, oracle

    st = oracle.extract_float_state(model)
    xs = x[:sample].cpu()
    all_threads = torch.get_num_threads()
    sweep = {}
    for nt in sorted({min(8, all_threads), min(32, all_threads), all_threads}):
        torch.set_num_threads(nt)
        fakequant_port.forward(st, xs[:2])
        t0 = time.perf_counter()
        fakequant_port.forward(st, xs[:4])
        sweep[nt] = 4 / (time.perf_counter() - t0)
    best_nt = max(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    fakequant_port.forward(st, xs[:2])
    secs, y = [], None
    for _ in range(runs):
        t0 = time.perf_counter()
        y = fakequant_port.forward(st, xs)
        secs.append(time.perf_counter() - t0)
    torch.set_num_threads(all_threads)
    n = xs.shape[0]
    med = sorted(secs)[len(secs) // 2]
    out = dict(value=round(n / med, 3), unit="images/s", cores=best_nt, kind="port",
               sample=f"{runs} timed forwards (median) of the first {n} images of the benchmarked batch after a thread-count sweep "
                      f"{ {k: round(v, 2) for k, v in sweep.items()} } img/s on 4 images and a 2-image warm-up (torch-CPU fp32 fake-quant port of "
                      f"the reference path; seconds per forward: {[round(r, 2) for r in secs]})",
               gpu_logits_bit_equal=bool(torch.equal(y, gpu_logits[:n].cpu())), images_compared=n)
    try:   # the port with its tensors on the GPU: what `model.cuda()` of the reference's own path costs on this part
        dev = gpu_logits.device
        xg = x[:32]
        fakequant_port.forward(st, xg[:2], dev)
        torch.cuda.synchronize()
        gs = []
        for _ in range(3):
            t0 = time.perf_counter()
            yg = fakequant_port.forward(st, xg, dev)
            torch.cuda.synchronize()
            gs.append(time.perf_counter() - t0)
        out["port_on_gpu"] = dict(value=round(xg.shape[0] / sorted(gs)[1], 2), unit="images/s", images=int(xg.shape[0]),
                                  seconds=[round(g, 3) for g in gs], logits_equal_integer_path=bool(torch.equal(yg, gpu_logits[:xg.shape[0]])),
                                  note="fp32 fake-quant port, tensors on the MI355X (torch / MIOpen fp32 convs, host Decimal frexp per layer)")
    except Exception as exc:  # a secondary figure must never take the bench line down
        out["port_on_gpu"] = f"failed: {type(exc).__name__}: {exc}"
    return out


def plan_rows(arch, scheme):
    """fused-plan byte model rows by layer name (hawq_amd/roofline.py:fused_plan_table)"""
    from hawq_amd import roofline
    return {r["name"].split("+")[0]: r for r in roofline.fused_plan_table(arch, scheme)}


def launch_model(name, rows, batch):
    """(bytes the launch has to move, MACs) for `batch` images from the FUSED plan's byte model: a fused expand -> reduce
    launch covers two rows and never moves the 8-bit tensor between them (the first row's write of it, the second row's read)."""
    parts = [p for p in name.split("+") if p != "identity"]
    if parts[0] not in rows:
        return 0, 0
    r = rows[parts[0]]
    byt = (r["read"] + r["write"]) * batch + r["weight_bytes"]
    mac = r["macs"] * batch
    for extra in parts[1:]:
        if extra in rows:
            e = rows[extra]
            byt += (e["write"] - e["read"]) * batch + e["weight_bytes"]
            mac += e["macs"] * batch
    return byt, mac


def dominant_kernel(eng, args, rows):
    """The rocprof kernel that takes the largest share of the forward, from the COMMITTED profile passes of this workload
    (tools/profile_round.sh -> tools/dominant_kernel.py -> profiles/dominant_kernel.json: rocprof name, launches per forward, average
    duration, matrix-pipe busy fraction), plus - computed here from THIS run's plan - the bytes the fused plan moves for the
    launches that run that kernel, against 8 TB/s over the profiled duration.  None when no profile of the workload is committed."""
    from hawq_amd import _lib, roofline
    try:
        with open(os.path.join(ROOT, "profiles", "dominant_kernel.json")) as fh:
            rec = json.load(fh).get(f"{args.arch}_{args.scheme}_b{args.batch}")
    except (OSError, ValueError):
        rec = None
    if not rec:
        return None
    L = _lib.load()
    n, nb, nb2, ng2 = (L.hawq_conv2d_num_tiles(), L.hawq_conv2d_num_band_tiles(), L.hawq_conv2d_num_band2_tiles(), L.hawq_conv2d_num_gemm2_tiles())
    first_special = n - nb + 1                       # 1-based tile ids: generic | band (rounds 1-4) | persistent band | band v2 | streaming 1x1
    fam = {"conv3x3_v2_kernel": range(n - ng2 - nb2 + 1, n - ng2 + 1), "gemm1x1_v2_kernel": range(n - ng2 + 1, n + 1),
           "band_persist_kernel": range(n - ng2 - nb2 - 1, n - ng2 - nb2 + 1), "conv3x3_band_kernel": range(first_special, n - ng2 - nb2 - 1),
           "conv_kernel": range(1, first_special)}
    ids = next((r for k, r in fam.items() if k in rec["rocprof_name"]), None)
    fused = {p.split("+")[0] for p in getattr(eng, "er_choice", {}) if eng.er_choice[p]}
    chains = max(1, getattr(eng, "chains", 1))
    names = None
    if ids is not None:
        names = [k for k, t in eng.tile_choice.items() if t in ids and k.split("+")[0] not in fused]
    elif "expand_" in rec["rocprof_name"]:
        # a fused expand (-> reduce) kernel: the pairs that share the variant id which as many pairs run as the trace saw launches per chain
        import re as _re
        per_chain = round(rec["launches_per_forward"] / chains)
        by_variant = {}
        for k, v in getattr(eng, "er_choice", {}).items():
            if v:
                by_variant.setdefault(v, []).append(k)
        cands = [ks for ks in by_variant.values() if len(ks) == per_chain]
        if len(cands) == 1:
            names = []
            for k in cands[0]:
                m = _re.match(r"(stage\d+)\.unit(\d+)\.quant_convbn3$", k)
                names.append(f"{k}+{m[1]}.unit{int(m[2]) + 1}.quant_convbn1" if m else k.split("@")[0])
    if names and rows:
        byt = sum(launch_model(k, rows, args.batch // chains)[0] for k in names)
        mac = sum(launch_model(k, rows, args.batch // chains)[1] for k in names)
        t = len(names) * rec["avg_us"] * 1e-6
        if names and t > 0:
            rec = dict(rec, launches_in_this_plan_per_chain=len(names), plan_bytes_of_those_launches=int(byt),
                       hbm_frac_plan_bytes=round(byt / t / 1e9 / roofline.HBM_PEAK_GBS, 4),
                       mfma_frac_plan_macs=round(2 * mac / t / (roofline.MFMA_I8_PEAK_TOPS * 1e12), 4),
                       note="fractions = this plan's bytes / MACs for the launches on that kernel over launches x the profiled average duration "
                            "(the profile ran the recorded plan with its concurrent sub-batch chains, so the duration includes their contention)")
    return rec


def write_per_op(path, ops, rows, batch):
    """Per-launch table: measured ms vs the fused plan's byte / MAC model of the layers each launch covers."""
    lines = ["| launch | ms | plan MB | GB/s | % of 8 TB/s | GMAC | TOPS |", "|---|---|---|---|---|---|---|"]
    for n, ms in ops:
        b, mc = launch_model(n, rows, batch)
        lines.append(f"| {n} | {ms:.4f} | {b / 1e6:.1f} | {b / ms / 1e6:.0f} | {b / ms / 1e6 / 80:.1f} | {mc / 1e9:.2f} | "
                     f"{2 * mc / ms / 1e9:.0f} |")
    lines.append(f"\nsum of launches {sum(ms for _, ms in ops):.4f} ms")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


class _stdout_to_stderr:
    """RCCL prints its version banner on STDOUT at init; this script's stdout is ONE JSON line.  Route fd 1 to fd 2
    while the collective library may print (C stdio is flushed before the descriptor is restored)."""

    def __enter__(self):
        sys.stdout.flush()
        self._libc = C.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rccl_world1_selfcheck(dev, logits):
    """Run the product's gather function (hawq_amd.dist.gather_logits -> all_gather_into_tensor) once through RCCL in a
    world of ONE rank, so that the collective code has executed on this GPU even in the N = 1 bench run.  Outside the
    timed region."""
    import torch.distributed as dist
    from hawq_amd.dist import gather_logits
    try:
        with _stdout_to_stderr():
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
            out = gather_logits(logits)   # an initialised group of one rank still runs the collective
            torch.cuda.synchronize()
            ok = bool(out.data_ptr() != logits.data_ptr() and torch.equal(out, logits))
            dist.destroy_process_group()
        return ok
    except Exception as exc:  # never let a rendezvous problem take the bench line down
        return f"failed: {type(exc).__name__}: {exc}"


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N ranks each opening an all-core OpenMP pool oversubscribe the host while the models are built and calibrated: share the cores
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_main(args, rank, world):
    """The N > 1 plumbing on CPU: gloo process group, stand-in forward (per-image independent, like the frozen network),
    weak and strong decomposition, gather through hawq_amd.dist.gather_logits, MAX-over-ranks clock, one JSON line."""
    import torch.distributed as dist
    from hawq_amd.dist import gather_logits, plans_identical, shard_bounds, share_plan
    dist.init_process_group("gloo")
    world_seen = dist.get_world_size()
    if world_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world_seen} ranks")
    # every rank "tunes" something else (as independent tuning passes do); after share_plan all hold rank 0's plan
    own = {"batch": args.batch, "chains": 2, "tiles": ".".join(str(3 + rank + i) for i in range(5)), "fused_variants": f"{1 + rank}.0",
           # chains whose choices differ are recorded one by one (IntegerEngine.export_plan): a nested list must travel too
           "per_chain": [{"tiles": ".".join(str(3 + rank + i) for i in range(5))}, {"tiles": ".".join(str(4 + rank + i) for i in range(5))}]}
    plan = share_plan(own)
    same_plan = plans_identical(plan)
    w = torch.randn(3 * 8 * 8, 10, generator=torch.Generator().manual_seed(0))

    def fwd(x):
        return x.reshape(x.shape[0], -1) @ w

    def run(strong):
        seed = 1 if strong else 1 + rank
        x = torch.randn(args.batch, 3, 8, 8, generator=torch.Generator().manual_seed(seed))
        lo, hi = shard_bounds(args.batch, rank, world) if strong else (0, args.batch)
        for _ in range(args.warmup):
            gather_logits(fwd(x[lo:hi]), args.batch if strong else None)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full = gather_logits(fwd(x[lo:hi]), args.batch if strong else None)
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if strong:    # every rank checks the gathered batch against the single-process result
            ok = torch.equal(full, fwd(x))
        else:         # ... its own block of the gathered tensor against its own forward
            ok = torch.equal(full[rank * args.batch:(rank + 1) * args.batch], fwd(x)) and full.shape[0] == world * args.batch
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        images = args.batch if strong else world * args.batch
        return dict(value=round(images * args.steps / float(t.item()), 1), ms_per_step=round(float(t.item()) / args.steps * 1e3, 4),
                    global_batch=images, batch_per_gpu=hi - lo, every_rank_parity=bool(flag.item()))

    res = {"weak": run(False), "strong": run(True)}
    if rank == 0:
        print(json.dumps({"metric": "images/sec", "value": res[args.scaling]["value"], "unit": "images/s", "n_gpus": world_seen,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": res[args.scaling]["ms_per_step"],
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
                          "data": "dry-spawn: CPU stand-in forward on gloo (launch / shard / gather plumbing only, not a measurement)",
                          "config": {"workload": "dry_spawn_stub", "ranks_seen": world_seen, "backend": "gloo", "plan": plan,
                                     "plan_identical_on_all_ranks": same_plan, "plan_is_rank0s": plan["tiles"] == "3.4.5.6.7"},
                          "weak": res["weak"], "strong": res["strong"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def run_extras(args, dev, plans, model, eng):
    """The secondary workloads of the bench line (uint8 input, RCCL world-of-one check, the other schedules / networks, MobileNetV2, the
    strong-scaling shards).  Runs in a CHILD process of the default bench (`--extras-child`): the headline measurement and its JSON line
    must not depend on ten more engines building, capturing and destroying hipGraphs and an RCCL communicator coming and going in the same
    address space (round 6: one full bench run in ~10 died in glibc's heap check - "corrupted size vs. prev_size" - right after the RCCL
    self-check; the headline had been measured by then but its line was lost with the process)."""
    from hawq_amd import roofline
    if os.environ.get("HAWQ_BENCH_FAIL_CHILD"):   # test hook: the child dies the way the sporadic abort did (the parent must still print its line)
        os.abort()
    extra = {}
    n2 = max(10, args.steps // 2)
    # the same workload fed with uint8 NHWC images: table look-up input quantiser, 19 MB instead of
    # 77 MB of input per batch; logits are bit-identical to the fp32-tensor path (tests/test_gpu_network.py)
    xu8 = torch.randint(0, 256, (args.batch, 224, 224, 3), dtype=torch.uint8, device=dev)
    # parity of this line: the same images as the fp32 tensor the reference's host pipeline builds (ToTensor + Normalize,
    # quant_train.py:432-440, on the CPU in fp32) through the timed engine - whose fp32 path the oracle fixture pins above
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    t32 = xu8.cpu().permute(0, 3, 1, 2).to(torch.float32).div(255)
    t32 = t32.sub_(torch.tensor(mean).view(1, 3, 1, 1)).div_(torch.tensor(std).view(1, 3, 1, 1))
    ref_u8 = eng(t32.to(dev)).clone()
    u8_equal = bool(torch.equal(eng.forward_uint8(xu8, mean, std), ref_u8))
    del t32, ref_u8
    # The GPU idled (and clocked down) during the CPU baseline: rounds 4-5 timed this line after 10 warm-up forwards (13 ms) and so on a part
    # still ramping its clocks - spin_up()'s own finding - while the headline above had a spun-up part; the driver then saw uint8 at 0.97 x
    # of fp32.  Now: the same spin-up as every other line, and the fp32 forward of the SAME engine timed beside it in
    # alternating blocks (fp32, uint8, fp32, uint8, ...), so that the ratio of the two inputs is measured in one thermal / clock state.
    spin_up(eng)

    def block(u8, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(eng.stream):
            for _ in range(n):
                eng.run_resident(u8=u8)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    nblk = max(5, n2 // 4)
    t_f32 = t_u8 = 0.0
    block(False, 5), block(True, 5)
    for _ in range(4):
        t_f32 += block(False, nblk)
        t_u8 += block(True, nblk)
    extra[f"{args.arch}_{args.scheme}_b{args.batch}_uint8_input"] = {
        "images_per_s": round(args.batch * 4 * nblk / t_u8, 1),
        "fp32_input_interleaved_images_per_s": round(args.batch * 4 * nblk / t_f32, 1),
        "uint8_over_fp32_interleaved": round(t_f32 / t_u8, 4),
        "gpu_logits_bit_equal": u8_equal, "parity_against": "the same images as normalised fp32 tensor through the timed (oracle-checked) engine"}
    extra["rccl_world1_gather_ok"] = rccl_world1_selfcheck(dev, eng.logits)
    del eng, model, xu8
    torch.cuda.empty_cache()
    for arch, scheme in (("resnet50", "uniform4"), ("resnet50", "bops_0.5"), ("resnet18", "uniform8"), ("resnet101", "uniform8"), ("resnet50b", "uniform8")):
        if (arch, scheme) == (args.arch, args.scheme):
            continue
        m2, e2, x2 = setup_workload(arch, scheme, args.batch, dev, seed=1, plans=plans)
        w2, g2, b2 = timed_steps(e2, n2, 5, 1)
        alg2 = roofline.algorithmic_bytes(arch, scheme, args.batch)
        extra[f"{arch}_{scheme}_b{args.batch}"] = {
            "images_per_s": round(args.batch * n2 / w2, 1), "gpu_ms": round(g2, 4), "gpu_ms_std": b2["std_ms"],
            "hbm_frac": round(alg2 / (g2 * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS, 4), "overflow": e2.overflowed(),
            "gpu_logits_bit_equal": golden_parity(arch, scheme, args.batch, 1, e2.logits),
            "concurrent_sub_batches": e2.chains, "plan_source": e2.plan_source}
        del m2, e2, x2
        torch.cuda.empty_cache()
    # MobileNetV2 (w1, W8A8) through its own fused integer plan (hawq_amd/engine_mbv2.py).  Checks beside the
    # number: all 128 x 1000 logits against the CPU oracle's fixture, and plan vs the module-by-module path (independent
    # kernels and fp32 glue) on the first 8 images - identical output integers (tests/test_gpu_network.py pins both to the
    # live reference's per-layer digests)
    extra["mobilenetv2_w1_uniform8_b%d" % args.batch] = mobilenet_line(args.batch, dev, n2)
    # what ONE GPU runs when the batch of 128 is sharded over 2 / 4 / 8 ranks (strong scaling):
    # the first 64 / 32 / 16 images of the headline workload, same engine configuration
    if args.batch == 128:
        for nb in (64, 32, 16):
            m2, e2, x2 = setup_workload(args.arch, args.scheme, args.batch, dev, seed=1, shard=(0, nb), plans=plans)
            w2, g2, b2 = timed_steps(e2, n2, 5, 1)
            extra[f"{args.arch}_{args.scheme}_shard_b{nb}"] = {
                "images_per_s": round(nb * n2 / w2, 1), "gpu_ms": round(g2, 4), "gpu_ms_std": b2["std_ms"],
                "gpu_logits_bit_equal": golden_parity(args.arch, args.scheme, args.batch, 1, e2.logits),
                "concurrent_sub_batches": e2.chains}
            del m2, e2, x2
            torch.cuda.empty_cache()
    return extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--scheme", default="uniform8")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step (weak) / per job per step (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="which decomposition `value` reports (N > 1 measures both)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads / per-kernel table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-op", default=None, help="write a per-launch roofline table (markdown) to this file")
    ap.add_argument("--dry-spawn", action="store_true", help="CPU / gloo stand-in forward: exercises the N > 1 launch, shard, gather and report path")
    ap.add_argument("--plan", default=os.path.join(ROOT, "profiles", "plans.json"),
                    help="recorded plans by workload (replayed instead of tuning when the workload has an entry)")
    ap.add_argument("--retune", action="store_true", help="ignore the recorded plans: tune every engine in this run")
    ap.add_argument("--save-plan", default=None, help="write the plans this run used / tuned to this file")
    ap.add_argument("--cpu-sample", type=int, default=64, help="images of the benchmarked batch the CPU baseline times")
    ap.add_argument("--extras-child", action="store_true", help="(internal) run only the secondary workloads and print them as one JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_spawn:
        return dry_main(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)   # communicator set-up (and RCCL's banner) happens at the first collective
            torch.cuda.synchronize()
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but RCCL sees {dist.get_world_size()} ranks")

    from hawq_amd import roofline
    from hawq_amd.dist import shard_bounds

    if args.batch % world:
        raise SystemExit("the batch must be divisible by the number of ranks (strong scaling shards ONE batch)")

    def all_ranks(ok):   # a parity flag counts only if it holds on every rank
        if world == 1 or ok is None:
            return ok
        f = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item())

    plans = Plans(args.plan, args.retune)
    # ---- weak: every rank its own batch (rank 0's is the fixture workload)
    seed = 1 + rank
    model, eng, x = setup_workload(args.arch, args.scheme, args.batch, dev, seed=seed, plans=plans, share=world > 1)
    from hawq_amd.dist import plans_identical
    same_plan = plans_identical(eng.export_plan())
    local_batch = args.batch
    wall, gpu_ms, blocks = timed_steps(eng, args.steps, args.warmup, world)
    overflow = eng.overflowed()
    parity = golden_parity(args.arch, args.scheme, args.batch, seed, eng.logits) if rank == 0 else None
    runs = {"weak": dict(value=round(world * args.batch * args.steps / wall, 1), ms_per_step=round(wall / args.steps * 1e3, 4),
                         global_batch=world * args.batch, batch_per_gpu=args.batch)}
    # ---- strong (N > 1 only): ONE batch of `batch` images, rank r evaluates images shard_bounds(batch, r, N)
    if world > 1:
        lo, hi = shard_bounds(args.batch, rank, world)
        # the SAME model object (built and calibrated once), a second engine for the shard's batch shape; one shared plan again
        m2, e2, x2 = setup_workload(args.arch, args.scheme, args.batch, dev, seed=1, shard=(lo, hi), plans=plans, model=model, share=True)
        same_plan = same_plan and plans_identical(e2.export_plan())
        w2, g2, b2 = timed_steps(e2, args.steps, args.warmup, world, batch_total=args.batch)
        p2 = all_ranks(golden_parity(args.arch, args.scheme, args.batch, 1, e2.logits, lo))
        runs["strong"] = dict(value=round(args.batch * args.steps / w2, 1), ms_per_step=round(w2 / args.steps * 1e3, 4),
                              global_batch=args.batch, batch_per_gpu=hi - lo, gpu_ms_per_step=round(g2, 4),
                              every_rank_logits_bit_equal_oracle=p2, concurrent_sub_batches=e2.chains)
        del m2, e2, x2
        torch.cuda.empty_cache()
    scaling = args.scaling if args.scaling in runs else "weak"
    head = runs[scaling]

    out = None
    if rank == 0:
        alg = roofline.algorithmic_bytes(args.arch, args.scheme, local_batch)
        macs = roofline.macs(args.arch, args.scheme, local_batch)
        gbs = alg / (gpu_ms * 1e-3) / 1e9
        fused_pairs = [n for n, v in eng.er_choice.items() if v and not n.endswith("@solo")]
        plan_bytes = roofline.fused_plan_bytes(args.arch, args.scheme, local_batch, fused_pairs)
        plan = dict(tiles=".".join(str(t) for t in eng.tile_choice.values()), fused_variants=".".join(str(t) for t in eng.er_choice.values()),
                    chains=eng.chains)
        # HBM bytes per launch from the committed PMC passes (counters cannot be collected in-process): recorded with the plan
        # (tile ids / fused variants / chains) they were collected for; this run's plan is printed beside it
        traffic, traffic_meta = None, {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic_meta = json.load(f).get(f"{args.arch}_{args.scheme}_b{local_batch}", {})
                traffic = traffic_meta.get("bytes_per_launch")
        except OSError:
            pass
        mfma_frac = 2 * macs / (gpu_ms * 1e-3) / (roofline.MFMA_I8_PEAK_TOPS * 1e12)
        hbm_traffic_frac = traffic / (gpu_ms * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS if traffic else None
        floor_ms = plan_bytes / (COPY_CEILING_GBS * 1e9) * 1e3
        out = {
            "metric": "images/sec", "value": head["value"], "unit": "images/s", "n_gpus": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{args.arch}_{args.scheme}_b{args.batch}", "arch": args.arch,
                       "scheme": args.scheme, "batch_per_gpu": head["batch_per_gpu"], "global_batch": head["global_batch"],
                       "image": 224, "parallelism": f"dp{world}", "weights": "synthetic seed 0, ranges calibrated on 8 images",
                       "residual_uint16_overflow": overflow,
                       "fast_contract_conv_launches": f"{eng.n_fast}/{eng.n_conv}", "exact_tie_requant_launches": eng.n_tie,
                       "per_channel_k0_launches": getattr(eng, "n_ck0", None),
                       "autotuned_tiles": plan["tiles"],
                       # candidate expand->reduce pairs: variant id of the fused launch, 0 = two separate launches were faster
                       "fused_expand_reduce_launches": len(fused_pairs), "fused_variants": plan["fused_variants"],
                       "fused_pairs": fused_pairs,
                       # expand convs without a fusable successor that run the wave-private kernel (fused_wp.hip) alone
                       "wave_private_solo_launches": [n[:-5] for n, v in eng.er_choice.items() if v and n.endswith("@solo")],
                       # ms per forward of the independently tuned plans the engine chose between
                       "plan_trials_ms": getattr(eng, "plan_trials_ms", None),
                       "chain_timing_ms": {str(k): round(v, 4) for k, v in getattr(eng, "chain_timing_ms", {}).items()},
                       "fused_split_tiles": ".".join(f"{a}.{b}" for a, b in getattr(eng, "er_split_tiles", {}).values()),
                       "concurrent_sub_batches": eng.chains,
                       # where the plan came from (profiles/plans.json replayed, or tuned here) and, N > 1, whether every rank runs it
                       "plan_source": eng.plan_source, "plan_identical_on_all_ranks": same_plan},
            # all logits of rank 0's images against the CPU oracle's golden logits of the same workload
            "parity": {"gpu_logits_bit_equal_oracle": parity, "images_compared": local_batch if parity is not None else 0,
                       "fixture": f"tests/golden/b128_{args.arch}_{args.scheme}.npz"},
            "timing": dict(blocks, gpu_ms_per_step=round(gpu_ms, 4),
                           note="HIP events on the engine stream between steps of the ONE timed region (weak run)"),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / roofline.HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": ("committed PMC passes (profiles/traffic.json, git head "
                                            f"{traffic_meta.get('git_head', '?')}), not collected by this run; FETCH_SIZE x 2 + WRITE_SIZE count "
                                            "requests at the L2's memory side, Infinity-Cache hits included: an UPPER bound on HBM bytes "
                                            "(calibration: profiles/r03_pmc_calibration.md)") if traffic else None,
                         "traffic_plan": traffic_meta.get("plan") or traffic_meta.get("config"), "this_run_plan": plan,
                         "traffic_plan_matches_this_run": (traffic_meta.get("plan") == plan) if traffic else None,
                         # counter bytes / time / 8 TB/s: upper bound on the physical bandwidth utilisation ("frac" prices the
                         # canonical SURVEY 8(d) byte model, which the fused plan undercuts)
                         "hbm_traffic_frac": round(hbm_traffic_frac, 4) if hbm_traffic_frac else None,
                         "bound_note": "the schema offers hbm | mfma: HBM is the nearer roofline (hbm_traffic_frac vs mfma_frac), but the "
                                       "counters put the forward well below BOTH - see limiter and plan_floor_ratio",
                         "limiter": "neither roofline: epilogue VALU (~14-17 instructions per residual output), per-launch latency of one "
                                    "workgroup in stages 3-4, lock-step phases inside workgroups (DESIGN.md 5)",
                         "kernel": "one hipGraph launch = whole forward of one batch",
                         "gpu_ms_per_launch": round(gpu_ms, 4), "algorithmic_bytes_per_launch": alg,
                         # what the fused plan must move at minimum (hawq_amd/roofline.py:fused_plan_table); "traffic" is its
                         # measured counterpart; floor = those bytes at the measured copy ceiling of the part
                         "fused_plan_bytes_per_launch": plan_bytes,
                         "plan_floor_ms": round(floor_ms, 4), "plan_floor_ratio": round(gpu_ms / floor_ms, 3),
                         # counter traffic over the bytes the plan has to move (> 1: re-reads / Infinity-Cache-side requests beyond the model)
                         "traffic_over_plan_bytes": round(traffic / plan_bytes, 3) if traffic and plan_bytes else None,
                         "mfma_frac": round(mfma_frac, 4)},
        }
        try:
            out["roofline"]["dominant_kernel"] = dominant_kernel(eng, args, plan_rows(args.arch, args.scheme) if args.arch in roofline.ARCH else {})
        except Exception as exc:   # a reporting extra must never cost the bench line
            out["roofline"]["dominant_kernel"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world > 1:
            out["weak"], out["strong"] = runs["weak"], runs.get("strong")
        if (not args.no_extra or args.per_op) and world == 1:
            ops = eng.profile_ops()
            tot = sum(ms for _, ms in ops)
            rows = plan_rows(args.arch, args.scheme)
            top = sorted(ops, key=lambda t: -t[1])[:8]
            # eager launches timed one by one; with concurrent sub-batches this is sub-batch 0 alone
            out["roofline"]["eager_sum_ms"] = round(tot, 4)
            nb = args.batch if eng.chains == 1 else eng.subs[0]._batch[0]
            out["roofline"]["eager_sum_batch"] = nb
            out["roofline"]["top_launches"] = [{"name": n, "ms": round(ms, 4)} for n, ms in top]
            # the single most expensive kernel launch against the bytes / MACs the FUSED plan moves for it
            dn, dms = top[0]
            db, dm = launch_model(dn, rows, nb)
            out["roofline"]["dominant_launch"] = {
                "name": dn, "ms": round(dms, 4), "batch": nb,
                "plan_bytes": db, "achieved_GBps": round(db / dms / 1e6, 1),
                "hbm_frac": round(db / dms / 1e6 / roofline.HBM_PEAK_GBS, 4),
                "mfma_frac": round(2 * dm / (dms * 1e-3) / (roofline.MFMA_I8_PEAK_TOPS * 1e12), 4)}
            if args.per_op:
                write_per_op(args.per_op, ops, rows, nb)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, x, eng.logits, sample=args.cpu_sample)
        else:
            out["cpu_baseline"] = None
    if not args.no_extra and world == 1 and rank == 0:
        if args.extras_child:
            out = {"extra": run_extras(args, dev, plans, model, eng), "plans_used": plans.used}
            print(json.dumps(out), flush=True)
            return
        # free the headline engine first: the child builds its own by REPLAYING the plan this process ran (recorded or tuned just now);
        # the other workloads replay their recorded plans - or are tuned by the child when this run was asked to --retune
        del eng, model
        torch.cuda.empty_cache()
        import tempfile
        child_plans = {} if args.retune else dict(plans.loaded)
        child_plans.update(plans.used)
        tmp = tempfile.NamedTemporaryFile("w", suffix="_plans.json", delete=False)
        json.dump(child_plans, tmp)
        tmp.close()
        cmd = [sys.executable, os.path.abspath(__file__), "--extras-child", "--no-cpu-baseline", "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--arch", args.arch, "--scheme", args.scheme, "--batch", str(args.batch), "--plan", tmp.name]
        extra = None
        for attempt in (1, 2):   # one retry: the failure seen was sporadic
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, timeout=900,
                                   env=dict(os.environ, HAWQ_PLAN_LABEL=(os.path.relpath(args.plan, ROOT) if not args.retune else "the plans this run tuned")))
                line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
                if r.returncode == 0 and line:
                    child = json.loads(line[-1])
                    extra = child["extra"]
                    plans.used.update(child.get("plans_used", {}))
                    if attempt > 1:
                        extra["extras_child_attempts"] = attempt
                    break
                err = f"child exited with code {r.returncode}"
            except subprocess.TimeoutExpired as exc:   # a hang is not retried: the bench must end in bounded time
                extra = {"error": f"the secondary workloads did not complete (timeout after {exc.timeout:.0f} s); the headline above is unaffected"}
                break
            except Exception as exc:   # unparsable output
                err = f"{type(exc).__name__}: {exc}"
            extra = {"error": f"the secondary workloads did not complete ({err}); the headline above is unaffected"}
        os.unlink(tmp.name)
        out["extra"] = extra
    if args.save_plan and rank == 0:
        plans.save(args.save_plan)
    if world > 1:
        with _stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
