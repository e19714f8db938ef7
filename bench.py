#!/usr/bin/env python
"""Throughput bench of the MI355X integer forward (BASELINE.json metric: images/s, ResNet50
W8A8 at batch 128 per GPU; W4A4 / mixed / ResNet18 reported in ``extra``).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--arch resnet50] [--scheme uniform8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frozen forward of one batch: fp32 images resident in HBM -> fp32 logits in HBM
(one hipGraph launch of the fused integer plan).  With N > 1 the path shards by batch, no data-path
collective, and each step ends with one RCCL all_gather of the logits (the reference's DataParallel
gather, quant_train.py:358):
  --scaling weak   (default) every rank processes its own batch of 128; value = N * 128 * K / t
  --scaling strong ONE batch of 128 is sharded 128/N images per rank; value = 128 * K / t
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def setup_workload(arch, scheme, batch, dev, seed, shard=None):
    """Synthetic weights (seed 0), ranges calibrated on 8 synthetic images, `batch` synthetic images (seed);
    ``shard = (lo, hi)`` keeps only that slice of the batch (strong scaling: this rank's images)."""
    from hawq_amd.api import build_quantized_resnet, calibrate
    from hawq_amd.engine import IntegerEngine
    from hawq_amd.skeleton import synthetic_images

    model = build_quantized_resnet(arch, scheme, seed=0).to(dev)
    calibrate(model, synthetic_images(8, seed=0).to(dev))
    eng = IntegerEngine(model, use_graph=True)
    x = synthetic_images(batch, seed=seed)
    if shard is not None:
        x = x[shard[0]:shard[1]]
    x = x.to(dev)
    eng(x)  # allocate, autotune, warm up, capture the hipGraph
    return model, eng, x


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


def timed_steps(eng, steps, warmup, world, gathered):
    """W untimed + K timed steps on the engine stream; returns (wall seconds, GPU ms per step, block stats).
    Block stats: the K steps are cut into >= 10 blocks (HIP events recorded between steps, no synchronisation) so that
    a run reports mean +- std of the per-step time (tvm_benchmark/test_resnet_inference_time.py:257-271 protocol)."""
    import torch.distributed as dist
    from hawq_amd import _lib

    sp = eng.stream.cuda_stream
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    _lib.call("hawq_event_create", C.byref(ev0))
    _lib.call("hawq_event_create", C.byref(ev1))
    blk = max(1, steps // 10)
    marks = []

    def step():
        eng.run_resident()
        if world > 1:
            dist.all_gather_into_tensor(gathered, eng.logits)

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
    return t1 - t0, total / steps, stats


def cpu_baseline(model, x, gpu_logits, budget_s=25.0):
    """Time the CPU fake-quant port (oracle/fakequant_port.py: the reference's frozen forward with its cost structure -
    per-forward BN fold + weight re-quantisation + Decimal batch_frexp) on the host cores at the FULL batch of the
    workload (B = 128): one warm-up forward of 2 images (thread pools, oneDNN primitives), then whole
    forwards until `budget_s` is used (at least 1, at most 3); its logits double as a parity check of the GPU result."""
    from oracle import fakequant_port, oracle

    st = oracle.extract_float_state(model)
    xs = x.cpu()
    fakequant_port.forward(st, xs[:2])
    runs, y = [], None
    while len(runs) < 3 and (not runs or sum(runs) + runs[-1] <= budget_s):
        t0 = time.perf_counter()
        y = fakequant_port.forward(st, xs)
        runs.append(time.perf_counter() - t0)
    n = xs.shape[0]
    parity = bool(torch.equal(y, gpu_logits.cpu()))
    return dict(value=round(n * len(runs) / sum(runs), 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(runs)} whole forward(s) of the same batch of {n} after a 2-image warm-up (torch-CPU fp32 "
                       f"fake-quant port of the reference path; seconds per forward: {[round(r, 1) for r in runs]})",
                gpu_logits_bit_equal=parity, images_compared=n)


def launch_model(name, rows, batch):
    """(algorithmic bytes, MACs) of one launch for `batch` images: the canonical per-layer model
    (hawq_amd/roofline.py:layer_table) summed over the layers the launch covers."""
    parts = []
    base = name.split("+")[0]
    if base in rows:
        parts.append(rows[base])
    if name.endswith("+identity"):
        parts.append(rows[base.rsplit(".", 1)[0] + ".quant_identity_convbn"])
    for extra in name.split("+")[1:]:   # a fused launch covers a second layer (expand conv + next unit's reduce conv)
        if extra in rows:
            parts.append(rows[extra])
    if name in ("hawq_quantize_input", "hawq_stem_fused"):
        parts.append(rows["quant_input"])
    if name in ("hawq_stem_conv7", "hawq_stem_fused"):
        parts += [r for k, r in rows.items() if k.startswith("quant_init")]
    if name == "hawq_avgpool_requant":
        parts.append(rows["final_pool+quant_act_output"])
    return (sum(r["act_bytes"] for r in parts) * batch + sum(r["weight_bytes"] for r in parts),
            sum(r["macs"] for r in parts) * batch)


def write_per_op(path, ops, rows, batch):
    """Per-launch table: measured ms vs the canonical byte/MAC model of the layers each launch covers."""
    def model(name):
        return launch_model(name, rows, batch)
    lines = ["| launch | ms | model MB | GB/s | % of 8 TB/s | GMAC | TOPS |", "|---|---|---|---|---|---|---|"]
    for n, ms in ops:
        b, mc = model(n)
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


def rccl_world1_selfcheck(dev, logits):
    """Run the gather path (hawq_amd.dist.gather_logits -> all_gather_into_tensor) once through RCCL in a world of ONE
    rank, so that the collective code has executed on this GPU even in the N = 1 bench run.  Outside the timed region."""
    import socket
    import torch.distributed as dist
    try:
        with _stdout_to_stderr():
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            out = torch.empty_like(logits)
            dist.all_gather_into_tensor(out, logits.contiguous())
            torch.cuda.synchronize()
            ok = bool(torch.equal(out, logits))
            dist.destroy_process_group()
        return ok
    except Exception as exc:  # never let a rendezvous problem take the bench line down
        return f"failed: {type(exc).__name__}: {exc}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--scheme", default="uniform8")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step (weak) / per job per step (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads / per-kernel table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-op", default=None, help="write a per-launch roofline table (markdown) to this file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    from hawq_amd import roofline
    from hawq_amd.dist import shard_bounds

    strong = args.scaling == "strong"
    if strong and args.batch % world:
        raise SystemExit("--scaling strong needs a batch divisible by the number of ranks")
    lo, hi = shard_bounds(args.batch, rank, world) if strong else (0, args.batch)
    seed = 1 if strong else 1 + rank
    model, eng, x = setup_workload(args.arch, args.scheme, args.batch, dev, seed=seed, shard=(lo, hi) if strong else None)
    local_batch = hi - lo
    gathered = torch.empty(world * local_batch, eng.logits.shape[1], device=dev) if world > 1 else None
    wall, gpu_ms, blocks = timed_steps(eng, args.steps, args.warmup, world, gathered)
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    overflow = eng.overflowed()
    ms_per_step = wall / args.steps * 1e3
    global_batch = args.batch if strong else world * args.batch
    value = global_batch * args.steps / wall
    parity = golden_parity(args.arch, args.scheme, args.batch, seed, eng.logits, lo)

    out = None
    if rank == 0:
        alg = roofline.algorithmic_bytes(args.arch, args.scheme, local_batch)
        macs = roofline.macs(args.arch, args.scheme, local_batch)
        gbs = alg / (gpu_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (counters cannot be collected in-process); valid for the
        # tile / sub-batch choice and the git head recorded beside it
        traffic, traffic_meta = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic_meta = json.load(f).get(f"{args.arch}_{args.scheme}_b{local_batch}", {})
                traffic = traffic_meta.get("bytes_per_launch")
        except OSError:
            pass
        mfma_frac = 2 * macs / (gpu_ms * 1e-3) / (roofline.MFMA_I8_PEAK_TOPS * 1e12)
        fused_pairs = [n for n, v in eng.er_choice.items() if v and not n.endswith("@solo")]
        hbm_traffic_frac = traffic / (gpu_ms * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS if traffic else None
        out = {
            "metric": "images/sec", "value": round(value, 1), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{args.arch}_{args.scheme}_b{args.batch}", "arch": args.arch,
                       "scheme": args.scheme, "batch_per_gpu": local_batch, "global_batch": global_batch,
                       "image": 224, "parallelism": f"dp{world}", "weights": "synthetic seed 0, ranges calibrated on 8 images",
                       "residual_uint16_overflow": overflow,
                       "fast_contract_conv_launches": f"{eng.n_fast}/{eng.n_conv}", "exact_tie_requant_launches": eng.n_tie,
                       "autotuned_tiles": ".".join(str(t) for t in eng.tile_choice.values()),
                       # candidate expand->reduce pairs: variant id of the fused launch, 0 = two separate launches were faster
                       "fused_expand_reduce_launches": len(fused_pairs), "fused_variants": ".".join(str(t) for t in eng.er_choice.values()),
                       "fused_pairs": fused_pairs,
                       # expand convs without a fusable successor that run the wave-private kernel (fused_wp.hip) alone
                       "wave_private_solo_launches": [n[:-5] for n, v in eng.er_choice.items() if v and n.endswith("@solo")],
                       # ms per forward of the independently tuned plans the engine chose between
                       "plan_trials_ms": getattr(eng, "plan_trials_ms", None),
                       "fused_split_tiles": ".".join(f"{a}.{b}" for a, b in getattr(eng, "er_split_tiles", {}).values()),
                       "concurrent_sub_batches": eng.chains},
            # all logits of this rank's images against the CPU oracle's golden logits of the same workload
            "parity": {"gpu_logits_bit_equal_oracle": parity, "images_compared": local_batch if parity is not None else 0,
                       "fixture": f"tests/golden/b128_{args.arch}_{args.scheme}.npz"},
            "timing": dict(blocks, gpu_ms_per_step=round(gpu_ms, 4),
                           note="HIP events on the engine stream between steps of the ONE timed region"),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / roofline.HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": ("committed PMC passes (profiles/traffic.json, git head "
                                            f"{traffic_meta.get('git_head', 'r01')}), not collected by this run") if traffic else None,
                         # measured HBM bytes / time / 8 TB/s: the PHYSICAL bandwidth utilisation ("frac" prices the
                         # canonical SURVEY 8(d) byte model, which the fused plan undercuts)
                         "hbm_traffic_frac": round(hbm_traffic_frac, 4) if hbm_traffic_frac else None,
                         "limiter": "neither roofline: HBM and MFMA utilisation are both well below 1 - the forward is bound "
                                    "by epilogue VALU, per-launch fill/drain and occupancy (DESIGN.md 5)",
                         "kernel": "one hipGraph launch = whole forward of one batch",
                         "gpu_ms_per_launch": round(gpu_ms, 4), "algorithmic_bytes_per_launch": alg,
                         # what the fused plan must move at minimum (hawq_amd/roofline.py:fused_plan_table); "traffic"
                         # is its measured counterpart
                         "fused_plan_bytes_per_launch": roofline.fused_plan_bytes(args.arch, args.scheme, local_batch,
                                                                                  fused_pairs),
                         "mfma_frac": round(mfma_frac, 4)},
        }
        if (not args.no_extra or args.per_op) and world == 1:
            ops = eng.profile_ops()
            tot = sum(ms for _, ms in ops)
            rows = {r["name"]: r for r in roofline.layer_table(args.arch, args.scheme)}
            top = sorted(ops, key=lambda t: -t[1])[:8]
            # eager launches timed one by one; with concurrent sub-batches this is sub-batch 0 alone
            out["roofline"]["eager_sum_ms"] = round(tot, 4)
            out["roofline"]["eager_sum_batch"] = args.batch if eng.chains == 1 else eng.subs[0]._batch[0]
            out["roofline"]["top_launches"] = [{"name": n, "ms": round(ms, 4)} for n, ms in top]
            # the single most expensive kernel launch against its own byte / MAC model
            dn, dms = top[0]
            db, dm = launch_model(dn, rows, out["roofline"]["eager_sum_batch"])
            out["roofline"]["dominant_launch"] = {
                "name": dn, "ms": round(dms, 4), "batch": out["roofline"]["eager_sum_batch"],
                "algorithmic_bytes": db, "achieved_GBps": round(db / dms / 1e6, 1),
                "hbm_frac": round(db / dms / 1e6 / roofline.HBM_PEAK_GBS, 4),
                "mfma_frac": round(2 * dm / (dms * 1e-3) / (roofline.MFMA_I8_PEAK_TOPS * 1e12), 4)}
            if args.per_op:
                write_per_op(args.per_op, ops, rows, out["roofline"]["eager_sum_batch"])
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, x, eng.logits)
        else:
            out["cpu_baseline"] = None
    if not args.no_extra and world == 1 and rank == 0:
        extra = {}
        n2 = max(10, args.steps // 2)
        # the same workload fed with uint8 NHWC images: table look-up input quantiser, 19 MB instead of
        # 77 MB of input per batch; logits are bit-identical to the fp32-tensor path (tests/test_gpu_network.py)
        xu8 = torch.randint(0, 256, (args.batch, 224, 224, 3), dtype=torch.uint8, device=dev)
        eng.forward_uint8(xu8)
        with torch.cuda.stream(eng.stream):   # warm-up: the GPU idled (and clocked down) during the CPU baseline
            for _ in range(10):
                eng.run_resident(u8=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(eng.stream):
            for _ in range(n2):
                eng.run_resident(u8=True)
        torch.cuda.synchronize()
        extra[f"{args.arch}_{args.scheme}_b{args.batch}_uint8_input"] = {
            "images_per_s": round(args.batch * n2 / (time.perf_counter() - t0), 1)}
        extra["rccl_world1_gather_ok"] = rccl_world1_selfcheck(dev, eng.logits)
        del eng, model, xu8
        torch.cuda.empty_cache()
        for arch, scheme in (("resnet50", "uniform4"), ("resnet50", "bops_0.5"), ("resnet18", "uniform8")):
            if (arch, scheme) == (args.arch, args.scheme):
                continue
            m2, e2, x2 = setup_workload(arch, scheme, args.batch, dev, seed=1)
            w2, g2, b2 = timed_steps(e2, n2, 5, 1, None)
            alg2 = roofline.algorithmic_bytes(arch, scheme, args.batch)
            extra[f"{arch}_{scheme}_b{args.batch}"] = {
                "images_per_s": round(args.batch * n2 / w2, 1), "gpu_ms": round(g2, 4), "gpu_ms_std": b2["std_ms"],
                "hbm_frac": round(alg2 / (g2 * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS, 4), "overflow": e2.overflowed(),
                "gpu_logits_bit_equal": golden_parity(arch, scheme, args.batch, 1, e2.logits),
                "concurrent_sub_batches": e2.chains}
            del m2, e2, x2
            torch.cuda.empty_cache()
        # what ONE GPU runs when the batch of 128 is sharded over 2 / 4 / 8 ranks (strong scaling):
        # the first 64 / 32 / 16 images of the headline workload, same engine configuration
        if args.batch == 128:
            for nb in (64, 32, 16):
                m2, e2, x2 = setup_workload(args.arch, args.scheme, args.batch, dev, seed=1, shard=(0, nb))
                w2, g2, b2 = timed_steps(e2, n2, 5, 1, None)
                extra[f"{args.arch}_{args.scheme}_shard_b{nb}"] = {
                    "images_per_s": round(nb * n2 / w2, 1), "gpu_ms": round(g2, 4), "gpu_ms_std": b2["std_ms"],
                    "gpu_logits_bit_equal": golden_parity(args.arch, args.scheme, args.batch, 1, e2.logits),
                    "concurrent_sub_batches": e2.chains}
                del m2, e2, x2
                torch.cuda.empty_cache()
        out["extra"] = extra
    if world > 1:
        with _stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
