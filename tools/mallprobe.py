"""Bounds the Infinity-Cache batch-tiling idea before anything is built: does a stage's launch group
get cheaper PER IMAGE when the batch is walked depth-first in sub-batches whose residual tensors fit the 256 MiB cache?

For S in (128, 64, 32, 16): 128 / S independent single-chain engines of S images each (own buffers, own tuned tiles) run
the SAME launch group (stem, stage1, ... or a range of stages) one sub-batch after the other on one stream, i.e. exactly
the launches a depth-first schedule would issue.  Prints us per 128 images for each group and S.

Usage: python tools/mallprobe.py [--conc]   (--conc: also two sub-batches at a time on two streams)"""
import ctypes as C
import os
import sys

import torch

os.environ["HAWQ_CHAINS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hawq_amd import _lib  # noqa: E402
from hawq_amd.api import build_quantized_resnet, calibrate  # noqa: E402
from hawq_amd.engine import IntegerEngine  # noqa: E402
from hawq_amd.skeleton import synthetic_images  # noqa: E402

dev = torch.device("cuda:0")
model = build_quantized_resnet("resnet50", "uniform8", seed=0).to(dev)
calibrate(model, synthetic_images(8, seed=0).to(dev))
x = synthetic_images(128, seed=1).to(dev)

GROUPS = [("stem", ("hawq_stem",)), ("stage1", ("stage1",)), ("stage2", ("stage2",)), ("stage3", ("stage3",)),
          ("stage4", ("stage4",)), ("stem+s1", ("hawq_stem", "stage1")), ("stem+s1+s2", ("hawq_stem", "stage1", "stage2")),
          ("s3+s4+tail", ("stage3", "stage4", "hawq_avgpool", "quant_output")), ("all", ("",))]


def group_ops(eng, prefixes):
    return [op for op, n in zip(eng._ops, eng._ops.names) if n.startswith(prefixes)]


def run_seq(engines, prefixes, main):
    """every engine's group, one engine after the other, all ordered on `main`"""
    for e in engines:
        e.stream.wait_stream(main)
        with torch.cuda.stream(e.stream):
            for op in group_ops(e, prefixes):
                op()
        main.wait_stream(e.stream)


def run_conc(engines, prefixes, main, lanes):
    """`lanes` sub-batches at a time: engine i runs behind engine i - lanes"""
    last = [None] * lanes
    for i, e in enumerate(engines):
        lane = i % lanes
        if last[lane] is None:
            e.stream.wait_stream(main)
        else:
            e.stream.wait_stream(last[lane].stream)
        with torch.cuda.stream(e.stream):
            for op in group_ops(e, prefixes):
                op()
        last[lane] = e
    for e in last:
        if e is not None:
            main.wait_stream(e.stream)


def time_it(fn, reps=10):
    """us per run of fn, captured once into a hipGraph (no host launch overhead in the figure), best of 3 x reps replays"""
    main = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    with torch.cuda.stream(main):
        fn(main)
        torch.cuda.synchronize()
        g = C.c_void_p()
        _lib.call("hawq_graph_begin", main.cuda_stream)
        try:
            fn(main)
        finally:
            _lib.call("hawq_graph_end", main.cuda_stream, C.byref(g))
        for _ in range(2):
            _lib.call("hawq_graph_launch", g, main.cuda_stream)
        for _ in range(3):
            e0.record(main)
            for _ in range(reps):
                _lib.call("hawq_graph_launch", g, main.cuda_stream)
            e1.record(main)
            e1.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e3
            best = t if best is None else min(best, t)
        _lib.call("hawq_graph_destroy", g)
    return best


SIZES = tuple(int(v) for v in os.environ.get("MALL_SIZES", "128,64,32,16").split(","))
results = {}
for S in SIZES:
    engines = []
    for k in range(128 // S):
        e = IntegerEngine(model, use_graph=False, chains=1)
        e(x[k * S:(k + 1) * S])
        engines.append(e)
    torch.cuda.synchronize()
    for gname, pre in GROUPS:
        results[(gname, S, 1)] = time_it(lambda main: run_seq(engines, pre, main))
        if "--conc" in sys.argv and len(engines) >= 2:
            results[(gname, S, 2)] = time_it(lambda main: run_conc(engines, pre, main, 2))
    print(f"S={S}: " + "  ".join(f"{g}={results[(g, S, 1)]:.0f}" + (f"/{results[(g, S, 2)]:.0f}" if (g, S, 2) in results else "")
                                 for g, _ in GROUPS), flush=True)
    del engines
    torch.cuda.empty_cache()

print("\n| launch group | " + " | ".join(f"{128 // S} x {S} img" for S in SIZES) + " |  (us per 128 images, sequential"
      + (" / two at a time" if "--conc" in sys.argv else "") + ")")
print("|---|" + "---|" * len(SIZES))
for g, _ in GROUPS:
    print(f"| {g} | " + " | ".join(f"{results[(g, S, 1)]:.0f}" + (f" / {results[(g, S, 2)]:.0f}" if (g, S, 2) in results else "")
                                  for S in SIZES) + " |")
