"""Experiment: the sub-batch chains of one forward as free-running streams (each chain its own hipGraph, no fork / join
per step) with a fixed start offset between them, against the fork/join graph the engine replays today.
Usage: python tools/freerun_probe.py [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
model, eng, x = bench.setup_workload("resnet50", "uniform8", 128, dev, 1)
eng(x)
torch.cuda.synchronize()
print("chains", len(eng.subs), "parity", bench.golden_parity("resnet50", "uniform8", 128, 1, eng.logits))


def run_joined(k):
    with torch.cuda.stream(eng.stream):
        for _ in range(k):
            eng.run_resident()


def run_free(k, offsets_cycles, graphs=True):
    if graphs:
        for sub, off in zip(eng.subs, offsets_cycles):
            with torch.cuda.stream(sub.stream):
                if off:
                    torch.cuda._sleep(int(off))
                for _ in range(k):
                    sub.run_resident()
        return
    for sub, off in zip(eng.subs, offsets_cycles):
        if off:
            with torch.cuda.stream(sub.stream):
                torch.cuda._sleep(int(off))
    for _ in range(k):   # direct launches, the chains interleaved on the host
        for sub in eng.subs:
            sub._launch_all()


for sub in eng.subs:   # capture the per-chain graphs
    with torch.cuda.stream(sub.stream):
        sub.run_resident()
torch.cuda.synchronize()
for rep in range(2):
    for name, fn in [("joined", lambda: run_joined(steps))] + [
            (f"free offset {ms:.2f} ms", (lambda ms=ms: run_free(steps, [0, ms * 2.1e6]))) for ms in (0.0, 0.5)] + [
            (f"direct launches, offset {ms:.2f} ms", (lambda ms=ms: run_free(steps, [0, ms * 2.1e6], False))) for ms in (0.0, 0.35, 0.7)] + [
            ("direct launches, joined", lambda: [eng._launch_all() for _ in range(steps)])]:
        fn() if False else None
        torch.cuda.synchronize()
        eng.logits.zero_()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bench.golden_parity("resnet50", "uniform8", 128, 1, eng.logits)
        print(f"{name:24s} {dt / steps * 1e3:.4f} ms/step  {128 * steps / dt:9.1f} img/s  parity={ok}", flush=True)
