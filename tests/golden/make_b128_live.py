"""The BENCHMARKED workloads tied to the LIVE reference (build container only; VERDICT r2 "B=128 workloads tied to the
live reference").

    python tests/golden/make_b128_live.py [arch scheme ...]      # writes tests/golden/b128live_<arch>_<scheme>.npz

`bench.py:setup_workload` builds: synthetic weights (seed 0), ranges calibrated on synthetic_images(8, seed 0), input
synthetic_images(128, seed 1).  tests/golden/make_b128.py pushes that workload through the ORACLE; this script pushes the
same weights and calibration batch through the UNMODIFIED reference (oracle/ref_live.py) and evaluates images [0, 16) of the
benchmark batch with it.  Stored (as in the b2 fixtures of make_golden.py): the reference's frozen ranges, its integer
checkpoint (scales and biases in full, weight_integer as digests plus the entries torch-CPU's non-IEEE sqrt moved), per-conv
accumulator digests of the 16 images, and its logits.  tests/test_gpu_b128.py loads the checkpoint into the engine exactly as
bench.py configures it (hipGraph, autotuned tiles, chosen sub-batch chains, batch 128) and requires the reference's logits.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden  # noqa: E402
from hawq_amd.skeleton import synthetic_images  # noqa: E402

CONFIGS = [("resnet18", "uniform8"), ("resnet50", "uniform8"), ("resnet50", "uniform4"), ("resnet50", "bops_0.5")]
LO, HI, CALIB, SEED = 0, 16, 8, 1

if __name__ == "__main__":
    args = sys.argv[1:]
    for arch, scheme in (list(zip(args[0::2], args[1::2])) if args else CONFIGS):
        x = synthetic_images(128, seed=SEED)[LO:HI].clone()
        fx = make_golden.net_fixture(arch, scheme, HI - LO, image=x, light=True, calib=synthetic_images(CALIB, seed=0))
        fx.update(slice_lo=np.array(LO), slice_hi=np.array(HI), calib=np.array(CALIB), seed=np.array(SEED))
        out = os.path.join(HERE, f"b128live_{arch}_{scheme}.npz")
        np.savez_compressed(out, **fx)
        print(f"wrote {out}: {len(fx['conv_wpatch'])} weight patches, |acc| max {int(fx['acc_absmax'])}", flush=True)
