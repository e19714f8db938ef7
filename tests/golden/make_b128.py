"""Oracle logits of the BENCHMARKED workloads at their own batch size (BASELINE.json configs[1..4]).

    python tests/golden/make_b128.py [arch scheme ...]      # writes tests/golden/b128_<arch>_<scheme>.npz

`bench.py:setup_workload` builds: synthetic weights (seed 0), ranges calibrated on synthetic_images(8, seed 0),
input synthetic_images(128, seed 1).  This script pushes exactly that workload through the CPU oracle
(oracle/oracle.py + oracle/hawq_oracle.c - the restatement pinned to the live reference by
tests/test_oracle_vs_golden.py) in slices of 16 images and stores all 128 x 1000 logits, top-1, and a
SHA-256 of every unit's un-clamped 16-bit residual tensor (NCHW, int32) per slice.

Test infrastructure: the fixtures are consumed by tests/test_gpu_b128.py and by bench.py's parity check
(`gpu_logits_bit_equal`); hawq_amd/ never reads them.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from hawq_amd.api import build_quantized_resnet  # noqa: E402
from hawq_amd.skeleton import synthetic_images  # noqa: E402
from oracle import oracle, oracle_mbv2  # noqa: E402

CONFIGS = [("resnet18", "uniform8"), ("resnet50", "uniform8"), ("resnet50", "uniform4"), ("resnet50", "bops_0.5")]
BATCH, SLICE, CALIB, SEED = 128, 16, 8, 1


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_mobilenet(arch, scheme):
    """the same for bench.py's MobileNetV2 line (bench.mobilenet_line): oracle/oracle_mbv2.py calibrates itself on the 8 calibration
    images as one un-frozen forward does, then runs the 128 benchmarked images; logits + a SHA-256 of every unit's 16-bit output"""
    model = build_quantized_resnet(arch, scheme, seed=0)
    st = oracle_mbv2.extract_float_state(model)
    oracle_mbv2.forward_int(st, synthetic_images(CALIB, seed=0).numpy(), calibrate=True)
    x = synthetic_images(BATCH, seed=SEED).numpy()
    logits, res_sha, names = [], [], None
    for b0 in range(0, BATCH, SLICE):
        t0 = time.time()
        y, tr = oracle_mbv2.forward_int(st, x[b0:b0 + SLICE])
        logits.append(y)
        names = [k for k in tr.keys() if k.endswith(".quant_act_int32.q")]
        res_sha.append([sha(tr[k].astype(np.int32)) for k in names])
        print(f"{arch} {scheme}: images {b0}..{b0 + SLICE - 1} in {time.time() - t0:.1f} s", flush=True)
    logits = np.concatenate(logits).astype(np.float32)
    out = os.path.join(HERE, f"b128_{arch}_{scheme}.npz")
    np.savez_compressed(out, logits=logits, top1=logits.argmax(1).astype(np.int64), input_sha=np.array(sha(x)),
                        residual_names=np.array(names), residual_sha=np.array(res_sha), slice=np.array(SLICE), calib=np.array(CALIB),
                        seed=np.array(SEED))
    print(f"wrote {out}", flush=True)


def make(arch, scheme):
    if arch.startswith("mobilenet"):
        return make_mobilenet(arch, scheme)
    model = build_quantized_resnet(arch, scheme, seed=0)
    st = oracle.extract_float_state(model)
    oracle.forward_int(st, synthetic_images(CALIB, seed=0).numpy(), calibrate=True)
    x = synthetic_images(BATCH, seed=SEED).numpy()
    logits, res_sha, res_max = [], [], 0
    names = None
    for b0 in range(0, BATCH, SLICE):
        t0 = time.time()
        y, tr = oracle.forward_int(st, x[b0:b0 + SLICE])
        logits.append(y)
        keys = [k for k in tr.keys() if k.endswith(".quant_act_int32.q")]
        names = keys
        res_sha.append([sha(tr[k].astype(np.int32)) for k in keys])
        res_max = max(res_max, max(int(tr[k].max()) for k in keys))
        print(f"{arch} {scheme}: images {b0}..{b0 + SLICE - 1} in {time.time() - t0:.1f} s", flush=True)
    logits = np.concatenate(logits).astype(np.float32)
    out = os.path.join(HERE, f"b128_{arch}_{scheme}.npz")
    np.savez_compressed(out, logits=logits, top1=logits.argmax(1).astype(np.int64), input_sha=np.array(sha(x)),
                        residual_names=np.array(names), residual_sha=np.array(res_sha), residual_max=np.array(res_max),
                        slice=np.array(SLICE), calib=np.array(CALIB), seed=np.array(SEED))
    print(f"wrote {out}: residual max {res_max}", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    cfgs = list(zip(args[0::2], args[1::2])) if args else CONFIGS
    for a, s in cfgs:
        make(a, s)
