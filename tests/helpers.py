"""Shared test helpers: golden fixture access, digests, model construction."""
from __future__ import annotations

import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NET_CONFIGS = [("resnet18", "uniform8"), ("resnet18", "uniform4"), ("resnet18", "bops_0.5"),
               ("resnet50", "uniform8"), ("resnet50", "uniform4"), ("resnet50", "bops_0.5")]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def net_fixture(arch, scheme, batch=2):
    return load(f"net_{arch}_{scheme}_b{batch}.npz")


def digest(a) -> np.ndarray:
    """Same 3-word digest as tests/golden/make_golden.py (NCHW order)."""
    a = np.ascontiguousarray(a).astype(np.int64).reshape(-1)
    w = (np.arange(a.size, dtype=np.int64) % 8191) + 1
    with np.errstate(over="ignore"):
        return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.int64)


def sha(t) -> str:
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def oracle_name(conv_name: str) -> str:
    return "stem" if conv_name.startswith("quant_init") else conv_name


def reference_ckpt(fx, st):
    """Integer checkpoint of the reference run stored in a net fixture -> oracle ``ckpt`` dict."""
    names = [str(n) for n in fx["conv_names"]]
    cbs = {"stem": st["stem"]}
    for u in st["units"]:
        for km, ks in (("quant_convbn1", "convbn1"), ("quant_convbn2", "convbn2"), ("quant_convbn3", "convbn3"),
                       ("quant_identity_convbn", "identity")):
            if ks in u:
                cbs[u["name"] + "." + km] = u[ks]
    ck, off = {}, 0
    for li, n in enumerate(names):
        on = oracle_name(n)
        co = cbs[on]["w"].shape[0]
        patch = [(int(i), int(v)) for l, i, v in fx["conv_wpatch"] if l == li]
        ck[on] = dict(scale=fx["conv_scale"][off:off + co], bias=fx["conv_bias"][off:off + co], wpatch=patch)
        off += co
    return ck
