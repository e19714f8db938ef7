"""Shared test helpers: golden fixture access, digests, model construction."""
from __future__ import annotations

import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NET_CONFIGS = [("resnet18", "uniform8"), ("resnet18", "uniform4"), ("resnet18", "bops_0.5"),
               ("resnet50", "uniform8"), ("resnet50", "uniform4"), ("resnet50", "bops_0.5")]
# "light" fixtures (no full accumulator tensors): further shipped graphs / schedules
NET_CONFIGS_EXTRA = [("resnet101", "uniform8"), ("resnet50b", "uniform4"), ("resnet50", "latency_0.5"),
                     ("resnet50", "modelsize_0.25")]


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


# ------------------------------------------------------------------ GPU-side helpers
def build_model(arch, scheme, device="cuda"):
    from hawq_amd.api import build_quantized_resnet
    return build_quantized_resnet(arch, scheme, seed=0).to(device)


def conv_modules(model):
    from hawq_amd.quant_modules import QuantBnConv2d
    return [(n, m) for n, m in model.named_modules() if isinstance(m, QuantBnConv2d)]


def act_modules(model):
    from hawq_amd.quant_modules import QuantAct
    return [(n, m) for n, m in model.named_modules() if isinstance(m, QuantAct)]


def load_reference_ranges(model, fx):
    """Set every QuantAct's frozen range from a net fixture (as a QAT checkpoint would)."""
    import torch
    for i, (n, m) in enumerate(act_modules(model)):
        assert n == str(fx["act_names"][i])
        m.x_min.fill_(float(fx["act_x_min"][i]))
        m.x_max.fill_(float(fx["act_x_max"][i]))
        m.compute_scale()
        assert m.act_scaling_factor.item() == float(fx["act_scale"][i]), n


def load_reference_integer_ckpt(model, fx):
    """Overwrite the conv modules' integer buffers with the reference run's (scales and biases in
    full, weight_integer = IEEE preparation + the recorded patches).  Returns #patched weights."""
    import numpy as np
    import torch
    off, npatch = 0, 0
    for li, (n, m) in enumerate(conv_modules(model)):
        assert n == str(fx["conv_names"][li])
        co = m.out_channels
        dev = m.weight_integer.device
        w = m.weight_integer.detach().cpu().numpy().copy()
        for l, idx, val in fx["conv_wpatch"]:
            if l == li:
                w.reshape(-1)[idx] = val
                npatch += 1
        assert np.array_equal(digest(w), fx["conv_wdigest"][li]), n
        m.weight_integer = torch.from_numpy(w).to(dev)
        m.convbn_scaling_factor = torch.from_numpy(fx["conv_scale"][off:off + co].copy()).to(dev)
        m.bias_integer = torch.from_numpy(fx["conv_bias"][off:off + co].astype(np.float32)).to(dev)
        off += co
    # the classifier: its bias integers depend on the (reference's) scale of quant_act_output
    fc = model.quant_output
    dev = fc.weight_integer.device
    assert np.array_equal(digest(fc.weight_integer.detach().cpu().numpy()), fx["fc_wdigest"])
    fc.fc_scaling_factor = torch.from_numpy(np.asarray(fx["fc_scale"], np.float32).reshape(-1).copy()).to(dev)
    fc.bias_integer = torch.from_numpy(np.asarray(fx["fc_bias"]).astype(np.float32).reshape(-1)).to(dev)
    return npatch
