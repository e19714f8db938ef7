"""TEST/BENCH INFRASTRUCTURE ONLY - CPU port of the reference's *fake-quant* forward, for timing.

BASELINE.json asks for "the repo's own CPU fake-quant path timed on the host cores of the same
box".  The reference itself (Python under /root/reference) cannot travel to the GPU box, so
this module restates its frozen forward WITH ITS COST STRUCTURE: fp32 tensors holding
integer*scale, per-forward BN folding and weight re-quantisation (quant_modules.py:441-484),
fp32 ``F.conv2d`` on ``x / S_a`` (:489-494), float64 emulation of the dyadic requantisation with
a per-channel host ``Decimal`` loop (quant_utils.py:188-213, 363-456), fp32 add/ReLU/max-pool
(q_resnet.py:114-135, 231-260).  It is written functionally over the oracle's float-state dict
(oracle.extract_float_state) rather than as nn.Modules.

bench.py's ``cpu_baseline`` leg times this ("kind": "port"); tests pin its logits to the
live-reference goldens.  The product path never imports it.
"""
from __future__ import annotations

import decimal
from decimal import Decimal

import numpy as np
import torch
import torch.nn.functional as F


_DEV = None   # None: CPU (the timed baseline); a torch.device: bench.py's secondary "port on the GPU" figure


def _t(a):
    t = torch.from_numpy(a)
    return t if _DEV is None else t.to(_DEV)


def _frexp(r):
    dev = r.device
    r = r.cpu()   # the reference does this D2H -> Decimal -> H2D round trip on every call too (quant_utils.py:202-213)
    m, e = np.frexp(r.reshape(-1).numpy())
    mm = np.array([int(Decimal(float(v) * (2 ** 31)).quantize(Decimal('1'), rounding=decimal.ROUND_HALF_UP))
                   for v in m])
    return torch.from_numpy(mm).view(r.shape).to(dev), torch.from_numpy(31. - e).view(r.shape).to(dev)


def _sym_scale(lo, hi, bits, per_channel):
    n = 2 ** (bits - 1) - 1
    s = torch.maximum(lo.abs(), hi.abs())
    return torch.clamp(s, min=1e-8) / n


def _act_scale(a):
    lo, hi = _t(a["x_min"]), _t(a["x_max"])
    if a["mode"] == "symmetric":
        return _sym_scale(lo, hi, a["bits"], False)
    return torch.clamp(hi - lo, min=1e-8) / float(2 ** a["bits"] - 1)


def _quant(x, bits, scale):
    n = 2 ** (bits - 1) - 1
    shape = (-1,) + (1,) * (x.dim() - 1) if scale.numel() > 1 and x.dim() > 1 else (-1,)
    return torch.clamp(torch.round(1. / scale.view(shape) * x), -n - 1, n)


def _convbn(cb, x, s_a):
    """QuantBnConv2d frozen forward incl. the per-forward parameter work."""
    w, g, b = _t(cb["w"]), _t(cb["gamma"]), _t(cb["beta"])
    mean, var = _t(cb["mean"]), _t(cb["var"])
    std = torch.sqrt((var + cb["eps"]).double()).float()  # correctly rounded (DESIGN.md "sqrt quirk")
    sf = g / std
    sw = w * sf.reshape(-1, 1, 1, 1)
    sb = (torch.zeros_like(mean) - mean) * sf + b
    flat = sw.contiguous().view(sw.shape[0], -1)
    s_w = _sym_scale(flat.min(dim=1).values, flat.max(dim=1).values, cb["bits"], True)
    w_int = _quant(sw, cb["bits"], s_w)
    bs = s_w.view(1, -1) * s_a.view(1, -1)
    b_int = _quant(sb, 32, bs)
    y = F.conv2d(x / s_a.view(1, -1, 1, 1), w_int, b_int, cb["stride"], cb["pad"]) * bs.view(1, -1, 1, 1)
    return y, s_w


def _fixedpoint(z, a, s_out, s_a, s_w, identity=None, s_ida=None, s_idw=None):
    """fixedpoint_fn.forward (float64 emulation), returns integer-valued fp32."""
    v = lambda t: t.view(1, -1, 1, 1) if z.dim() == 4 else t.view(1, -1)

    def branch(t, sa, sw):
        t_int = torch.round(t / v(sa) / v(sw))
        r = (v(sa).double() * v(sw).double()).float().double() / v(s_out).float().double()
        m, e = _frexp(r)
        return torch.round(t_int.double() * m.double() / (2.0 ** e))

    if identity is None:
        out = branch(z, s_a, s_w).float()
        if a["mode"] == "symmetric":
            n = 2 ** (a["bits"] - 1) - 1
            return torch.clamp(out, -n - 1, n)
        return torch.clamp(out, 0, 2 ** a["bits"] - 1)
    return (branch(identity, s_ida, s_idw) + branch(z - identity, s_a, s_w)).float()


def forward(st, x: torch.Tensor, dev=None) -> torch.Tensor:
    """Frozen fake-quant forward (torch fp32) on CPU - or, ``dev`` a GPU, with every tensor on that device (the
    reference's own `.cuda()` path; bench.py's secondary baseline); x fp32 [N,3,H,W]; returns fp32 logits."""
    global _DEV
    _DEV = dev
    try:
        return _forward(st, x if dev is None else x.to(dev))
    finally:
        _DEV = None


def _forward(st, x):
    one = torch.ones(1, device=x.device)
    with torch.no_grad():
        a = st["quant_input"]
        s = _act_scale(a)
        x = _quant(x, a["bits"], s) * s
        x, s_w = _convbn(st["stem"], x, s)
        x = F.max_pool2d(x, 3, 2, 1)
        a = st["quant_act_int32"]
        s0 = _act_scale(a)
        x = torch.relu(_fixedpoint(x, a, s0, s, s_w) * s0)
        s_prev = s0
        for u in st["units"]:
            a = u["quant_act"]
            s_a = _act_scale(a)
            if u["resize"]:
                x = _fixedpoint(x, a, s_a, s_prev, one) * s_a
                identity, s_idw = _convbn(u["identity"], x, s_a)
                s_ida = s_a
            else:
                identity, s_idw, s_ida = x, one, s_prev
                x = _fixedpoint(x, a, s_a, s_prev, one) * s_a
            s_x = s_a
            keys = ("convbn1", "convbn2", "convbn3") if "convbn3" in u else ("convbn1", "convbn2")
            for i, k in enumerate(keys):
                x, s_w = _convbn(u[k], x, s_x)
                if i < len(keys) - 1:
                    a = u[f"quant_act{i + 1}"]
                    s_n = _act_scale(a)
                    x = _fixedpoint(torch.relu(x), a, s_n, s_x, s_w) * s_n
                    s_x = s_n
            x = x + identity
            a = u["quant_act_int32"]
            s_o = _act_scale(a)
            x = torch.relu(_fixedpoint(x, a, s_o, s_x, s_w, identity, s_ida, s_idw) * s_o)
            s_prev = s_o
        x_int = F.avg_pool2d(torch.round(x / s_prev), 7, 1)
        x = torch.trunc(x_int + 0.01) * s_prev
        a = st["quant_act_output"]
        s8 = _act_scale(a)
        x = (_fixedpoint(x, a, s8, s_prev, one) * s8).view(x.size(0), -1)
        fc = st["fc"]
        w, b = _t(fc["w"]), _t(fc["b"])
        s_fc = _sym_scale(w.min(dim=1).values, w.max(dim=1).values, fc["bits"], True)
        w_int = _quant(w, fc["bits"], s_fc)
        bs = s_fc.view(1, -1) * s8.view(1, -1)
        b_int = _quant(b, 32, bs)
        return torch.round(F.linear(x / s8.view(1, -1), w_int, b_int)) * bs[0].view(1, -1)
