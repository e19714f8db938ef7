"""Known-answer vectors for the (f)-row widenings, recorded from the LIVE reference (build container only):
QuantConv2d incl. grouped / depthwise convolutions and percentile weight ranges (quant_modules.py:605-736),
QuantBnConv2d depthwise (MobileNetV2's 3x3), get_percentile_min_max (quant_utils.py:38-70) and the un-frozen
QuantAct's range tracking with and without percentiles (quant_modules.py:233-258).

    python tests/golden/make_kat_extra.py        # writes tests/golden/kat_extra.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_live  # noqa: E402

CONV_CASES = {  # tag: (cin, cout, k, stride, pad, groups, bias, weight_bit, per_channel, weight_percentile, hw, act range)
    "c3": (16, 8, 3, 1, 1, 1, True, 8, True, 0, 9, (-128, 127)),
    "c1nb": (32, 16, 1, 1, 0, 1, False, 8, True, 0, 6, (-128, 127)),
    "g4": (32, 16, 3, 1, 1, 4, True, 8, True, 0, 7, (-128, 127)),
    "dw": (32, 32, 3, 1, 1, 32, False, 8, True, 0, 10, (0, 127)),
    "dws2": (48, 48, 3, 2, 1, 48, True, 8, True, 0, 11, (0, 127)),
    "dw4": (32, 32, 3, 1, 1, 32, False, 4, True, 0, 8, (0, 15)),
    "pct": (16, 8, 3, 1, 1, 1, True, 8, True, 99, 9, (-128, 127)),
    "pct_t": (16, 8, 3, 1, 1, 1, True, 8, False, 99.5, 9, (-128, 127)),
}


def main():
    qr, qm, qu = ref_live.load_reference()
    g = torch.Generator().manual_seed(23)
    out = {}
    for tag, (cin, cout, k, stride, pad, groups, bias, wbit, pc, pct, hw, (lo, hi)) in CONV_CASES.items():
        conv = torch.nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=bias)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
            if bias:
                conv.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        m = qm.QuantConv2d(weight_bit=wbit, bias_bit=32 if bias else None, per_channel=pc, weight_percentile=pct)
        m.set_param(conv)
        s_a = torch.tensor([0.0173])
        q = torch.randint(lo, hi + 1, (2, cin, hw, hw), generator=g).float()
        with torch.no_grad():
            y, s_w = m(q * s_a, s_a)
        rec = dict(w=conv.weight, q=q, s_a=s_a, y=y, s_w=s_w, weight_integer=m.weight_integer.float())
        if bias:
            rec.update(b=conv.bias, bias_integer=m.bias_integer)
        for kx, v in rec.items():
            out[f"qconv_{tag}_{kx}"] = v.detach().numpy()
        out[f"qconv_{tag}_cfg"] = np.array([cin, cout, k, stride, pad, groups, int(bias), wbit, int(pc), hw], np.int64)
        out[f"qconv_{tag}_pct"] = np.array([pct], np.float64)
    # QuantBnConv2d with a depthwise conv (q_mobilenetv2.py's 3x3) and a percentile weight range
    for tag, (c, stride, pct) in {"bndw": (32, 1, 0), "bndws2": (48, 2, 0), "bnpct": (32, 1, 99)}.items():
        conv = torch.nn.Conv2d(c, c, 3, stride, 1, groups=c if tag != "bnpct" else 1, bias=False)
        bn = torch.nn.BatchNorm2d(c)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.1)
        m = qm.QuantBnConv2d(weight_bit=8, bias_bit=32, per_channel=True, fix_BN=True, weight_percentile=pct)
        m.set_param(conv, bn)
        m.fix()
        m.eval()
        s_a = torch.tensor([0.021])
        q = torch.randint(0, 128, (2, c, 9, 9), generator=g).float()
        with torch.no_grad():
            y, s_w = m((q * s_a, s_a))
        for kx, v in dict(w=conv.weight, gamma=bn.weight, beta=bn.bias, mean=bn.running_mean, var=bn.running_var, q=q, s_a=s_a, y=y,
                          s_w=s_w, weight_integer=m.weight_integer, bias_integer=m.bias_integer).items():
            out[f"{tag}_{kx}"] = v.detach().numpy()
        out[f"{tag}_cfg"] = np.array([c, stride, pct], np.float64)
    # get_percentile_min_max
    for i, (n, lowp, upp) in enumerate([(1000, 0.1, 99.9), (4097, 1, 99), (50, 10, 90), (100000, 0.01, 99.99), (777, 0, 99.9)]):
        x = torch.randn(n, generator=g) * 3 + 0.5
        if i == 2:
            x[::5] = 0.0
        lo, hi = qu.get_percentile_min_max(x, lowp, upp, output_tensor=True)
        out[f"pct{i}_x"], out[f"pct{i}_cfg"] = x.numpy(), np.array([lowp, upp], np.float64)
        out[f"pct{i}_out"] = np.array([float(lo), float(hi)], np.float32)
    # un-frozen QuantAct: range tracking over three forwards (initialisation, then two momentum updates)
    for tag, (bits, mode, pct, mom) in {"mm": (8, "symmetric", 0, 0.99), "mmx": (8, "symmetric", 0, -1), "ps": (8, "symmetric", 99.9, 0.95),
                                         "pa": (4, "asymmetric", 99.0, 0.9)}.items():
        a = qm.QuantAct(activation_bit=bits, act_range_momentum=mom, quant_mode=mode, act_percentile=pct)
        xs, rng, ys = [], [], []
        for it in range(3):
            x = torch.randn(2, 6, 9, 9, generator=g) * (1.0 + it)
            if mode == "asymmetric":
                x = torch.relu(x)
            with torch.no_grad():
                y, s = a(x)
            xs.append(x.numpy()), ys.append(y.numpy())
            rng.append([float(a.x_min), float(a.x_max), float(s)])
        out[f"act_{tag}_x"], out[f"act_{tag}_y"] = np.stack(xs), np.stack(ys)
        out[f"act_{tag}_rng"] = np.array(rng, np.float32)
        out[f"act_{tag}_cfg"] = np.array([bits, pct, mom], np.float64)
    np.savez_compressed(os.path.join(HERE, "kat_extra.npz"), **out)
    print("wrote kat_extra.npz with", len(out), "arrays")


if __name__ == "__main__" and "--mobilenet" not in sys.argv:
    main()


def mobilenet_fixture(scheme, batch=2):
    """Whole-network fixture of the LIVE reference's Q_MobileNetV2 (q_mobilenetv2.py): frozen ranges, integer buffers
    (scales + biases in full; weight_integer as SHA-256 per layer plus the entries where torch-CPU's non-IEEE sqrt moved
    a weight relative to hawq_amd's IEEE preparation), logits."""
    import hashlib
    from hawq_amd.api import build_quantized_model
    from hawq_amd.skeleton import synthetic_images
    qr, qm, qu = ref_live.load_reference()
    q = ref_live.build_reference_model("mobilenetv2_w1", scheme, seed=0)
    x = synthetic_images(batch, seed=0)
    ref_live.calibrate_and_freeze(q, x)
    # one frozen forward with taps: raw F.conv2d outputs (= int32 accumulators incl. bias) and every QuantAct's integer output
    act_out = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: act_out.__setitem__(n, torch.round(o[0] / o[1].reshape(-1)[0]).to(torch.int64)))
             for n, m in q.named_modules() if type(m).__name__ == "QuantAct"]
    y, conv_taps, _ = ref_live.forward_with_taps(q, x)
    for hk in hooks:
        hk.remove()
    ours = build_quantized_model("mobilenetv2_w1", scheme, seed=0)
    out = dict(logits=y.numpy(), top1=y.argmax(1).numpy(), input_sha=np.array(hashlib.sha256(x.numpy().tobytes()).hexdigest()))
    acts = [(n, m) for n, m in q.named_modules() if type(m).__name__ == "QuantAct"]
    out["act_names"] = np.array([n for n, _ in acts])
    out["act_x_min"] = np.array([float(m.x_min) for _, m in acts], np.float32)
    out["act_x_max"] = np.array([float(m.x_max) for _, m in acts], np.float32)
    out["act_scale"] = np.array([float(m.act_scaling_factor.reshape(-1)[0]) for _, m in acts], np.float32)
    convs = [(n, m) for n, m in q.named_modules() if type(m).__name__ in ("QuantBnConv2d", "QuantConv2d")]
    mine = dict(ours.named_modules())
    names, scales, biases, shas, patches = [], [], [], [], []
    s_prev = {}
    for li, (n, m) in enumerate(convs):
        names.append(n)
        sc = (m.convbn_scaling_factor if type(m).__name__ == "QuantBnConv2d" else m.conv_scaling_factor).detach().reshape(-1)
        scales.append(sc.numpy().astype(np.float32))
        b = m.bias_integer
        biases.append(np.zeros(sc.numel(), np.float64) if b is None else b.detach().reshape(-1).numpy().astype(np.float64))
        w_ref = m.weight_integer.detach().float().numpy()
        shas.append(hashlib.sha256(np.ascontiguousarray(w_ref.astype(np.int8)).tobytes()).hexdigest())
        mm = mine[n]
        if type(m).__name__ == "QuantBnConv2d":   # hawq_amd's own (IEEE) preparation, on the host
            mm.prepare(torch.ones(1))
        else:
            from hawq_amd.quant_utils import quantize_weight_per_channel
            mm.weight_integer = quantize_weight_per_channel(mm.weight, mm.weight_bit, mm.per_channel, mm.weight_percentile)[0]
        w_own = mm.weight_integer.detach().float().numpy()
        for idx in np.nonzero(w_own.reshape(-1) != w_ref.reshape(-1))[0]:
            patches.append((li, int(idx), int(w_ref.reshape(-1)[idx])))
    assert len(conv_taps) == len(convs)   # call order == registration order in this graph
    sys.path.insert(0, HERE)
    from make_golden import digest
    out.update(conv_names=np.array(names), conv_scale=np.concatenate(scales), conv_bias=np.concatenate(biases),
               conv_wsha=np.array(shas), conv_wpatch=np.array(patches, np.int64).reshape(-1, 3),
               # 3-word digests (make_golden.digest, NCHW order) of rint(raw conv output) per conv and of every QuantAct's integers
               conv_accdigest=np.stack([digest(np.rint(t.numpy().astype(np.float64)).astype(np.int64)) for t in conv_taps]),
               act_outdigest=np.stack([digest(act_out[n].numpy()) for n, _ in acts]),
               act_outmax=np.array([int(act_out[n].abs().max()) for n, _ in acts], np.int64))
    np.savez_compressed(os.path.join(HERE, f"net_mobilenetv2_w1_{scheme}_b{batch}.npz"), **out)
    print("mobilenetv2_w1", scheme, "top1", out["top1"], "weight patches", len(patches), flush=True)


if __name__ == "__main__" and "--mobilenet" in sys.argv:
    for scheme in ("uniform8", "uniform4", "bops_0.5"):
        mobilenet_fixture(scheme)
