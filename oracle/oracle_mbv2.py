"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's frozen Q_MobileNetV2 forward
(utils/models/q_mobilenetv2.py:60-93 unit, 176-209 network; quant_utils.py:363-456 fixedpoint_fn; quant_modules.py:389-494
QuantBnConv2d, 605-736 QuantConv2d).  Never imported by hawq_amd/ or bench.py: tests and the fixture generator
tests/golden/make_b128.py only (bench.py compares its MobileNetV2 line with the fixture this file wrote).

Built on the primitives of oracle/oracle.py (exact integer convs in C, dyadic round-half-even, batch_frexp).  What differs from
the ResNet restatement is kept literal where the reference leaves the integers:

  * ReLU6 sits between a conv and its QuantAct on the fp32 tensor (q_mobilenetv2.py:66-72, 185-187): this file computes that fp32
    tensor - fl(fl(acc) * fl(S_w[c] * S_a)) - clips it at 0 and 6.0 in binary32 and lets fixedpoint_fn case 0 turn it back into
    integers, z = round(x / S_a / S_w[c]) with its two binary32 divisions (quant_utils.py:390-392).  The product path restates
    ReLU6 as ReLU + the QuantAct's clamp (hawq_amd/engine_mbv2.py); this file does NOT, which is what makes it a checker of that;
  * a unit ends without an activation: quant_act_int32 is case 1 with an identity (two requants, summed, no clamp, no ReLU,
    quant_utils.py:415-455) and case 0 without (clamped to the signed 16-bit range, :390-413);
  * the classifier is a QuantConv2d: the reference runs an fp32 conv on the un-rounded x / S_a (quant_modules.py:727-736); the
    integers it stands for are conv(x_int, W_int), which is what this file returns (float(acc) * fl(S_w[c] * S_a)) - the reference's
    own logits carry float noise of the order of an ulp around that.

PINNED: tests/test_oracle_vs_golden.py runs this file on the live reference's fixtures (tests/golden/net_mobilenetv2_w1_*_b2.npz,
written by tests/golden/make_kat_extra.py from the UNMODIFIED reference): the int32 accumulators of all 54 convs and the integers behind
every QuantAct must have the recorded digests, for three schedules.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32


def depthwise3x3(x, w, bias, stride):
    """exact integer depthwise 3x3 / pad 1 conv: x [N,C,H,W] ints, w [C,1,3,3], bias [C] -> int64 (F.conv2d(groups = C))"""
    x = np.asarray(x, np.int64)
    n, c, h, wd = x.shape
    ho, wo = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((n, c, ho, wo), np.int64) + np.asarray(bias, np.int64).reshape(1, c, 1, 1)
    for kh in range(3):
        for kw in range(3):
            out += xp[:, :, kh:kh + (ho - 1) * stride + 1:stride, kw:kw + (wo - 1) * stride + 1:stride] * np.asarray(w, np.int64)[:, 0, kh, kw].reshape(1, c, 1, 1)
    return out


def extract_float_state(q):
    """float parameters, bit widths, modes and frozen ranges of a Q_MobileNetV2 (the reference's or hawq_amd's: same attribute
    names); no integers are taken from the model"""
    act = O._to_np

    def a(m):
        return dict(bits=int(m.activation_bit), mode=str(m.quant_mode), x_min=act(m.x_min).astype(f32), x_max=act(m.x_max).astype(f32))

    def cb(m):
        c, b = m.conv, m.bn
        return dict(bits=int(m.weight_bit), w=act(c.weight), gamma=act(b.weight), beta=act(b.bias), mean=act(b.running_mean),
                    var=act(b.running_var), eps=float(b.eps), stride=int(c.stride[0]), pad=int(c.padding[0]), groups=int(c.groups))

    st = dict(quant_input=a(q.quant_input), init_block=cb(q.init_block), quant_act_int32=a(q.quant_act_int32), units=[])
    for sname, stage in q.features.named_children():
        if not sname.startswith("stage"):
            continue
        for uname, u in stage.named_children():
            d = dict(name=f"features.{sname}.{uname}", residual=bool(u.residual), quant_act=a(u.quant_act), quant_act_int32=a(u.quant_act_int32),
                     conv2=cb(u.conv2), quant_act2=a(u.quant_act2), conv3=cb(u.conv3))
            if hasattr(u, "conv1"):
                d["conv1"], d["quant_act1"] = cb(u.conv1), a(u.quant_act1)
            st["units"].append(d)
    st["quant_act_before_final_block"] = a(q.quant_act_before_final_block)
    st["final_block"] = cb(q.features.final_block)
    st["quant_act_int32_final"] = a(q.quant_act_int32_final)
    st["quant_act_output"] = a(q.quant_act_output)
    oc = q.output
    st["output"] = dict(bits=int(oc.weight_bit), w=act(oc.weight))
    return st


def _conv(tr, name, cbp, q_in, s_a, ckpt):
    """QuantBnConv2d frozen forward on integers (dense or depthwise): (acc int64, S_w, fl(S_w * S_a)); ``ckpt[name]`` may
    substitute a reference run's integer checkpoint as in oracle.forward_int"""
    w_f, b_f = O.fold_bn(cbp["w"], cbp["gamma"], cbp["beta"], cbp["mean"], cbp["var"], cbp["eps"])
    w_int, s_w = O.quantize_weight(w_f, cbp["bits"])
    b_int, bs = O.quantize_bias(b_f, s_w, s_a)
    if ckpt is not None and name in ckpt:
        ov = ckpt[name]
        s_w = np.asarray(ov["scale"], f32)
        bs = (s_w * f32(np.asarray(s_a, f32).reshape(-1)[0])).astype(f32)
        b_int = np.asarray(ov["bias"], np.int64)
        w_int = w_int.copy()
        for idx, val in ov.get("wpatch", ()):
            w_int.reshape(-1)[idx] = val
    if cbp["groups"] == 1:
        acc = O.conv2d(q_in, w_int, b_int, cbp["stride"], cbp["pad"])
    else:
        assert cbp["groups"] == w_int.shape[0] and w_int.shape[1:] == (1, 3, 3) and cbp["pad"] == 1
        acc = depthwise3x3(q_in, w_int, b_int, cbp["stride"])
    tr[name + ".weight_integer"], tr[name + ".bias_integer"], tr[name + ".convbn_scaling_factor"], tr[name + ".acc"] = w_int, b_int, s_w, acc
    return acc, s_w, bs


def _relu6_f32(acc, bs):
    """the fp32 tensor behind a conv + ReLU6: fl(fl(acc) * fl(S_w[c] * S_a)) clipped at 0 and 6.0 (quant_modules.py:491-494, nn.ReLU6)"""
    x = (acc.astype(f32) * bs.reshape(1, -1, 1, 1)).astype(f32)
    return np.minimum(np.maximum(x, f32(0)), f32(6.0))


def _relu6_then_quant_act(x, s_a, s_w, s_out, bits, mode):
    """ReLU6'd fp32 tensor -> QuantAct (fixedpoint_fn case 0) exactly as the reference runs it"""
    z = ((x / f32(np.asarray(s_a, f32).reshape(-1)[0])).astype(f32) / np.asarray(s_w, f32).reshape(1, -1, 1, 1)).astype(f32)
    z_int = np.rint(z.astype(np.float64)).astype(np.int64)                            # quant_utils.py:392
    m, e = O.requant_table(s_a, s_w, s_out)
    return O.dyadic(z_int, m, e, O.act_range(bits, mode))


def forward_int(st, x, ckpt=None, calibrate: bool = False):
    """integer forward of a frozen Q_MobileNetV2; returns (logits fp32 [N, classes], Trace keyed by the reference's module names).
    With ``calibrate`` every QuantAct range is (re)initialised from this batch exactly as ONE un-frozen reference forward does
    (quant_modules.py:233-250: the first call sets x_min / x_max to the min / max of the fp32 tensor it is given) and written back
    into ``st`` - what hawq_amd.api.calibrate does to the product's model."""
    tr = O.Trace()
    one = np.ones(1, f32)

    def sc(a, xf=None):
        if calibrate:
            a["x_min"], a["x_max"] = np.asarray([xf.min()], f32), np.asarray([xf.max()], f32)
        return O.act_scale(a["x_min"], a["x_max"], a["bits"], a["mode"])

    f_of = lambda r, s: (r.astype(f32) * f32(s[0])).astype(f32)    # the fp32 tensor integers r at scale s stand for

    x = np.ascontiguousarray(x, f32)
    a = st["quant_input"]
    s_in = sc(a, x)
    q = O.quantize_f32(x, s_in[0], a["bits"], a["mode"])
    tr["quant_input.q"] = q
    acc, s_w, bs = _conv(tr, "init_block", st["init_block"], q, s_in, ckpt)
    a = st["quant_act_int32"]
    x6 = _relu6_f32(acc, bs)
    s16 = sc(a, x6)
    r = _relu6_then_quant_act(x6, s_in, s_w, s16, a["bits"], a["mode"])
    tr["quant_act_int32.q"] = r
    s_prev = s16
    for u in st["units"]:
        n = u["name"]
        a = u["quant_act"]
        r_f = f_of(r, s_prev) if calibrate else None
        s_a = sc(a, r_f)
        m, e = O.requant_table(s_prev, one, s_a)
        qa = O.dyadic(r, m, e, O.act_range(a["bits"], a["mode"]))
        tr[n + ".quant_act.q"] = qa
        xq, s_x = qa, s_a
        for conv, act in (("conv1", "quant_act1"), ("conv2", "quant_act2")):
            if conv not in u:
                continue
            acc, s_w, bs = _conv(tr, f"{n}.{conv}", u[conv], xq, s_x, ckpt)
            a = u[act]
            x6 = _relu6_f32(acc, bs)
            s_n = sc(a, x6)
            xq = _relu6_then_quant_act(x6, s_x, s_w, s_n, a["bits"], a["mode"])
            tr[f"{n}.{act}.q"] = xq
            s_x = s_n
        acc3, s_w3, bs3 = _conv(tr, n + ".conv3", u["conv3"], xq, s_x, ckpt)   # no activation after the projection
        a = u["quant_act_int32"]
        z_f = None
        if calibrate:   # the tensor quant_act_int32 sees: conv3's fp32 output (+ the identity's, q_mobilenetv2.py:84-88)
            z_f = (acc3.astype(f32) * bs3.reshape(1, -1, 1, 1)).astype(f32)
            if u["residual"]:
                z_f = (z_f + r_f).astype(f32)
        s_o = sc(a, z_f)
        m2, e2 = O.requant_table(s_x, s_w3, s_o)
        if u["residual"]:   # case 1: identity = the block input, still at the previous unit's 16-bit scale; no clamp
            m1, e1 = O.requant_table(s_prev, one, s_o)
            r = O.dyadic(r, m1, e1) + O.dyadic(acc3, m2, e2)
        else:               # case 0: clamped to the signed 16-bit range
            r = O.dyadic(acc3, m2, e2, O.act_range(a["bits"], a["mode"]))
        tr[n + ".quant_act_int32.q"] = r
        s_prev = s_o
    a = st["quant_act_before_final_block"]
    s_b = sc(a, f_of(r, s_prev) if calibrate else None)
    m, e = O.requant_table(s_prev, one, s_b)
    qb = O.dyadic(r, m, e, O.act_range(a["bits"], a["mode"]))
    tr["quant_act_before_final_block.q"] = qb
    acc, s_w, bs = _conv(tr, "features.final_block", st["final_block"], qb, s_b, ckpt)
    a = st["quant_act_int32_final"]
    x6 = _relu6_f32(acc, bs)
    s_f = sc(a, x6)
    r = _relu6_then_quant_act(x6, s_b, s_w, s_f, a["bits"], a["mode"])
    tr["quant_act_int32_final.q"] = r
    pooled = O.avgpool_trunc(r)
    tr["features.final_pool.q"] = pooled
    a = st["quant_act_output"]
    s_8 = sc(a, f_of(pooled, s_f) if calibrate else None)
    m, e = O.requant_table(s_f, one, s_8)
    qf = O.dyadic(pooled, m, e, O.act_range(a["bits"], a["mode"]))
    tr["quant_act_output.q"] = qf
    oc = st["output"]
    w = np.asarray(oc["w"], f32)
    w_int, s_oc = O.quantize_weight(w.reshape(w.shape[0], -1), oc["bits"])
    if ckpt is not None and "output" in ckpt:
        s_oc = np.asarray(ckpt["output"]["scale"], f32)
        w_int = w_int.copy()
        for idx, val in ckpt["output"].get("wpatch", ()):
            w_int.reshape(-1)[idx] = val
    acc = O.linear(qf, w_int, np.zeros(w_int.shape[0], np.int64))
    tr["output.weight_integer"], tr["output.conv_scaling_factor"], tr["output.acc"] = w_int, s_oc, acc
    logits = (acc.astype(f32) * (s_oc * f32(s_8[0])).astype(f32).reshape(1, -1)).astype(f32)   # quant_modules.py:735
    tr["logits"] = logits
    return logits, tr
