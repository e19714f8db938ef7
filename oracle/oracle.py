"""TEST INFRASTRUCTURE ONLY - CPU oracle for the frozen integer forward of HAWQ's quantized ResNets.

numpy + oracle/hawq_oracle.c (pure C, exact integers).  Only tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg may import this package; the product path
(hawq_amd/) never does and fails loudly without its HIP library.

What is restated, and from where (paths relative to the HAWQ tree):
  * host-side parameter preparation (BN folding, per-channel weight scales, weight/bias
    integers, activation scales):  quant_modules.py:441-484, 97-118, 262-270;
    quant_utils.py:73-97, 128-185
  * dyadic requantisation tables:  quant_utils.py:188-213, 394-404, 419-449
  * the integer graph:  q_resnet.py:53-74, 114-135, 231-260, 291-316
  * calibration ranges (one un-frozen forward):  quant_modules.py:233-258

Pinned by tests/test_oracle_vs_golden.py against fixtures generated from the live
reference (tests/golden/make_golden.py) - see DESIGN.md "Oracle".
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hawq_oracle.c")
_SO = os.path.join(_HERE, "_build", "libhawq_oracle.so")
_lib = None

f32 = np.float32
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
L = C.c_int64


def build(force: bool = False) -> str:
    """gcc -O2 -fopenmp oracle/hawq_oracle.c -> oracle/_build/libhawq_oracle.so"""
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC",
             _SRC, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        l.hq_num_threads.restype = C.c_int
        l.hq_frexp_me.argtypes = [f64p, L, i64p, i32p]
        l.hq_dyadic_nchw.argtypes = [i64p, L, L, L, i64p, i32p, L, C.c_int, L, L, i64p]
        l.hq_quantize_f32.argtypes = [f32p, L, C.c_float, C.c_float, C.c_float, i64p]
        l.hq_conv2d_nchw.argtypes = [i16p, L, L, L, L, i8p, C.c_void_p, L, L, L, L, L, i64p]
        l.hq_linear.argtypes = [i16p, L, L, i8p, C.c_void_p, L, i64p]
        l.hq_maxpool_nchw.argtypes = [i64p, L, L, L, L, L, L, i64p]
        l.hq_avgpool_trunc.argtypes = [i64p, L, L, i64p]
        _lib = l
    return _lib


def num_threads() -> int:
    return lib().hq_num_threads()


# ----------------------------------------------------------------------------- primitives
def frexp_me(r):
    """batch_frexp (quant_utils.py:188-213): r -> (m = HALF_UP(mant*2^31), e = 31 - exp)."""
    r = np.ascontiguousarray(np.asarray(r, dtype=np.float64).reshape(-1))
    m = np.empty(r.size, np.int64)
    e = np.empty(r.size, np.int32)
    lib().hq_frexp_me(r, r.size, m, e)
    return m, e


def requant_table(s_a, s_w, s_out):
    """(m, e) of fixedpoint_fn (quant_utils.py:394-404): r = dbl(fl(S_a*S_w)) / dbl(fl(S_out))."""
    a = np.asarray(s_a, f32).reshape(-1)
    w = np.asarray(s_w, f32).reshape(-1)
    prod = (a * w).astype(f32)  # the binary64 product of two binary32 is exact => one rounding
    r = prod.astype(np.float64) / np.float64(f32(np.asarray(s_out, f32).reshape(-1)[0]))
    return frexp_me(r)


def dyadic(acc, m, e, clamp=None):
    """round_half_even(acc*m / 2^e) per channel (dim 1) on [N,C,...] int64; optional clamp."""
    acc = np.ascontiguousarray(acc, dtype=np.int64)
    shp = acc.shape
    n, c = shp[0], shp[1]
    hw = int(np.prod(shp[2:])) if len(shp) > 2 else 1
    out = np.empty_like(acc)
    m = np.ascontiguousarray(m, np.int64)
    e = np.ascontiguousarray(e, np.int32)
    lo, hi = (clamp if clamp is not None else (0, 0))
    lib().hq_dyadic_nchw(acc, n, c, hw, m, e, m.size, int(clamp is not None), int(lo), int(hi), out)
    return out


def act_range(bits: int, mode: str):
    """Integer clamp range (quant_utils.py:255, 304, 366-369, 410-413)."""
    if mode == "symmetric":
        return -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    return 0, 2 ** bits - 1


def quantize_f32(x, scale, bits, mode="symmetric"):
    """Symmetric/AsymmetricQuantFunction.forward on fp32 data with a scalar scale."""
    x = np.ascontiguousarray(x, f32)
    inv = f32(1.0) / f32(scale)
    lo, hi = act_range(bits, mode)
    q = np.empty(x.shape, np.int64)
    lib().hq_quantize_f32(x.reshape(-1), x.size, inv, f32(lo), f32(hi), q.reshape(-1))
    return q


def conv2d(x, w, bias, stride, pad):
    """Exact integer conv: x [N,Ci,H,W] ints, w [Co,Ci,KH,KW] int8-range, bias [Co] -> int64."""
    x = np.ascontiguousarray(x, np.int16)
    w = np.ascontiguousarray(w, np.int8)
    n, ci, h, wd = x.shape
    co, _, kh, kw = w.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    out = np.empty((n, co, ho, wo), np.int64)
    b = None if bias is None else np.ascontiguousarray(bias, np.int64)
    lib().hq_conv2d_nchw(x, n, ci, h, wd, w, None if b is None else b.ctypes.data, co, kh, kw,
                         stride, pad, out)
    return out


def linear(x, w, bias):
    x = np.ascontiguousarray(x, np.int16)
    w = np.ascontiguousarray(w, np.int8)
    out = np.empty((x.shape[0], w.shape[0]), np.int64)
    b = None if bias is None else np.ascontiguousarray(bias, np.int64)
    lib().hq_linear(x, x.shape[0], x.shape[1], w, None if b is None else b.ctypes.data, w.shape[0], out)
    return out


def maxpool(x, k=3, stride=2, pad=1):
    x = np.ascontiguousarray(x, np.int64)
    n, c, h, w = x.shape
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    out = np.empty((n, c, ho, wo), np.int64)
    lib().hq_maxpool_nchw(x, n * c, h, w, k, stride, pad, out)
    return out


def avgpool_trunc(x):
    x = np.ascontiguousarray(x, np.int64)
    n, c = x.shape[:2]
    out = np.empty((n, c), np.int64)
    lib().hq_avgpool_trunc(x, n * c, int(np.prod(x.shape[2:])), out)
    return out


# ----------------------------------------------------------------- host-side parameter prep
def sym_scale(lo, hi, bits):
    """symmetric_linear_quantization_params (quant_utils.py:128-152)."""
    n = f32(2 ** (bits - 1) - 1)
    s = np.maximum(np.abs(np.asarray(lo, f32)), np.abs(np.asarray(hi, f32)))
    return (np.maximum(s, f32(1e-8)) / n).astype(f32)


def asym_scale(lo, hi, bits):
    """asymmetric_linear_quantization_params (quant_utils.py:155-185); zero-point unused."""
    n = f32(2 ** bits - 1)
    return (np.maximum(np.asarray(hi, f32) - np.asarray(lo, f32), f32(1e-8)) / n).astype(f32)


def act_scale(x_min, x_max, bits, mode):
    """QuantAct scale from frozen ranges (quant_modules.py:262-270)."""
    return (sym_scale if mode == "symmetric" else asym_scale)(x_min, x_max, bits).reshape(-1)[:1]


def fold_bn(w, gamma, beta, mean, var, eps):
    """BN folding (quant_modules.py:441-449), binary32 throughout."""
    w, gamma, beta, mean, var = (np.asarray(t, f32) for t in (w, gamma, beta, mean, var))
    std = np.sqrt((var + f32(eps)).astype(f32)).astype(f32)
    sf = (gamma / std).astype(f32)
    w_f = (w * sf.reshape(-1, 1, 1, 1)).astype(f32)
    b_f = (((f32(0) - mean).astype(f32) * sf).astype(f32) + beta).astype(f32)
    return w_f, b_f


def quantize_weight(w, bits):
    """Per-output-channel symmetric weight quantisation (quant_modules.py:452-457, 477-480 /
    97-115): returns (W_int int64, S_w float32[Co])."""
    w = np.asarray(w, f32)
    flat = w.reshape(w.shape[0], -1)
    s = sym_scale(flat.min(1), flat.max(1), bits)
    n = 2 ** (bits - 1) - 1
    inv = (f32(1.0) / s).astype(f32).reshape((-1,) + (1,) * (w.ndim - 1))
    q = np.clip(np.rint((inv * w).astype(f32)), -n - 1, n)
    return q.astype(np.int64), s


def quantize_bias(b, s_w, s_a, bits=32):
    """bias -> integer at scale fl(S_w[c]*S_a) (quant_modules.py:482-484 / 117-118)."""
    bs = (np.asarray(s_w, f32) * f32(np.asarray(s_a, f32).reshape(-1)[0])).astype(f32)
    n = f32(2 ** (bits - 1) - 1)
    q = np.clip(np.rint(((f32(1.0) / bs).astype(f32) * np.asarray(b, f32)).astype(f32)), -n - f32(1), n)
    return q.astype(np.int64), bs


# ------------------------------------------------------------------------ network restatement
def _to_np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def extract_float_state(q):
    """Pull float parameters, bit-widths, modes and frozen ranges out of a Q_ResNet (the
    reference's or hawq_amd's - same attribute names, q_resnet.py:16-316).  No integers are
    taken from the model: everything integer is re-derived by this oracle."""
    def act(m):
        return dict(bits=int(m.activation_bit), mode=str(m.quant_mode),
                    x_min=_to_np(m.x_min).astype(f32), x_max=_to_np(m.x_max).astype(f32))

    def convbn(m):
        c, b = m.conv, m.bn
        return dict(bits=int(m.weight_bit), w=_to_np(c.weight), gamma=_to_np(b.weight), beta=_to_np(b.bias),
                    mean=_to_np(b.running_mean), var=_to_np(b.running_var), eps=float(b.eps),
                    stride=int(c.stride[0]), pad=int(c.padding[0]))

    st = dict(bottleneck=hasattr(q, "quant_init_convbn"), quant_input=act(q.quant_input))
    stem = q.quant_init_convbn if st["bottleneck"] else q.quant_init_block_convbn
    st["stem"] = convbn(stem)
    st["quant_act_int32"] = act(q.quant_act_int32)
    st["units"] = []
    for si, n in enumerate(q.channel):
        for ui in range(n):
            u = getattr(q, f"stage{si + 1}.unit{ui + 1}")
            d = dict(name=f"stage{si + 1}.unit{ui + 1}", resize=bool(u.resize_identity),
                     quant_act=act(u.quant_act), convbn1=convbn(u.quant_convbn1), quant_act1=act(u.quant_act1),
                     convbn2=convbn(u.quant_convbn2), quant_act_int32=act(u.quant_act_int32))
            if st["bottleneck"]:
                d["quant_act2"] = act(u.quant_act2)
                d["convbn3"] = convbn(u.quant_convbn3)
            if d["resize"]:
                d["identity"] = convbn(u.quant_identity_convbn)
            st["units"].append(d)
    st["quant_act_output"] = act(q.quant_act_output)
    fc = q.quant_output
    st["fc"] = dict(bits=int(fc.weight_bit), w=_to_np(fc.weight), b=_to_np(fc.bias))
    return st


class Trace(dict):
    """Intermediate integers of one forward, keyed like the reference's module names."""


def _conv_block(tr, name, cb, q_in, s_a, ckpt=None):
    """QuantBnConv2d frozen forward on integers: returns (acc int64 NCHW, S_w, bias_scale).

    ``ckpt[name] = dict(scale=..., bias=..., wpatch=[(flat_idx, value), ...])`` substitutes the
    integer checkpoint of a reference run (quant_train.py:665-670) for this oracle's own IEEE
    preparation; see DESIGN.md "sqrt quirk" for why the two can differ in a handful of entries."""
    w_f, b_f = fold_bn(cb["w"], cb["gamma"], cb["beta"], cb["mean"], cb["var"], cb["eps"])
    w_int, s_w = quantize_weight(w_f, cb["bits"])
    b_int, bs = quantize_bias(b_f, s_w, s_a)
    if ckpt is not None and name in ckpt:
        ov = ckpt[name]
        s_w = np.asarray(ov["scale"], f32)
        bs = (s_w * f32(np.asarray(s_a, f32).reshape(-1)[0])).astype(f32)
        b_int = np.asarray(ov["bias"], np.int64)
        w_int = w_int.copy()
        for idx, val in ov.get("wpatch", ()):
            w_int.reshape(-1)[idx] = val
    acc = conv2d(q_in, w_int, b_int, cb["stride"], cb["pad"])
    tr[name + ".weight_integer"] = w_int
    tr[name + ".bias_integer"] = b_int
    tr[name + ".convbn_scaling_factor"] = s_w
    tr[name + ".acc"] = acc
    return acc, s_w, bs


def _f32_of(acc, bs):
    """The fp32 tensor the reference carries: fl(fl(acc) * fl(S_w[c]*S_a))  (quant_modules.py:491-494)."""
    return (acc.astype(f32) * bs.reshape(1, -1, 1, 1)).astype(f32)


def forward_int(st, x, calibrate: bool = False, ckpt=None):
    """Integer forward of Q_ResNet18/50/101 (q_resnet.py:53-74 / 114-135).

    ``st``: extract_float_state().  x fp32 [N,3,H,W].  With ``calibrate`` the QuantAct ranges
    are (re)initialised from this batch exactly as one un-frozen reference forward does
    (quant_modules.py:233-250: first call sets x_min/x_max to the batch min/max) and written
    back into ``st``.  Returns (logits fp32 [N,classes], Trace)."""
    tr = Trace()
    relu = lambda a: np.maximum(a, 0)

    def scale_of(a, xf=None):
        if calibrate:
            a["x_min"] = np.asarray([xf.min()], f32)
            a["x_max"] = np.asarray([xf.max()], f32)
        return act_scale(a["x_min"], a["x_max"], a["bits"], a["mode"])

    x = np.ascontiguousarray(x, f32)
    a = st["quant_input"]
    s_in = scale_of(a, x)
    q = quantize_f32(x, s_in[0], a["bits"], a["mode"])
    tr["quant_input.q"], tr["quant_input.S"] = q, s_in

    acc, s_w, bs = _conv_block(tr, "stem", st["stem"], q, s_in, ckpt)
    acc = maxpool(acc, 3, 2, 1)
    a = st["quant_act_int32"]
    s0 = scale_of(a, _f32_of(acc, bs) if calibrate else None)
    m, e = requant_table(s_in, s_w, s0)
    r = relu(dyadic(acc, m, e, act_range(a["bits"], a["mode"])))
    tr["quant_act_int32.q"], tr["quant_act_int32.S"] = r, s0
    s_prev = s0
    one = np.ones(1, f32)

    for u in st["units"]:
        n = u["name"]
        r_f = (r.astype(f32) * s_prev[0]).astype(f32) if calibrate else None  # fl(q*S), post-ReLU
        a = u["quant_act"]
        s_a = scale_of(a, r_f)
        m, e = requant_table(s_prev, one, s_a)
        qa = dyadic(r, m, e, act_range(a["bits"], a["mode"]))
        tr[n + ".quant_act.q"], tr[n + ".quant_act.S"] = qa, s_a
        if u["resize"]:
            acc_id, s_idw, bs_id = _conv_block(tr, n + ".quant_identity_convbn", u["identity"], qa, s_a, ckpt)
            s_ida = s_a
            id_f = _f32_of(acc_id, bs_id) if calibrate else None
        else:
            acc_id, s_idw, s_ida, id_f = r, one, s_prev, r_f

        acc1, s_w1, bs1 = _conv_block(tr, n + ".quant_convbn1", u["convbn1"], qa, s_a, ckpt)
        a = u["quant_act1"]
        acc1 = relu(acc1)
        s_1 = scale_of(a, _f32_of(acc1, bs1) if calibrate else None)
        m, e = requant_table(s_a, s_w1, s_1)
        q1 = dyadic(acc1, m, e, act_range(a["bits"], a["mode"]))
        tr[n + ".quant_act1.q"], tr[n + ".quant_act1.S"] = q1, s_1

        if "convbn3" in u:
            acc2, s_w2, bs2 = _conv_block(tr, n + ".quant_convbn2", u["convbn2"], q1, s_1, ckpt)
            a = u["quant_act2"]
            acc2 = relu(acc2)
            s_2 = scale_of(a, _f32_of(acc2, bs2) if calibrate else None)
            m, e = requant_table(s_1, s_w2, s_2)
            q2 = dyadic(acc2, m, e, act_range(a["bits"], a["mode"]))
            tr[n + ".quant_act2.q"], tr[n + ".quant_act2.S"] = q2, s_2
            acc3, s_w3, bs3 = _conv_block(tr, n + ".quant_convbn3", u["convbn3"], q2, s_2, ckpt)
            s_last = s_2
        else:
            acc3, s_w3, bs3 = _conv_block(tr, n + ".quant_convbn2", u["convbn2"], q1, s_1, ckpt)
            s_last = s_1

        a = u["quant_act_int32"]
        z_f = (_f32_of(acc3, bs3) + id_f).astype(f32) if calibrate else None  # q_resnet.py:251/307
        s_o = scale_of(a, z_f)
        m1, e1 = requant_table(s_ida, s_idw, s_o)
        m2, e2 = requant_table(s_last, s_w3, s_o)
        r = relu(dyadic(acc_id, m1, e1) + dyadic(acc3, m2, e2))  # no clamp (quant_utils.py:456)
        tr[n + ".quant_act_int32.q"], tr[n + ".quant_act_int32.S"] = r, s_o
        s_prev = s_o

    pooled = avgpool_trunc(r)
    tr["final_pool.q"] = pooled
    a = st["quant_act_output"]
    s_8 = scale_of(a, (pooled.astype(f32) * s_prev[0]).astype(f32) if calibrate else None)
    m, e = requant_table(s_prev, one, s_8)
    qf = dyadic(pooled, m, e, act_range(a["bits"], a["mode"]))
    tr["quant_act_output.q"], tr["quant_act_output.S"] = qf, s_8

    fc = st["fc"]
    w_int, s_fc = quantize_weight(fc["w"], fc["bits"])
    b_int, bs = quantize_bias(fc["b"], s_fc, s_8)
    acc = linear(qf, w_int, b_int)
    tr["quant_output.weight_integer"], tr["quant_output.bias_integer"] = w_int, b_int
    tr["quant_output.fc_scaling_factor"], tr["quant_output.acc"] = s_fc, acc
    logits = (acc.astype(f32) * bs.reshape(1, -1)).astype(f32)  # quant_modules.py:127-130
    tr["logits"] = logits
    return logits, tr
