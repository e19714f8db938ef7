"""Pin the CPU oracle (oracle/) against fixtures produced by the LIVE reference.

These are `-m "not gpu"` tests: they establish that oracle/hawq_oracle.c + oracle/oracle.py
restate the reference's arithmetic exactly, so that the GPU parity tests may use the oracle
as the checker at sizes/inputs for which no golden exists.
"""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H

f32 = np.float32


@pytest.fixture(scope="module")
def kf():
    return H.load("kat_functions.npz")


@pytest.fixture(scope="module")
def km():
    return H.load("kat_modules.npz")


def test_frexp_matches_batch_frexp(kf):
    m, e = oracle.frexp_me(kf["frexp_r"])
    assert np.array_equal(m, kf["frexp_m"]) and np.array_equal(e, kf["frexp_e"])


def test_scale_formulas(kf):
    lo, hi = kf["rng_lo"], kf["rng_hi"]
    for b in (4, 8, 16):
        assert np.array_equal(oracle.sym_scale(lo, hi, b), kf[f"sym_scale_pc_{b}"])
        assert np.array_equal(oracle.sym_scale(lo, hi, b), kf[f"sym_scale_{b}"].reshape(-1))
        assert np.array_equal(oracle.asym_scale(lo, hi, b), kf[f"asym_scale_{b}"].reshape(-1))


def test_linear_quantize(kf):
    x = kf["q_x"]
    for tag in "abc":
        s = kf[f"symq_{tag}_scale"][0]
        assert np.array_equal(oracle.quantize_f32(x, s, 8, "symmetric"), kf[f"symq_{tag}_8"])
        assert np.array_equal(oracle.quantize_f32(x, s, 4, "symmetric"), kf[f"symq_{tag}_4"])
        assert np.array_equal(oracle.quantize_f32(x, s, 4, "asymmetric"), kf[f"asymq_{tag}_4"])
    # per-channel form used for weights goes through quantize_weight's formula
    s = kf["symq_pc_scale"]
    for bits in (8, 4):
        n = 2 ** (bits - 1) - 1
        inv = (f32(1) / s).astype(f32).reshape(-1, 1, 1, 1)
        q = np.clip(np.rint((inv * x).astype(f32)), -n - 1, n)
        assert np.array_equal(q, kf[f"symq_pc_{bits}"])


CASES = ["rand8", "rand4", "rand16", "tie8", "tie16", "up8"]


@pytest.mark.parametrize("tag", CASES)
def test_fixedpoint_case0(kf, tag):
    bits, sym = kf[f"fp0_{tag}_bits"]
    mode = "symmetric" if sym else "asymmetric"
    acc = kf["fp_acc"].astype(np.int64)
    m, e = oracle.requant_table(kf[f"fp0_{tag}_sa"], kf[f"fp0_{tag}_sw"], kf[f"fp0_{tag}_sout"])
    y = oracle.dyadic(acc, m, e, oracle.act_range(int(bits), mode))
    assert np.array_equal(y, kf[f"fp0_{tag}_y"].astype(np.int64))


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("itag", ["pass", "conv"])
def test_fixedpoint_case1(kf, tag, itag):
    acc, idn = kf["fp_acc"].astype(np.int64), kf["fp_idn"].astype(np.int64)
    k = f"fp1_{tag}_{itag}"
    m1, e1 = oracle.requant_table(kf[k + "_sida"], kf[k + "_sidw"], kf[f"fp0_{tag}_sout"])
    m2, e2 = oracle.requant_table(kf[f"fp0_{tag}_sa"], kf[f"fp0_{tag}_sw"], kf[f"fp0_{tag}_sout"])
    y = oracle.dyadic(idn, m1, e1) + oracle.dyadic(acc, m2, e2)
    assert np.array_equal(y, kf[k + "_y"].astype(np.int64))


def test_avgpool_trunc(kf):
    assert np.array_equal(oracle.avgpool_trunc(kf["avg_x"].astype(np.int64)),
                          kf["avg_y"].reshape(kf["avg_y"].shape[:2]).astype(np.int64))


@pytest.mark.parametrize("tag", ["c3", "c1s2", "c3s2", "c7"])
def test_quant_bn_conv_module(km, tag):
    g = lambda k: km[f"conv_{tag}_{k}"]
    cin, cout, k, stride, pad, bits, hw = g("cfg")
    w_f, b_f = oracle.fold_bn(g("w"), g("gamma"), g("beta"), g("mean"), g("var"), 1e-5)
    w_int, s_w = oracle.quantize_weight(w_f, int(bits))
    b_int, bs = oracle.quantize_bias(b_f, s_w, g("s_a"))
    # the sqrt quirk may move a scale by one ulp; everything else must be exact
    assert np.allclose(s_w, g("s_w"), rtol=2e-7, atol=0)
    if np.array_equal(s_w, g("s_w")):
        assert np.array_equal(w_int, g("weight_integer").astype(np.int64))
        assert np.array_equal(b_int, g("bias_integer").astype(np.int64))
    acc = oracle.conv2d(g("q").astype(np.int64), g("weight_integer").astype(np.int64),
                        g("bias_integer").astype(np.int64), int(stride), int(pad))
    # The reference's fp32 output is conv(x/S_a un-rounded) * scale (quant_modules.py:490-494): it is
    # NOT bit-equal to fl(acc)*scale; the next QuantAct recovers the integer by rint(z/S_a/S_w)
    # (quant_utils.py:392), which is the quantity that must match.
    z_int = np.rint(((g("y") / g("s_a")[0]).astype(f32) / g("s_w").reshape(1, -1, 1, 1)).astype(f32))
    assert np.array_equal(z_int.astype(np.int64), acc)


def test_quant_linear_module(km):
    w_int, s = oracle.quantize_weight(km["lin_w"], 8)
    b_int, bs = oracle.quantize_bias(km["lin_b"], s, km["lin_s_a"])
    assert np.array_equal(s, km["lin_fc_scaling_factor"])
    assert np.array_equal(w_int, km["lin_weight_integer"].astype(np.int64))
    assert np.array_equal(b_int, km["lin_bias_integer"].astype(np.int64))
    acc = oracle.linear(km["lin_q"].astype(np.int64), w_int, b_int)
    assert np.array_equal((acc.astype(f32) * bs.reshape(1, -1)).astype(f32), km["lin_y"])


def test_quant_avgpool_module(km):
    p = oracle.avgpool_trunc(km["pool_q"].astype(np.int64))
    y = (p.astype(f32) * km["pool_s"][0]).astype(f32)
    assert np.array_equal(y, km["pool_y"].reshape(y.shape))


def test_quant_act_input_case(km):
    lo, hi = km["act_in_rng"]
    s = oracle.act_scale(np.array([lo], f32), np.array([hi], f32), 8, "symmetric")
    assert np.array_equal(s, km["act_in_s"])
    q = oracle.quantize_f32(km["act_in_x"], s[0], 8, "symmetric")
    assert np.array_equal((q.astype(f32) * s[0]).astype(f32), km["act_in_y"])


def _build_state(arch, scheme):
    """Float state rebuilt from seeds through hawq_amd's own skeleton (no reference needed)."""
    import hashlib
    from hawq_amd.bit_schedules import get_bit_config
    from hawq_amd.skeleton import build_float_resnet, init_synthetic
    from tests.state_from_skeleton import state_from_skeleton

    fl = init_synthetic(build_float_resnet(arch), 0)
    return state_from_skeleton(fl, arch, get_bit_config(arch, scheme))


def _check_oracle_against_fixture(arch, scheme, fx, batch=2):
    """Oracle integer forward == the reference run recorded in ``fx``: frozen ranges, logits, top-1, every conv's
    accumulators and weight integers, the classifier's accumulators."""
    from hawq_amd.skeleton import synthetic_images

    x = synthetic_images(batch, 0).numpy()
    assert H.sha(x) == str(fx["input_sha"]), "RNG stream differs from the one the goldens were made with"
    st = _build_state(arch, scheme)
    # (1) calibration restatement reproduces the reference's frozen ranges (the reference's integer checkpoint is
    #     used underneath: the sqrt quirk may flip a weight integer)
    ck = H.reference_ckpt(fx, st)
    logits, tr = oracle.forward_int(st, x, calibrate=True, ckpt=ck)
    acts = [st["quant_input"], st["quant_act_int32"]]
    for u in st["units"]:
        acts += [u[k] for k in ("quant_act", "quant_act1", "quant_act2", "quant_act_int32") if k in u]
    acts.append(st["quant_act_output"])
    order = {str(n): i for i, n in enumerate(fx["act_names"])}
    got_names = ["quant_input", "quant_act_int32"]
    for u in st["units"]:
        got_names += [u["name"] + "." + k for k in ("quant_act", "quant_act1", "quant_act2", "quant_act_int32") if k in u]
    got_names.append("quant_act_output")
    for n, a in zip(got_names, acts):
        i = order[n]
        assert a["x_min"][0] == fx["act_x_min"][i] and a["x_max"][0] == fx["act_x_max"][i], n
    # (2) logits / top-1 / accumulators
    assert np.array_equal(logits, fx["logits"])
    assert np.array_equal(logits.argmax(1), fx["top1"])
    for li, n in enumerate(fx["conv_names"]):
        on = H.oracle_name(str(n))
        assert np.array_equal(H.digest(tr[on + ".acc"]), fx["conv_accdigest"][li]), n
        assert np.array_equal(H.digest(tr[on + ".weight_integer"]), fx["conv_wdigest"][li]), n
    assert np.array_equal(tr["quant_output.acc"], fx["fc_acc"])
    for k in (fx.files if hasattr(fx, "files") else fx):
        if k.startswith("acc_full."):
            assert np.array_equal(tr[H.oracle_name(k[9:]) + ".acc"], np.asarray(fx[k]).astype(np.int64))


@pytest.mark.parametrize("arch,scheme", H.NET_CONFIGS + H.NET_CONFIGS_EXTRA)
def test_network_forward_matches_reference(arch, scheme):
    """Oracle integer forward == live reference logits, accumulators and frozen ranges (committed fixtures)."""
    _check_oracle_against_fixture(arch, scheme, H.net_fixture(arch, scheme))


@pytest.mark.parametrize("scheme", ["uniform8", "uniform4", "bops_0.5"])
def test_mobilenetv2_oracle_matches_the_live_reference_fixture(scheme):
    """oracle/oracle_mbv2.py (the CPU restatement of the frozen Q_MobileNetV2 forward, ReLU6 kept on the fp32 tensor as the reference
    has it) against the fixture the UNMODIFIED reference wrote (tests/golden/make_kat_extra.py --mobilenet): on its frozen ranges
    and integer checkpoint, the int32 accumulators of ALL 54 convs (17 depthwise, the classifier) and the integers behind EVERY
    QuantAct have the recorded digests; the oracle's own weight preparation reproduces the reference's integer weights (SHA-256)
    after the recorded patches; logits equal the reference's up to the float noise of its fp32 classifier conv, same top-1."""
    import hashlib
    import torch
    from hawq_amd.api import build_quantized_model
    from hawq_amd.quant_modules import QuantAct
    from hawq_amd.skeleton import synthetic_images
    from oracle import oracle_mbv2
    fx = H.load(f"net_mobilenetv2_w1_{scheme}_b2.npz")
    model = build_quantized_model("mobilenetv2_w1", scheme, seed=0)
    acts = [(n, m) for n, m in model.named_modules() if isinstance(m, QuantAct)]
    assert [n for n, _ in acts] == [str(n) for n in fx["act_names"]]
    for i, (_, m) in enumerate(acts):
        m.x_min.fill_(float(fx["act_x_min"][i])), m.x_max.fill_(float(fx["act_x_max"][i]))
    st = oracle_mbv2.extract_float_state(model)
    ckpt, off = {}, 0
    for li, n in enumerate(str(v) for v in fx["conv_names"]):
        co = dict(model.named_modules())[n].conv.out_channels if n != "output" else model.output.out_channels
        ckpt[n] = dict(scale=fx["conv_scale"][off:off + co], bias=fx["conv_bias"][off:off + co].astype(np.int64),
                       wpatch=[(int(idx), int(val)) for l, idx, val in fx["conv_wpatch"] if l == li])
        off += co
    x = synthetic_images(2, seed=0).numpy()
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(fx["input_sha"])
    logits, tr = oracle_mbv2.forward_int(st, x, ckpt)
    for li, n in enumerate(str(v) for v in fx["conv_names"]):
        assert hashlib.sha256(np.ascontiguousarray(tr[n + ".weight_integer"].astype(np.int8)).tobytes()).hexdigest() == str(fx["conv_wsha"][li]), n
        assert np.array_equal(H.digest(tr[n + ".acc"]), fx["conv_accdigest"][li]), n
    for ai, n in enumerate(str(v) for v in fx["act_names"]):
        assert np.array_equal(H.digest(tr[n + ".q"]), fx["act_outdigest"][ai]), n
        assert int(np.abs(tr[n + ".q"]).max()) == int(fx["act_outmax"][ai]), n
    ref = fx["logits"]
    assert np.array_equal(logits.argmax(1), fx["top1"])
    s = (tr["output.conv_scaling_factor"].astype(np.float64) * float(fx["act_scale"][-1])).reshape(1, -1)
    assert np.array_equal(np.rint(logits / s), np.rint(ref / s))        # the same integers ...
    assert np.abs(logits - ref).max() <= 2 * np.spacing(np.abs(ref).max())   # ... within the reference's own float noise


def _all_schedules():
    from hawq_amd.bit_schedules import bit_config_dict
    out = []
    for key in sorted(bit_config_dict):
        if not key.startswith("bit_config_resnet"):
            continue   # the oracle restates the ResNet graph; Q_MobileNetV2 is pinned to the live reference's own fixtures
        arch, scheme = key[len("bit_config_"):].split("_", 1)
        out.append((arch, scheme))
    return out


@pytest.mark.reference
@pytest.mark.parametrize("arch,scheme", [c for c in _all_schedules() if c not in H.NET_CONFIGS + H.NET_CONFIGS_EXTRA])
def test_every_other_shipped_schedule_against_the_live_reference(arch, scheme):
    """The 16 ResNet bit schedules that have no committed fixture: run the unmodified reference here (build container
    only; skipped where /root/reference is absent) and hold the oracle to the same checks, on one image."""
    sys_path_golden = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")
    import sys
    if sys_path_golden not in sys.path:
        sys.path.insert(0, sys_path_golden)
    import make_golden
    fx = make_golden.net_fixture(arch, scheme, 1, light=True)
    _check_oracle_against_fixture(arch, scheme, fx, batch=1)


@pytest.mark.parametrize("arch,scheme", [("resnet18", "uniform8"), ("resnet50", "uniform8")])
def test_ieee_prep_differs_from_reference_only_by_sqrt_quirk(arch, scheme):
    """Oracle's own (IEEE) parameter preparation vs the reference's buffers: scales within 1 ulp,
    weight integers identical except the recorded handful, logits still identical here."""
    from hawq_amd.skeleton import synthetic_images

    fx = H.net_fixture(arch, scheme)
    st = _build_state(arch, scheme)
    x = synthetic_images(2, 0).numpy()
    logits, tr = oracle.forward_int(st, x, calibrate=True)
    off = 0
    ndiff = 0
    for li, n in enumerate(fx["conv_names"]):
        s = tr[H.oracle_name(str(n)) + ".convbn_scaling_factor"]
        ref = fx["conv_scale"][off:off + s.size]
        off += s.size
        assert np.allclose(s, ref, rtol=3e-7, atol=0)
        ndiff += int((s != ref).sum())
    assert len(fx["conv_wpatch"]) <= 4
    assert ndiff < 0.02 * off
    # ... and on these fixtures the few moved weight integers flip no rounding: the logits are the reference's
    assert np.array_equal(logits, fx["logits"]) and np.array_equal(logits.argmax(1), fx["top1"])


def _set_ranges_from_fixture(st, fx):
    order = {str(n): i for i, n in enumerate(fx["act_names"])}

    def put(a, name):
        i = order[name]
        a["x_min"] = np.array([fx["act_x_min"][i]], f32)
        a["x_max"] = np.array([fx["act_x_max"][i]], f32)

    put(st["quant_input"], "quant_input")
    put(st["quant_act_int32"], "quant_act_int32")
    for u in st["units"]:
        for k in ("quant_act", "quant_act1", "quant_act2", "quant_act_int32"):
            if k in u:
                put(u[k], u["name"] + "." + k)
    put(st["quant_act_output"], "quant_act_output")
    return st


@pytest.mark.parametrize("arch,scheme", [("resnet18", "uniform8"), ("resnet50", "uniform4"), ("resnet50", "bops_0.5")])
def test_fakequant_port_matches_reference(arch, scheme):
    """The torch-CPU fake-quant port that bench.py times as ``cpu_baseline`` computes the
    reference's logits (same top-1; bit-equal where the sqrt quirk does not intervene)."""
    import torch
    from hawq_amd.skeleton import synthetic_images
    from oracle import fakequant_port

    fx = H.net_fixture(arch, scheme)
    st = _set_ranges_from_fixture(_build_state(arch, scheme), fx)
    y = fakequant_port.forward(st, synthetic_images(2, 0)).numpy()
    assert np.array_equal(y.argmax(1), fx["top1"])
    assert np.array_equal(y, fx["logits"])


def test_b128_fixture_is_what_the_oracle_computes():
    """tests/golden/b128_*.npz (the benchmarked workloads at batch 128, consumed by the GPU parity tests and by
    bench.py) were written by tests/golden/make_b128.py from this oracle: recompute one 16-image slice of the
    ResNet18 fixture and its per-unit residual digests."""
    import hashlib
    from hawq_amd.api import build_quantized_resnet
    from hawq_amd.skeleton import synthetic_images
    from oracle import oracle
    fx = H.load("b128_resnet18_uniform8.npz")
    model = build_quantized_resnet("resnet18", "uniform8", seed=0)
    st = oracle.extract_float_state(model)
    oracle.forward_int(st, synthetic_images(int(fx["calib"]), seed=0).numpy(), calibrate=True)
    x = synthetic_images(128, seed=int(fx["seed"])).numpy()
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == str(fx["input_sha"])
    s = int(fx["slice"])
    y, tr = oracle.forward_int(st, x[3 * s:4 * s])
    assert np.array_equal(y, fx["logits"][3 * s:4 * s])
    assert np.array_equal(fx["logits"].argmax(1), fx["top1"])
    for n, want in zip(fx["residual_names"], fx["residual_sha"][3]):
        assert hashlib.sha256(np.ascontiguousarray(tr[str(n)].astype(np.int32)).tobytes()).hexdigest() == str(want), n
