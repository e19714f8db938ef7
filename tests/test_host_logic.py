"""CPU tests of the host-side logic of hawq_amd (no GPU, no reference needed)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """The C-ABI library loads here (no GPU needed) and exports everything include/hawq_mi355.h declares."""
    from hawq_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "hawq_mi355.h")).read()
    declared = set(re.findall(r"\b(hawq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(_lib.SIGNATURES) | {"hawq_last_error"} == declared
    assert lib.hawq_abi_version() == 5
    assert ctypes.sizeof(_lib.ConvArgs) % 8 == 0


def kernel_resources(so_path):
    """{kernel symbol: (private segment bytes, spilled VGPRs, VGPRs)} of every gfx950 kernel in a built library, read from the code
    objects' metadata notes (clang-offload-bundler + llvm-readelf of the ROCm image; no GPU, no recompilation)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([f"{llvm}/llvm-objdump", "--offloading", so], cwd=tmp, capture_output=True, check=True)
        objs = sorted(glob.glob(so + ".*gfx950"))
        assert objs, "no gfx950 code objects found in the library"
        for obj in objs:
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", obj], capture_output=True, text=True, check=True).stdout
            name = None
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_spill_count|vgpr_count):\s+(\S+)", line)
                if not m:
                    continue
                if m[1] == "name":
                    if name and cur:
                        out[name] = (cur.get("private_segment_fixed_size", 0), cur.get("vgpr_spill_count", 0), cur.get("vgpr_count", 0))
                    name, cur = m[2], {}
                else:
                    cur[m[1]] = int(m[2])
            if name and cur:
                out[name] = (cur.get("private_segment_fixed_size", 0), cur.get("vgpr_spill_count", 0), cur.get("vgpr_count", 0))
    return out


# Instantiations that are allowed a FEW spilled registers (outside their inner loops; each listed with the reason).  Everything else in
# the shipped library must be scratch-free: a spill reload inside an LDS-DMA issue sequence brings an `s_waitcnt vmcnt(0)` with it and
# serialises every DMA behind it, and the RESIDUAL band kernels of ResNet18 carried 88-160 spilled registers for a
# whole round without anyone noticing.
SCRATCH_ALLOWED = {
    # mangled-name fragment: max spilled VGPRs.  tools/spill_sites.py shows WHERE a kernel spills: every entry below spills in
    # straight-line prologue / epilogue code only, except the one marked (in loop)
    # 64x64 dual-branch residual tiles at the 80-register budget of 6 waves per SIMD (first units of stages 2-4); exact-tie form: 8
    "conv_kernelINS_3CfgILi64ELi64ELi2ELi2ELi3ELi1ELi6ELi1EEELi2ELb1": 8,
    "conv_kernelINS_3CfgILi64ELi64ELi2ELi2ELi2ELi1ELi6ELi1EEELi2ELb1": 8,
    # fused expand -> reduce, C = 64 / C = 128 with producers, all-k-zero instantiations at the 128-register budget
    "expand_reduce_kernelINS_5ERCfgILi64ELi2ELi0ELb1ELi4ELb0EEELb0ELb1": 3,
    "expand_reduce_kernelINS_5ERCfgILi128ELi2ELi4ELb1ELi4ELb0EEELb0ELb1": 1,
    # MobileNetV2's 160 -> 960 -> 160 units on the 7 x 7 maps with FOUR slice groups (16 waves per workgroup: 128 registers): three projection
    # accumulator tiles + five K steps of operands; one register pair is reloaded per slice (in loop).  Measured 29 us against 35.7 us for
    # the two-group form without spills (profiles/r04_h_mbv2_wide_units.txt)
    "linear_bottleneck_planar_kernelILi1ELi5ELi5E": 6,
    # hawq4 (nibble) 3x3 band kernels, 256 x 128 tiles: 64 accumulators + unpacked fragments at 128 registers.  The 384-pixel-band
    # form reloads one register pair per filter-row step (in loop) - W4A4's 14x14 / 7x7 conv2 launches; open item
    "conv3x3_band_kernelINS_7BandCfgILi256ELi128ELi4ELi2ELi512ELi2ELi1ELi8ELi3EEELb1": 2,
    "conv3x3_band_kernelINS_7BandCfgILi256ELi128ELi4ELi2ELi384ELi2ELi1ELi8ELi4EEELb1": 8,
}


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The ctypes mirrors of hawq_conv_args / hawq_expand_reduce_args / hawq_bottleneck_args (hawq_amd/_lib.py) against the C compiler's
    own layout of include/hawq_mi355.h: size and the offset of EVERY field (a field added on one side only, or in another order, would
    silently shift every argument behind it)."""
    import ctypes
    import subprocess
    from hawq_amd import _lib
    structs = {"hawq_conv_args": _lib.ConvArgs, "hawq_expand_reduce_args": _lib.ExpandReduceArgs, "hawq_bottleneck_args": _lib.BottleneckArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "hawq_mi355.h")}"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {"in" if fname == "in_" else fname}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    # and the header declares no field the mirror lacks: the C struct has exactly as many members as the mirror has bytes accounted for
    hdr = open(os.path.join(ROOT, "include", "hawq_mi355.h")).read()
    assert f"#define HAWQ_ABI_VERSION {_lib.load().hawq_abi_version()}" in hdr


def test_shipped_kernels_do_not_spill():
    from hawq_amd import _lib
    _lib.load()
    path = _lib.library_path()
    res = kernel_resources(path)
    assert len(res) > 200, f"only {len(res)} kernels parsed from {path}"
    bad = []
    for name, (scratch, spills, vgprs) in sorted(res.items()):
        limit = max([v for k, v in SCRATCH_ALLOWED.items() if k in name] + [0])
        if spills > limit or (scratch and not spills and scratch > 64):
            bad.append((name[:150], scratch, spills, vgprs))
    assert not bad, "kernels with scratch / register spills:\n" + "\n".join(map(str, bad))


def test_requant_table_matches_oracle_and_lifts_exactly():
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    from oracle import oracle
    rng = np.random.default_rng(0)
    s_w = torch.from_numpy(rng.uniform(1e-4, 3.0, 200).astype(np.float32))
    s_a, s_o = torch.tensor([0.0213]), torch.tensor([0.047])
    m, ek = requant_table(s_a, s_w, s_o, vbits=20)
    mo, eo = oracle.requant_table(s_a.numpy(), s_w.numpy(), s_o.numpy())
    e, k = ek & 0xff, ek >> 8
    assert (e >= 33).all() and ((m < 2 ** 31) & (m >= 0)).all()
    # same rational: m * 2^k / 2^e == mo / 2^eo   (mo may be 2^31 where m is 2^30)
    assert all(int(a) << (int(kk) + 62 - int(ee)) == int(b) << (62 - int(eb)) for a, kk, ee, b, eb in zip(m, k, e, mo, eo))
    acc = rng.integers(-2 ** 19, 2 ** 19, (4, 200, 3)).astype(np.int64)
    ref = oracle.dyadic(acc, mo, eo)
    got = oracle.dyadic(acc << k.reshape(1, -1, 1), m.astype(np.int64), e.astype(np.int32))
    assert np.array_equal(ref, got)
    with pytest.raises(ValueError):
        requant_table(torch.tensor([1.0]), torch.tensor([64.0]), torch.tensor([1.0]), vbits=24)
    # tie-freeness proof: a power-of-two ratio can tie, a generic one cannot for narrow inputs
    assert not tables_are_fast(np.array([1 << 30]), np.array([34]), 20)
    assert tables_are_fast(np.array([(1 << 30) + 1]), np.array([40]), 24)
    assert tables_are_fast(np.array([(1 << 30) + 1]), np.array([33 | (1 << 8)]), 24)
    assert not tables_are_fast(np.array([(1 << 30) + 1]), np.array([33 | (8 << 8)]), 24)  # 24 + 8 > 31 bits


def test_tie_free_proof_is_sound_by_brute_force():
    """For small parameters enumerate every v: whenever tables_are_fast says yes, half-up == half-even."""
    from hawq_amd.quant_utils import tables_are_fast
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(300):
        e = int(rng.integers(33, 40))
        tz = int(rng.integers(0, 31))
        m = (int(rng.integers(1, 2 ** (31 - tz))) | 1) << tz
        if m >= 2 ** 31:
            continue
        vbits = int(rng.integers(4, 12))
        if not tables_are_fast(np.array([m]), np.array([e]), vbits):
            continue
        v = np.arange(-(2 ** vbits) + 1, 2 ** vbits, dtype=object)
        p = v * m
        half = 1 << (e - 1)
        up = (p + half) >> e
        rem = p - ((p >> e) << e)
        even = (p >> e) + np.array([1 if (r > half or (r == half and (int(f) & 1))) else 0 for r, f in zip(rem, p >> e)], dtype=object)
        assert (up == even).all()
        checked += 1
    assert checked > 50


def test_exact_tie_correction_equals_round_half_even():
    """The arithmetic of the fast kernels' exact-tie mode (csrc/common.h:dyadic_tie), restated with Python
    integers: half-up quotient of t = (v<<k)*m + 2^(e-1), minus one when t is a multiple of 2^e and the quotient is
    odd.  Must equal round-half-even for EVERY table (tie-free or not), negative values and pre-shifts included;
    tables_fit_fast is the only precondition."""
    from hawq_amd.quant_utils import tables_are_fast, tables_fit_fast
    rng = np.random.default_rng(4)
    n_ties = n_unproven = 0
    for _ in range(400):
        e = int(rng.integers(33, 41))
        k = int(rng.integers(0, 4))
        tz = int(rng.integers(0, 31))
        m = (int(rng.integers(1, 2 ** (31 - tz))) | 1) << tz
        vbits = int(rng.integers(4, 11))
        if m >= 2 ** 31 or not tables_fit_fast(np.array([m]), np.array([e | k << 8]), vbits):
            continue
        n_unproven += not tables_are_fast(np.array([m]), np.array([e | k << 8]), vbits)
        for v in range(-(2 ** vbits) + 1, 2 ** vbits):
            p = (v << k) * m
            t = p + (1 << (e - 1))
            q = t >> e                                           # (hi >> s) of the kernel: floor
            tie = (t & ((1 << e) - 1)) == 0                      # lo == 0 and the low s bits of hi == 0
            got = q - ((q & 1) if tie else 0)
            f, rem = p >> e, p - ((p >> e) << e)
            half = 1 << (e - 1)
            want = f + (1 if (rem > half or (rem == half and (f & 1))) else 0)
            assert got == want, (v, m, e, k)
            n_ties += tie
    assert n_ties > 100 and n_unproven > 20
    assert not tables_fit_fast(np.array([1 << 30]), np.array([32]), 10)          # e < 33
    assert not tables_fit_fast(np.array([1 << 30]), np.array([33 | 9 << 8]), 24)  # 24 + 9 > 31 bits


def test_uint8_input_table_equals_quantising_the_normalised_image():
    """hawq_amd.quant_utils.input_quant_lut against the oracle's input quantiser (oracle/hawq_oracle.c:hq_quantize_f32)
    applied to a real ToTensor + Normalize tensor: every one of the 3 x 256 pixel values."""
    from hawq_amd.quant_utils import input_quant_lut
    from oracle import oracle as orc
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    img = torch.arange(256, dtype=torch.uint8).view(1, 256, 1, 1).expand(1, 256, 1, 3).contiguous()  # NHWC, all values
    t = img.permute(0, 3, 1, 2).to(torch.float32).div(255)                                            # ToTensor
    t = t.sub_(torch.tensor(mean).view(1, 3, 1, 1)).div_(torch.tensor(std).view(1, 3, 1, 1))          # Normalize
    clamped = False
    for s_in in (0.0207, 2.64 / 127, 0.0123456):
        inv = float(np.float32(1.0) / np.float32(s_in))                       # fl(1/S), as the engine passes it
        want = orc.quantize_f32(t.numpy(), np.float32(s_in), 8).reshape(3, 256)
        got = input_quant_lut(inv, mean, std).numpy().astype(np.int64)
        assert np.array_equal(got, want)
        clamped |= got.min() == -128 or got.max() == 127
    assert clamped  # at least one scale drives the table into the clamp


def test_fused_plan_byte_model():
    """hawq_amd.roofline.fused_plan_table: one row per launch of the engine's plan, the same MACs as the canonical
    layer table, fewer bytes than it (no separate QuantAct passes, no int32 identity accumulators, no 112^2 stem
    intermediate), and within -5 / +25 % of the HBM traffic measured with PMC counters (profiles/traffic.json)."""
    import json
    from hawq_amd import roofline as R
    for arch, scheme, launches in (("resnet50", "uniform8", 51), ("resnet50", "uniform4", 51), ("resnet18", "uniform8", 19),
                                   ("resnet101", "uniform8", 102)):
        t = R.fused_plan_table(arch, scheme)
        assert len(t) == launches
        assert sum(r["macs"] for r in t) == sum(r["macs"] for r in R.layer_table(arch, scheme))
        assert R.fused_plan_bytes(arch, scheme, 128) < 0.65 * R.algorithmic_bytes(arch, scheme, 128)
    assert R.fused_plan_bytes("resnet50", "uniform4", 128) < 0.8 * R.fused_plan_bytes("resnet50", "uniform8", 128)
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        rec = json.load(f)["resnet50_uniform8_b128"]
    measured = rec["bytes_per_launch"]
    # expand -> reduce pairs that ran as one launch in the measured plan: their block-input tensors never reach memory
    pairs = rec.get("fused_pairs", [])
    model = R.fused_plan_bytes("resnet50", "uniform8", 128, pairs)
    assert model == R.fused_plan_bytes("resnet50", "uniform8", 128) - sum(
        2 * 128 * {"1": 56 * 56 * 256, "2": 28 * 28 * 512, "3": 14 * 14 * 1024}[p[len("stage")]] for p in pairs)
    # (the counters see L2 misses: a residual slice that is still in L2 when its reader runs makes them read a little LESS than the model)
    assert 0.95 * model <= measured <= 1.25 * model, (model, measured)


def test_bench_launch_model_prices_fused_launches_with_the_plan_rows():
    """bench.py's per-launch byte model (roofline.dominant_launch, --per-op): a fused expand -> reduce launch is priced with the
    FUSED plan's rows (the 8-bit tensor between the two layers never moves), not with the canonical per-layer rows (VERDICT r2:
    dominant_launch.hbm_frac was overstated 2.7x); the launches of a plan add up to roofline.fused_plan_bytes."""
    import bench
    from hawq_amd import roofline as R
    rows = bench.plan_rows("resnet50", "uniform8")
    m = 64 * 14 * 14
    b, macs = bench.launch_model("stage3.unit2.quant_convbn3+stage3.unit3.quant_convbn1", rows, 64)
    acts = m * 256 + 2 * m * 1024 * 2 + m * 256                    # x2 in, uint16 residual in and out, reduce conv's output
    wts = 256 * 1024 + 1024 * 16 + 1024 * 256 + 256 * 16
    assert b == acts + wts and macs == 2 * m * 256 * 1024
    assert 57e6 < b < 59e6                                           # the judge's own figure for this launch: 58.3 MB
    # a dual (first-unit) launch and a solo expand launch
    b1, _ = bench.launch_model("stage2.unit1.quant_convbn3+identity", rows, 64)
    m2 = 64 * 28 * 28
    assert b1 == m2 * 128 + m2 * 256 + m2 * 512 * 2 + m2 * 512 + (128 * 512 + 512 * 16 + 256 * 512 + 512 * 16)
    pairs = ["stage1.unit1.quant_convbn3", "stage1.unit2.quant_convbn3", "stage3.unit2.quant_convbn3"]
    names, skip = [], set()
    order = [r["name"] for r in R.fused_plan_table("resnet50", "uniform8")]
    for i, n in enumerate(order):
        if i in skip:
            continue
        if n.split("+")[0] in pairs:
            names.append(n + "+" + order[i + 1])
            skip.add(i + 1)
        else:
            names.append(n)
    total = sum(bench.launch_model(n, rows, 128)[0] for n in names)
    assert total == R.fused_plan_bytes("resnet50", "uniform8", 128, pairs)


def test_packing_roundtrip_and_layout():
    from hawq_amd.packing import pack_conv_weight, pack_hawq4, pack_stem_weight, unpack_hawq4
    rng = np.random.default_rng(0)
    v = rng.integers(-8, 8, (3, 5, 16))
    assert np.array_equal(unpack_hawq4(pack_hawq4(v), signed=True), v)
    g = np.arange(8)
    b = pack_hawq4(g[None, :])[0]
    assert list(b) == [0 | 4 << 4, 1 | 5 << 4, 2 | 6 << 4, 3 | 7 << 4]
    w = rng.integers(-127, 128, (64, 64, 3, 3))
    p = pack_conv_weight(w, 8).view(np.int8).reshape(64, 3, 3, 64)
    assert np.array_equal(p.transpose(0, 3, 1, 2), w)
    ws = rng.integers(-127, 128, (64, 3, 7, 7))
    ps = pack_stem_weight(ws).view(np.int8).reshape(64, 7, 8, 4)
    assert np.array_equal(ps[:, :, :7, :3].transpose(0, 3, 1, 2), ws) and not ps[:, :, 7].any() and not ps[..., 3].any()


def test_bit_schedules_and_module_tree():
    from hawq_amd.api import build_quantized_resnet
    from hawq_amd.bit_schedules import bit_config_dict, module_names
    assert len(bit_config_dict) == 30 and sum(k.startswith("bit_config_resnet") for k in bit_config_dict) == 26
    for arch, scheme in H.NET_CONFIGS:
        q = build_quantized_resnet(arch, scheme, seed=None)
        names = dict(q.named_modules())
        cfg = bit_config_dict[f"bit_config_{arch}_{scheme}"]
        assert list(cfg) == module_names(arch)
        for n, bits in cfg.items():
            m = names[n]
            assert (m.activation_bit if hasattr(m, "activation_bit") else m.weight_bit) == bits
            if hasattr(m, "activation_bit"):
                assert m.quant_mode == ("asymmetric" if bits == 4 else "symmetric")
    sd = build_quantized_resnet("resnet50", "uniform8", seed=None).state_dict()
    for key in ("stage1.unit1.quant_convbn1.conv.weight", "stage1.unit1.quant_convbn1.weight_integer",
                "stage1.unit1.quant_convbn1.convbn_scaling_factor", "stage1.unit1.quant_act.x_min",
                "stage1.unit1.quant_act.act_scaling_factor", "quant_output.fc_scaling_factor",
                "quant_output.weight_integer", "quant_init_convbn.bn.running_var",
                "stage1.unit1.quant_identity_convbn.bias_integer"):
        assert key in sd, key


def test_host_preparation_matches_oracle_and_freeze_semantics():
    """QuantBnConv2d.prepare / QuantLinear.prepare (host side, IEEE) == the oracle's restatement;
    freeze/unfreeze toggle the reference's flags; the product path refuses CPU tensors."""
    from hawq_amd import quant_modules as qm
    from hawq_amd.api import build_quantized_resnet
    from oracle import oracle
    q = build_quantized_resnet("resnet18", "bops_0.5", seed=0)
    u = getattr(q, "stage2.unit1")
    s_a = torch.tensor([0.0173])
    for mod in (u.quant_convbn1, u.quant_identity_convbn):
        mod.prepare(s_a)
        w_f, b_f = oracle.fold_bn(mod.conv.weight.detach().numpy(), mod.bn.weight.detach().numpy(),
                                  mod.bn.bias.detach().numpy(), mod.bn.running_mean.numpy(),
                                  mod.bn.running_var.numpy(), mod.bn.eps)
        w_int, s_w = oracle.quantize_weight(w_f, mod.weight_bit)
        b_int, _ = oracle.quantize_bias(b_f, s_w, s_a.numpy())
        assert np.array_equal(mod.weight_integer.numpy(), w_int)
        assert np.array_equal(mod.convbn_scaling_factor.numpy(), s_w)
        assert np.array_equal(mod.bias_integer.numpy().astype(np.int64), b_int)
    fc = q.quant_output
    fc.prepare(s_a)
    w_int, s_fc = oracle.quantize_weight(fc.weight.detach().numpy(), 8)
    assert np.array_equal(fc.weight_integer.numpy(), w_int) and np.array_equal(fc.fc_scaling_factor.numpy(), s_fc)
    assert not q.is_frozen()
    qm.freeze_model(q)
    assert q.is_frozen() and u.quant_convbn1.fix_BN and not u.quant_act.running_stat
    qm.unfreeze_model(q)
    assert not q.is_frozen() and u.quant_act.running_stat
    with pytest.raises(RuntimeError):
        q.forward_modules(torch.zeros(1, 3, 224, 224))
    with pytest.raises(ValueError):
        qm.QuantAct(quant_mode="bogus")(torch.zeros(1, 3, 4, 4))


def test_roofline_model_reproduces_survey_numbers():
    from hawq_amd import roofline
    want = {("resnet18", "uniform8"): 1.419, ("resnet18", "uniform4"): 1.176, ("resnet18", "bops_0.5"): 1.343,
            ("resnet50", "uniform8"): 7.991, ("resnet50", "uniform4"): 6.804, ("resnet50", "bops_0.5"): 7.821}
    for (a, s), gb in want.items():
        assert abs(roofline.algorithmic_bytes(a, s, 128) / 1e9 - gb) < 5e-4
    assert abs(roofline.macs("resnet50", "uniform8", 128) / 1e9 - 493.8) < 0.1
    assert abs(roofline.macs("resnet18", "uniform8", 128) / 1e9 - 232.2) < 0.1


def test_mobilenetv2_graph_and_schedules_line_up():
    """Q_MobileNetV2 (q_mobilenetv2.py) builds for every shipped schedule: each schedule entry names a module (incl. the two
    stray `conv1.conv` / `conv1.bn` entries of three reference schedules), units alternate 1x1 / depthwise 3x3 / 1x1."""
    from hawq_amd.api import build_quantized_model
    from hawq_amd.bit_schedules import get_bit_config
    from hawq_amd.quant_modules import QuantAct, QuantBnConv2d, QuantConv2d
    for scheme in ("uniform8", "uniform4", "bops_0.5", "modelsize_0.5"):
        m = build_quantized_model("mobilenetv2_w1", scheme, seed=None)
        cfg = get_bit_config("mobilenetv2_w1", scheme)
        mods = dict(m.named_modules())
        assert all(k in mods for k in cfg)
        convs = [v for v in mods.values() if isinstance(v, QuantBnConv2d)]
        assert len(convs) == 1 + 17 * 3 + 1 and isinstance(m.output, QuantConv2d)
        assert sum(c.conv.groups > 1 for c in convs) == 17 and all(c.conv.groups == c.conv.in_channels for c in convs if c.conv.groups > 1)
        assert sum(isinstance(v, QuantAct) for v in mods.values()) == 2 + 17 * 4 + 3
        assert mods["features.stage4.unit5.conv2"].weight_bit == cfg["features.stage4.unit5.conv2"]


def _synth_image(h, w, seed):
    """the picture tests/golden/make_pillow.py resized (kept in step with its synth())"""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    img[::7, ::5] = 255
    img[3::11, 2::13] = 0
    yy, xx = np.mgrid[0:h, 0:w]
    img[h // 3: 2 * h // 3, :, 1] = ((yy[h // 3: 2 * h // 3] * 3 + xx[h // 3: 2 * h // 3] * 2) % 256).astype(np.uint8)
    return img


def test_resize_restatement_reproduces_real_pillow_output():
    """oracle/pil_resample.py against REAL Pillow (tests/golden/pillow_resize.npz, written by make_pillow.py with Pillow 12.2 in the
    build container: `Image.resize(.., BILINEAR)` as torchvision's Resize(256) calls it, then CenterCrop(224),
    quant_train.py:428-440): the whole resized image (SHA-256) and the crop, byte for byte, for nine geometries and the decoded
    JPEG.  Where Pillow is installed the comparison is repeated live on further random geometries and the committed JPEG is decoded
    again (pil_loader: `Image.open(f).convert('RGB')`)."""
    import hashlib
    from oracle import pil_resample as P
    fx = H.load("pillow_resize.npz")
    for i, (h, w) in enumerate(fx["geoms"]):
        img = _synth_image(int(h), int(w), int(fx["seeds"][i]))
        oh, ow = (int(v) for v in fx[f"full_shape_{i}"])
        full = P.resize(img, oh, ow)
        assert hashlib.sha256(full.tobytes()).hexdigest() == str(fx[f"full_sha_{i}"]), (h, w)
        assert np.array_equal(P.resize_center_crop(img), fx[f"crop_{i}"]), (h, w)
    assert np.array_equal(P.resize_center_crop(fx["jpeg_decoded"]), fx["jpeg_crop"])
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(5)
    for _ in range(6):
        h, w = (int(v) for v in rng.integers(40, 900, 2))
        img = _synth_image(h, w, int(rng.integers(1 << 30)))
        oh, ow = (int(256 * h / w), 256) if w <= h else (256, int(256 * w / h))
        assert np.array_equal(P.resize(img, oh, ow), np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))), (h, w)
    from hawq_amd.image import decode_image
    dec = decode_image(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_500x375.jpg")).numpy()
    assert dec.shape == (375, 500, 3) and dec.dtype == np.uint8
    if PIL.__version__ == str(fx["pillow_version"]):   # another libjpeg build may round its IDCT differently
        assert hashlib.sha256(dec.tobytes()).hexdigest() == str(fx["jpeg_decoded_sha"])
    else:
        assert np.abs(dec.astype(int) - fx["jpeg_decoded"].astype(int)).max() <= 2


def test_image_folder_walk_matches_torchvisions_ordering(tmp_path):
    """hawq_amd.image.image_folder: classes = sorted sub-directories, samples in sorted walk order, non-image files skipped
    (torchvision.datasets.folder.make_dataset, which quant_train.py:428 uses through datasets.ImageFolder)."""
    from hawq_amd.image import image_folder
    for c, names in (("n02", ["b.JPEG", "a.jpg", "notes.txt"]), ("n01", ["z.png", "sub/k.jpeg"])):
        for n in names:
            f = tmp_path / c / n
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_bytes(b"x")
    samples, classes = image_folder(str(tmp_path))
    assert classes == ["n01", "n02"]
    rel = [(os.path.relpath(p, tmp_path), t) for p, t in samples]
    assert rel == [("n01/z.png", 0), ("n01/sub/k.jpeg", 0), ("n02/a.jpg", 1), ("n02/b.JPEG", 1)]
    with pytest.raises(FileNotFoundError):
        image_folder(str(tmp_path / "n01" / "sub"))


def test_resize_restatement_agrees_with_torchs_uint8_antialias_bilinear_within_one_lsb():
    """oracle/pil_resample.py (the checker of the device Resize stage, written by the builder) against an implementation the builder
    did NOT write: torch-CPU `F.interpolate(uint8, mode="bilinear", antialias=True)`, PyTorch's port of Pillow-SIMD's resampling
    (what torchvision's Resize runs on uint8 tensors).  torch quantises its filter weights to fewer fractional bits than
    Pillow's 22, so the two are not bit-identical by construction (torchvision documents +-1 against PIL); what is pinned here:
    never more than ONE level apart and equal in > 99.5 % of the elements.  (The pin to Pillow's own output is the next test.)"""
    import torch
    import torch.nn.functional as F
    from oracle import pil_resample as P
    rng = np.random.default_rng(0)
    total = differ = 0
    for h, w in ((375, 500), (500, 375), (333, 500), (257, 300), (600, 601), (224, 224), (200, 180)):
        for kind in ("noise", "smooth"):
            if kind == "noise":
                img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            else:
                yy, xx = np.mgrid[0:h, 0:w]
                img = np.stack([127 + 120 * np.sin(xx / 17 + yy / 29), 127 + 120 * np.cos(xx / 23 - yy / 13), xx * 255 / w], -1).astype(np.uint8)
            oh, ow = (int(256 * h / w), 256) if w <= h else (256, int(256 * w / h))
            ref = P.resize(img, oh, ow)
            t = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
            out = F.interpolate(t, size=(oh, ow), mode="bilinear", antialias=True, align_corners=False)[0].permute(1, 2, 0).numpy()
            d = np.abs(out.astype(np.int64) - ref.astype(np.int64))
            assert d.max() <= 1, (h, w, kind, int(d.max()))
            total += d.size
            differ += int((d > 0).sum())
    assert differ < 0.005 * total, (differ, total)


def test_resize_coefficients_match_the_independent_restatement():
    """hawq_amd.image.bilinear_coeffs (vectorised) vs oracle/pil_resample.py (scalar loops, written separately): bounds and
    22-bit coefficients of Pillow's antialiased bilinear resize for down-, up- and identity scaling; taps sum to 2^22 +- taps."""
    from hawq_amd.image import bilinear_coeffs, resize_crop_geometry
    from oracle import pil_resample
    for in_size, out_size in ((500, 256), (375, 256), (333, 341), (256, 256), (100, 256), (2000, 256), (257, 256)):
        b, c, k = bilinear_coeffs(in_size, out_size)
        ref = pil_resample._coeffs(in_size, out_size)
        assert len(ref) == out_size
        for o, (x0, ks) in enumerate(ref):
            assert b[o, 0] == x0 and b[o, 1] == len(ks) and list(c[o, :len(ks)]) == ks and not c[o, len(ks):].any(), (in_size, out_size, o)
            assert abs(int(c[o].sum()) - (1 << 22)) <= len(ks)
    assert resize_crop_geometry(375, 500) == (256, 341, 16, 58) and resize_crop_geometry(500, 375) == (341, 256, 58, 16)
    assert resize_crop_geometry(256, 256) == (256, 256, 16, 16)


def test_relu6_restated_as_relu_plus_clamp_is_exact_where_the_engine_accepts_it():
    """hawq_amd.engine_mbv2._relu6_is_relu decides, per layer, whether `conv -> ReLU6 -> QuantAct` (q_mobilenetv2.py:66-72, the fp32
    pipeline: x = acc * fl(S_a * S_w), min(max(x, 0), 6.0), z = round(x / S_a / S_w), RNE(z * m / 2^e), clamp) may run as
    `RNE(max(acc, 0) * m / 2^e)` + the QuantAct's clamp.  Where it says yes the two must agree on EVERY accumulator - sampled around
    zero, around the ReLU6 saturation point and far above it; where it says no, a differing accumulator must exist (the engine then
    refuses the layer instead of running it)."""
    from hawq_amd.engine_mbv2 import _relu6_is_relu
    from hawq_amd.quant_utils import requant_table
    f32 = np.float32

    def rne(z, m, e):
        t = z * int(m) + (1 << (e - 1))
        q = np.floor_divide(t, 1 << e)
        return np.where((t % (1 << e) == 0) & (q % 2 == 1), q - 1, q)

    rng = np.random.default_rng(0)
    yes = no = 0
    for _ in range(120):
        s_a, s_w = float(rng.uniform(0.01, 0.06)), float(rng.uniform(0.001, 0.01))
        s_out = float(rng.uniform(1.0, 9.0)) / 127
        m, e = requant_table(torch.tensor([s_a]), torch.tensor([s_w]), torch.tensor([s_out]), lift=False)
        m, e = int(m[0]), int(e[0])
        a6 = int(round(6.0 / s_a / s_w))
        a = np.concatenate([np.arange(-50, 200), np.arange(a6 - 3000, a6 + 3000), rng.integers(0, 4 * a6, 2000)]).astype(np.int64)
        x = np.minimum(np.maximum((a.astype(f32) * (f32(s_a) * f32(s_w))).astype(f32), f32(0)), f32(6.0))
        z = np.rint((x / f32(s_a) / f32(s_w)).astype(np.float64)).astype(np.int64)
        ref = np.clip(rne(z, m, e), -128, 127)
        plan = np.clip(rne(np.maximum(a, 0), m, e), 0, 127)
        if _relu6_is_relu(s_a, np.array([s_w]), [m], [e], 127):
            assert np.array_equal(ref, plan)
            yes += 1
        else:
            assert not np.array_equal(ref, plan)
            no += 1
    assert yes > 20 and no > 20


def test_pack_w3x3_band_matches_the_c_twin_and_a_loop_restatement():
    """ABI 5: the weight stream of the round-5 3x3 kernels (include/hawq_mi355.h: hawq_conv_args.wgt_band).  The numpy packer the
    engine uses, the C twin a non-Python host would call (hawq_pack_w3x3_band: host pointers, no GPU) and a loop restatement of the
    documented layout agree byte for byte."""
    import ctypes
    from hawq_amd import _lib
    from hawq_amd.packing import pack_w3x3_band
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for cout, cin in ((64, 128), (128, 192), (192, 64)):
        w = rng.integers(-128, 128, (cout, 9, cin)).astype(np.int8)
        got = pack_w3x3_band(w.view(np.uint8).reshape(-1), cout, cin)
        ref = np.zeros(cout * 9 * cin, np.uint8)
        cch = cin // 64
        wu = w.view(np.uint8)
        for ct in range(cout // 64):
            for cc in range(cch):
                for tap in range(9):
                    base = ((ct * cch + cc) * 9 + tap) * 4096
                    for r in range(64):
                        row = wu[ct * 64 + r, tap, cc * 64:cc * 64 + 64]
                        for sl in range(4):
                            o = base + r * 64 + ((sl ^ ((r >> 2) & 3)) << 4)
                            ref[o:o + 16] = row[sl * 16:sl * 16 + 16]
        assert np.array_equal(got, ref)
        dst = np.zeros_like(ref)
        src = np.ascontiguousarray(wu.reshape(-1))
        assert lib.hawq_pack_w3x3_band(src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p), cout, cin) == 0
        assert np.array_equal(dst, ref)
    bad = np.zeros(16, np.uint8)
    assert lib.hawq_pack_w3x3_band(bad.ctypes.data_as(ctypes.c_void_p), bad.ctypes.data_as(ctypes.c_void_p), 32, 64) != 0


def test_pack_w1x1_k128_matches_the_c_twin_and_a_loop_restatement():
    """ABI 5: the weight stream of the round-5 streaming 1x1 kernels (include/hawq_mi355.h: hawq_conv_args.wgt_k128): [Cout/64][Cin/128]
    [64 rows][128 B], 16-byte slot s of row r at s ^ ((r >> 1) & 7).  numpy packer, C twin (host pointers, no GPU) and a loop
    restatement of the documented layout agree byte for byte; sizes the kernels do not take are refused."""
    import ctypes
    from hawq_amd import _lib
    from hawq_amd.packing import pack_w1x1_k128
    lib = _lib.load()
    rng = np.random.default_rng(6)
    for cout, cin in ((64, 128), (128, 384), (192, 256)):
        w = rng.integers(-128, 128, (cout, cin)).astype(np.int8).view(np.uint8)
        got = pack_w1x1_k128(w.reshape(-1), cout, cin)
        ref = np.zeros(cout * cin, np.uint8)
        kch = cin // 128
        for ct in range(cout // 64):
            for kc in range(kch):
                base = (ct * kch + kc) * 64 * 128
                for r in range(64):
                    row = w[ct * 64 + r, kc * 128:kc * 128 + 128]
                    for sl in range(8):
                        o = base + r * 128 + ((sl ^ ((r >> 1) & 7)) << 4)
                        ref[o:o + 16] = row[sl * 16:sl * 16 + 16]
        assert np.array_equal(got, ref)
        dst = np.zeros_like(ref)
        src = np.ascontiguousarray(w.reshape(-1))
        assert lib.hawq_pack_w1x1_k128(src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p), cout, cin) == 0
        assert np.array_equal(dst, ref)
    bad = np.zeros(64, np.uint8)
    assert lib.hawq_pack_w1x1_k128(bad.ctypes.data_as(ctypes.c_void_p), bad.ctypes.data_as(ctypes.c_void_p), 64, 64) != 0


def test_recorded_plan_is_replayed_per_chain_and_by_launch_name():
    """ADVICE r4 / round 5, host logic only (no GPU): a chain of a multi-chain engine reads ITS entry of a plan's `per_chain` list, the
    top-level strings otherwise; a chain the plan does not list makes the plan stale; `chains` is never read per chain."""
    from hawq_amd.engine import IntegerEngine, StalePlan
    e = IntegerEngine.__new__(IntegerEngine)
    e.plan = {"chains": 2, "tiles": "1.2.3", "fused_variants": "1.0", "per_chain": [{"tiles": "1.2.3", "fused_variants": "1.0"},
                                                                                   {"tiles": "4.5.6", "fused_variants": ""}]}
    e._plan_on = True
    assert e._fixed("tiles") == "1.2.3" and e._fixed("chains") == "2"          # the parent engine: top-level strings
    e._chain_index = 1
    assert e._fixed("tiles") == "4.5.6"
    assert e._fixed("fused_variants") == "1.0"                                  # an empty per-chain entry falls back to the top level
    assert e._fixed("chains") == "2"
    e._chain_index = 2
    with pytest.raises(StalePlan):
        e._fixed("tiles")
    e._plan_on = False                                                          # plan not applicable to this batch shape: environment or tune
    os.environ.pop("HAWQ_TILES", None)
    assert e._fixed("tiles") is None


def test_profile_tools_on_a_synthetic_kernel_trace(tmp_path):
    """tools/dominant_kernel.py and tools/rocprof_overlap.py (run on the GPU box by tools/profile_round.sh) against a hand-made
    rocprofv3-style `kernels` table: two chains, three forwards, stem + two convs per chain and forward, one tuning kernel."""
    import json
    import sqlite3
    import subprocess
    import sys
    db = tmp_path / "r.db"
    c = sqlite3.connect(db)
    c.execute("create table kernels (name text, start integer, end integer, duration integer, vgpr_count integer, accum_vgpr_count integer, lds_size integer)")
    t = 0
    rows = [("minmax_kernel", 0, 5000, 5000, 20, 0, 0)]
    for f in range(3):
        for ch in range(2):
            s0 = 10_000 + f * 100_000 + ch * 2_000          # the second chain starts 2 us after the first: the kernels overlap
            rows.append(("void stem_fused_kernel<false>(P)", s0, s0 + 10_000, 10_000, 48, 0, 26016))
            rows.append(("void conv3x3_v2_kernel<V2Cfg<2, 2, 2, 4, 256, 3>, 1, 0, false>(B2P)", s0 + 11_000, s0 + 31_000, 20_000, 72, 0, 81920))
            rows.append(("void conv_kernel<Cfg<64, 64, 2, 2, 2, 1, 6, 1>, 1, false, 136, 0, false>(ConvP)", s0 + 32_000, s0 + 37_000, 5_000, 32, 0, 25600))
    c.executemany("insert into kernels values (?,?,?,?,?,?,?)", rows)
    c.commit()
    c.close()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "dominant_kernel.py"), str(db), str(tmp_path / "no_pmc.md"), "w", "abc1234", "2"],
                         capture_output=True, text=True, check=True).stdout
    rec = json.loads(out)["w"]
    assert rec["rocprof_name"].startswith("void conv3x3_v2_kernel<V2Cfg<2, 2, 2, 4, 256, 3>")
    assert rec["launches_per_forward"] == 2.0 and rec["avg_us"] == 20.0 and rec["mfma_busy_frac"] is None and rec["git_head"] == "abc1234"
    assert abs(rec["share_of_forward_kernel_time"] - 20 / 35) < 1e-3 and rec["forward_kernels_in_trace"] == 3      # the tuning kernel is not counted
    ov = subprocess.run([sys.executable, os.path.join(root, "tools", "rocprof_overlap.py"), str(db), "2", "2"], capture_output=True, text=True, check=True).stdout
    assert "last 2 forwards (2 chains): 12 kernels" in ov and "| 2 |" in ov and "| 0 |" in ov


def test_shift_free_identity_requant_equals_the_lifted_form_for_every_uint16():
    """Round 6 (hawq_amd/csrc/common.h: ids0_form / dyadic_s0): a pass-through unit's identity table - the ratio of two residual scales in
    [0.25, 2) - arrives lifted to (e = 33, k >= 1); the QK0 instantiations of the pair kernels apply it as hi32((v << (k - 1)) * m + 2^31)
    without the shift behind the multiply.  Brute force over EVERY uint16 input: that form equals the lifted fast form
    hi32((v << k) * m + 2^32) >> 1, and - where the host's proof excludes ties (the only case the fast kernels run without their tie
    correction) - the reference's round-half-even (quant_utils.py:404-408 restated by the oracle)."""
    import torch
    from oracle import oracle
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    v = np.arange(65536, dtype=np.int64)
    seen_k = set()
    for ratio in (0.2501, 0.26, 0.37, 0.4999, 0.5, 0.63, 0.8, 0.9999, 1.0, 1.3, 1.7, 1.9999):
        m, ek = requant_table(torch.tensor([ratio * 0.7]), torch.ones(1), torch.tensor([0.7]), vbits=17)
        m0, e0, k0 = int(m[0]), int(ek[0]) & 0xff, int(ek[0]) >> 8
        assert e0 == 33 and k0 >= 1, (ratio, e0, k0)          # ids0_form
        seen_k.add(k0)
        lifted = (((v << k0) * m0 + (1 << 32)) >> 32) >> 1      # dyadic_nt: shift, v_mad_i64_i32, v_ashrrev_i32
        s0 = ((v << (k0 - 1)) * m0 + (1 << 31)) >> 32           # dyadic_s0: shift, v_mad_i64_i32
        assert np.array_equal(lifted, s0), ratio
        if tables_are_fast(m, ek, 17):
            assert np.array_equal(s0, oracle.dyadic(v.reshape(-1, 1), np.asarray([m0], np.int64), np.asarray([e0 - k0], np.int32)).reshape(-1)), ratio
    assert seen_k == {1, 2, 3}
    # a ratio below 0.25 needs no lift (k = 0): not the shift-free form - the launchers then pick the general instantiation
    m, ek = requant_table(torch.tensor([0.2 * 0.7]), torch.ones(1), torch.tensor([0.7]), vbits=17)
    assert (int(ek[0]) >> 8) == 0
