"""Generate the golden fixtures in this directory from the LIVE reference (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Everything here is produced by executing /root/reference's own quant_utils.py /
quant_modules.py / q_resnet.py unmodified (oracle/ref_live.py); nothing is computed by the
oracle or by hawq_amd.  The fixtures travel with the repo because /root/reference does
not exist on the GPU box.

Files
  kat_functions.npz      known-answer vectors for the L1 functions (quant_utils.py)
  kat_modules.npz        small QuantBnConv2d / QuantLinear / QuantAveragePool2d / QuantAct cases
  net_<arch>_<scheme>_b<B>.npz   whole-network fixtures: frozen ranges, reference integer
                         buffers (scales + bias in full, weight_integer as digests plus the
                         few entries where torch-CPU's non-IEEE sqrt moved a weight), per-layer
                         accumulator digests, logits.
Inputs/weights are regenerated from seeds (hawq_amd.skeleton); their SHA-256 is stored so a
different RNG stream is detected rather than mis-reported as a parity failure.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_live  # noqa: E402
from hawq_amd.skeleton import synthetic_images  # noqa: E402


def digest(a) -> np.ndarray:
    """Order-sensitive 3-word digest of an integer tensor in its given (NCHW) order."""
    a = np.ascontiguousarray(a).astype(np.int64).reshape(-1)
    w = (np.arange(a.size, dtype=np.int64) % 8191) + 1
    with np.errstate(over="ignore"):
        return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.int64)


def sha(t) -> str:
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def kat_functions(qu):
    out = {}
    g = torch.Generator().manual_seed(7)
    # batch_frexp (quant_utils.py:188-213)
    r = torch.cat([torch.rand(200, generator=g).double() * 10 ** torch.randint(-8, 3, (200,), generator=g).double(),
                   torch.tensor([0.5, 0.25, 1.0, 2.0 ** -20, 0.9999999999999999, 0.75, 1.5, 3.0])])
    m, e = qu.batch_frexp(r)
    out["frexp_r"], out["frexp_m"], out["frexp_e"] = r.numpy(), m.numpy().astype(np.int64), e.numpy().astype(np.int32)

    # linear quantisation (quant_utils.py:73-97, 237-258, 281-308)
    x = torch.randn(4, 6, 5, 5, generator=g) * 3
    x.view(-1)[:8] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 1e9, -1e9])  # ties at scale 1
    for tag, s in (("a", torch.tensor([1.0])), ("b", torch.tensor([0.0371])), ("c", torch.tensor([2.9e-3]))):
        out[f"symq_{tag}_scale"] = s.numpy()
        out[f"symq_{tag}_8"] = qu.SymmetricQuantFunction.apply(x, 8, s).numpy()
        out[f"symq_{tag}_4"] = qu.SymmetricQuantFunction.apply(x, 4, s).numpy()
        out[f"asymq_{tag}_4"] = qu.AsymmetricQuantFunction.apply(x, 4, s).numpy()
    out["q_x"] = x.numpy()
    # per-channel weight style (scale per dim-0)
    sw = torch.rand(4, generator=g) * 0.05 + 1e-3
    out["symq_pc_scale"] = sw.numpy()
    out["symq_pc_8"] = qu.SymmetricQuantFunction.apply(x, 8, sw).numpy()
    out["symq_pc_4"] = qu.SymmetricQuantFunction.apply(x, 4, sw).numpy()

    # scale formulas (quant_utils.py:128-185)
    lo = -torch.rand(16, generator=g) * 4
    hi = torch.rand(16, generator=g) * 4
    lo[0], hi[0] = 0.0, 0.0
    out["rng_lo"], out["rng_hi"] = lo.numpy(), hi.numpy()
    for b in (4, 8, 16):
        out[f"sym_scale_pc_{b}"] = qu.symmetric_linear_quantization_params(b, lo, hi, True).numpy()
        out[f"sym_scale_{b}"] = np.stack([qu.symmetric_linear_quantization_params(b, lo[i], hi[i], False).numpy()
                                          for i in range(16)])
        out[f"asym_scale_{b}"] = np.stack([qu.asymmetric_linear_quantization_params(b, lo[i], hi[i], True)[0].numpy()
                                           for i in range(16)])

    # fixedpoint_fn (quant_utils.py:363-456)
    C = 6
    acc = torch.randint(-200000, 200000, (3, C, 4, 4), generator=g).float()
    acc[0, 0].view(-1)[:6] = torch.tensor([2.0, 6.0, -2.0, -6.0, 10.0, 1.0])  # ties for ratio 1/4
    idn = torch.randint(0, 40000, (3, C, 4, 4), generator=g).float()
    cases = {
        "rand8": (8, "symmetric", torch.tensor([0.0213]), torch.rand(C, generator=g) * 1e-3 + 1e-4, torch.tensor([0.047])),
        "rand4": (4, "asymmetric", torch.tensor([0.0213]), torch.rand(C, generator=g) * 1e-3 + 1e-4, torch.tensor([0.21])),
        "rand16": (16, "symmetric", torch.tensor([0.0213]), torch.rand(C, generator=g) * 1e-3 + 1e-4, torch.tensor([2.4e-4])),
        "tie8": (8, "symmetric", torch.tensor([0.5]), torch.full((C,), 0.25), torch.tensor([0.5])),
        "tie16": (16, "symmetric", torch.tensor([0.5]), torch.full((C,), 0.25), torch.tensor([0.5])),
        "up8": (8, "symmetric", torch.tensor([0.5]), torch.full((C,), 4.0), torch.tensor([0.5])),  # ratio 4 (e<31)
    }
    for tag, (bits, mode, s_a, s_w, s_out) in cases.items():
        z = acc * (s_a * s_w).view(1, -1, 1, 1)  # what QuantBnConv2d hands over
        y0 = qu.fixedpoint_fn.apply(z, bits, mode, s_out, 0, s_a, s_w)
        out[f"fp0_{tag}_z"], out[f"fp0_{tag}_y"] = z.numpy(), y0.numpy()
        out[f"fp0_{tag}_sa"], out[f"fp0_{tag}_sw"], out[f"fp0_{tag}_sout"] = s_a.numpy(), s_w.numpy(), s_out.numpy()
        out[f"fp0_{tag}_bits"] = np.array([bits, mode == "symmetric"])
        # case 1: identity passthrough (weight scale 1) and identity conv (per-channel)
        s_ida = torch.tensor([3.1e-4])
        for itag, s_idw in (("pass", torch.ones(1)), ("conv", torch.rand(C, generator=g) * 1e-3 + 1e-4)):
            ident = idn * (s_ida * s_idw).view(1, -1, 1, 1)
            y1 = qu.fixedpoint_fn.apply(z + ident, bits, mode, s_out, 1, s_a, s_w, ident, s_ida, s_idw)
            k = f"fp1_{tag}_{itag}"
            out[k + "_ident"], out[k + "_y"], out[k + "_sida"], out[k + "_sidw"] = (
                ident.numpy(), y1.numpy(), s_ida.numpy(), s_idw.numpy())
    out["fp_acc"], out["fp_idn"] = acc.numpy(), idn.numpy()

    # int averaging (quant_utils.py:334-337)
    v = torch.randint(0, 40000, (2, 8, 7, 7), generator=g).float()
    v[0, 0] = 3.0  # exact multiple
    v[0, 1].view(-1)[:48] = 5.0
    v[0, 1].view(-1)[48] = 4.0  # 49k+48 -> k
    out["avg_x"] = v.numpy()
    out["avg_y"] = qu.transfer_float_averaging_to_int_averaging.apply(torch.nn.AvgPool2d(7, 1)(v)).numpy()
    return out


def kat_modules(qm):
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, (cin, cout, k, stride, pad, bits, hw) in {
        "c3": (16, 8, 3, 1, 1, 8, 9), "c1s2": (32, 16, 1, 2, 0, 4, 8), "c3s2": (16, 8, 3, 2, 1, 8, 9),
        "c7": (3, 8, 7, 2, 3, 8, 20)}.items():
        conv = torch.nn.Conv2d(cin, cout, k, stride, pad, bias=False)
        bn = torch.nn.BatchNorm2d(cout)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        m = qm.QuantBnConv2d(weight_bit=bits, bias_bit=32, per_channel=True, fix_BN=True)
        m.set_param(conv, bn)
        m.fix()
        m.eval()
        s_a = torch.tensor([0.0173])
        lo, hi = ((-128, 127) if bits == 8 else (0, 15))
        q = torch.randint(lo, hi + 1, (2, cin, hw, hw), generator=g).float()
        with torch.no_grad():
            y, s_w = m((q * s_a, s_a))
        for kx, v in dict(w=conv.weight, gamma=bn.weight, beta=bn.bias, mean=bn.running_mean, var=bn.running_var,
                          q=q, s_a=s_a, y=y, s_w=s_w, weight_integer=m.weight_integer, bias_integer=m.bias_integer
                          ).items():
            out[f"conv_{tag}_{kx}"] = v.detach().numpy()
        out[f"conv_{tag}_cfg"] = np.array([cin, cout, k, stride, pad, bits, hw])
    # QuantLinear (quant_modules.py:79-130)
    lin = torch.nn.Linear(64, 10)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(10, 64, generator=g) * 0.05)
        lin.bias.copy_(torch.randn(10, generator=g) * 0.1)
    m = qm.QuantLinear(weight_bit=8, bias_bit=32, per_channel=True)
    m.set_param(lin)
    s_a = torch.tensor([0.031])
    q = torch.randint(-128, 128, (3, 64), generator=g).float()
    with torch.no_grad():
        y = m(q * s_a, s_a)
    for kx, v in dict(w=lin.weight, b=lin.bias, q=q, s_a=s_a, y=y, fc_scaling_factor=m.fc_scaling_factor,
                      weight_integer=m.weight_integer, bias_integer=m.bias_integer).items():
        out[f"lin_{kx}"] = v.detach().numpy()
    # QuantAveragePool2d (quant_modules.py:585-602)
    m = qm.QuantAveragePool2d(7, 1)
    s = torch.tensor([2.7e-4])
    q = torch.randint(0, 40000, (2, 8, 7, 7), generator=g).float()
    with torch.no_grad():
        y, _ = m(q * s, s)
    out["pool_q"], out["pool_s"], out["pool_y"] = q.numpy(), s.numpy(), y.numpy()
    # QuantAct input case (quant_modules.py:205-274), frozen range
    a = qm.QuantAct(activation_bit=8)
    a.x_min += -2.31
    a.x_max += 2.64
    a.fix()
    x = torch.randn(2, 3, 8, 8, generator=g)
    with torch.no_grad():
        y, s = a(x)
    out["act_in_x"], out["act_in_y"], out["act_in_s"] = x.numpy(), y.numpy(), s.numpy()
    out["act_in_rng"] = np.array([a.x_min.item(), a.x_max.item()], np.float32)
    return out


def net_fixture(arch, scheme, batch, image=None, light=False, calib=None):
    """`calib`: images the ranges are calibrated on (default: the evaluated batch itself, as in the b2 fixtures)."""
    from oracle import oracle  # only used for its IEEE host-prep, to express weight_integer as patches

    qr, qm, qu = ref_live.load_reference()
    q = ref_live.build_reference_model(arch, scheme, seed=0)
    x = synthetic_images(batch, seed=0) if image is None else image
    out = {"input_sha": np.array(sha(x.numpy())), "torch_version": np.array(torch.__version__)}
    h = hashlib.sha256()
    for p in q.state_dict().values():
        h.update(np.ascontiguousarray(p.numpy()).tobytes())
    out["weights_sha"] = np.array(h.hexdigest())
    ref_live.calibrate_and_freeze(q, x if calib is None else calib)
    y, convs, lins = ref_live.forward_with_taps(q, x)
    out["logits"] = y.numpy()
    out["top1"] = y.argmax(1).numpy()

    acts, conv_mods = [], []
    for name, m in q.named_modules():
        t = type(m).__name__
        if t == "QuantAct":
            acts.append((name, m))
        elif t == "QuantBnConv2d":
            conv_mods.append((name, m))
    out["act_names"] = np.array([n for n, _ in acts])
    out["act_x_min"] = np.array([m.x_min.item() for _, m in acts], np.float32)
    out["act_x_max"] = np.array([m.x_max.item() for _, m in acts], np.float32)
    out["act_scale"] = np.array([m.act_scaling_factor.item() for _, m in acts], np.float32)

    # conv call order of one forward = order of the taps; map module -> tap via forward hooks
    order = []
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: order.append(n)) for n, m in conv_mods]
    with torch.no_grad():
        q(x)
    for hk in hooks:
        hk.remove()
    assert len(order) == len(convs)
    tap = dict(zip(order, convs))
    st = oracle.extract_float_state(q)
    cbs = {"quant_init_convbn": st["stem"], "quant_init_block_convbn": st["stem"]}
    for u in st["units"]:
        for k_mod, k_st in (("quant_convbn1", "convbn1"), ("quant_convbn2", "convbn2"), ("quant_convbn3", "convbn3"),
                            ("quant_identity_convbn", "identity")):
            if k_st in u:
                cbs[u["name"] + "." + k_mod] = u[k_st]
    out["conv_names"] = np.array([n for n, _ in conv_mods])
    scales, biases, wdig, accdig, patches = [], [], [], [], []
    accmax = 0
    for li, (name, m) in enumerate(conv_mods):
        scales.append(m.convbn_scaling_factor.numpy())
        biases.append(m.bias_integer.numpy().astype(np.int64))
        wi = m.weight_integer.numpy().astype(np.int64)
        wdig.append(digest(wi))
        acc = np.rint(tap[name].numpy().astype(np.float64)).astype(np.int64)
        accdig.append(digest(acc))
        accmax = max(accmax, int(np.abs(acc).max()))
        cb = cbs[name]
        w_f, _ = oracle.fold_bn(cb["w"], cb["gamma"], cb["beta"], cb["mean"], cb["var"], cb["eps"])
        w_ieee, _ = oracle.quantize_weight(w_f, cb["bits"])
        for idx in np.argwhere(wi.reshape(-1) != w_ieee.reshape(-1)).reshape(-1):
            patches.append((li, int(idx), int(wi.reshape(-1)[idx])))
        if not light and name in ("stage4.unit2.quant_convbn1", "stage1.unit1.quant_convbn1"):
            out["acc_full." + name] = acc.astype(np.int32)
    out["conv_scale"] = np.concatenate(scales).astype(np.float32)
    out["conv_bias"] = np.concatenate(biases)
    out["conv_wdigest"] = np.stack(wdig)
    out["conv_accdigest"] = np.stack(accdig)
    out["conv_wpatch"] = np.array(patches, np.int64).reshape(-1, 3)
    out["acc_absmax"] = np.array(accmax)
    fc = q.quant_output
    out["fc_scale"] = fc.fc_scaling_factor.numpy()
    out["fc_bias"] = fc.bias_integer.numpy().astype(np.int64)
    out["fc_wdigest"] = digest(fc.weight_integer.numpy())
    out["fc_acc"] = np.rint(lins[0].numpy().astype(np.float64)).astype(np.int64)
    return out


EXTRA = (("resnet101", "uniform8", 2), ("resnet50b", "uniform4", 2), ("resnet50", "latency_0.5", 2),
         ("resnet50", "modelsize_0.25", 2))


def main():
    qr, qm, qu = ref_live.load_reference()
    if "--extra-only" in sys.argv:
        for arch, scheme, b in EXTRA:
            fx = net_fixture(arch, scheme, b, light=True)
            np.savez_compressed(os.path.join(HERE, f"net_{arch}_{scheme}_b{b}.npz"), **fx)
            print(arch, scheme, "acc_absmax", int(fx["acc_absmax"]), "wpatches", len(fx["conv_wpatch"]), flush=True)
        return
    np.savez_compressed(os.path.join(HERE, "kat_functions.npz"), **kat_functions(qu))
    np.savez_compressed(os.path.join(HERE, "kat_modules.npz"), **kat_modules(qm))
    for arch, scheme, b in (("resnet18", "uniform8", 2), ("resnet18", "uniform4", 2), ("resnet18", "bops_0.5", 2),
                            ("resnet50", "uniform8", 2), ("resnet50", "uniform4", 2), ("resnet50", "bops_0.5", 2)):
        fx = net_fixture(arch, scheme, b)
        np.savez_compressed(os.path.join(HERE, f"net_{arch}_{scheme}_b{b}.npz"), **fx)
        print(arch, scheme, "acc_absmax", int(fx["acc_absmax"]), "wpatches", len(fx["conv_wpatch"]), flush=True)
    # further shipped graphs / schedules (ResNet101, the stride-on-3x3 ResNet50b, mixed-width schedules): "light"
    # fixtures without the two full accumulator tensors
    for arch, scheme, b in EXTRA:
        fx = net_fixture(arch, scheme, b, light=True)
        np.savez_compressed(os.path.join(HERE, f"net_{arch}_{scheme}_b{b}.npz"), **fx)
        print(arch, scheme, "acc_absmax", int(fx["acc_absmax"]), "wpatches", len(fx["conv_wpatch"]), flush=True)
    # the one real-image fixture the reference ships (tvm_benchmark/models/input_image_batch_1.npy, NHWC)
    img = np.load(os.path.join(ref_live.REF_ROOT, "tvm_benchmark", "models", "input_image_batch_1.npy"))
    img = torch.from_numpy(np.ascontiguousarray(img.transpose(0, 3, 1, 2))).float()
    np.save(os.path.join(HERE, "real_image_nchw.npy"), img.numpy())
    fx = net_fixture("resnet18", "uniform8", 1, image=img)
    np.savez_compressed(os.path.join(HERE, "net_resnet18_uniform8_realimg.npz"), **fx)


if __name__ == "__main__":
    main()
