"""TEST INFRASTRUCTURE ONLY - runs the *live* reference (HAWQ) on CPU in the build container.

Imports /root/reference's quant_modules.py / quant_utils.py / q_resnet.py UNMODIFIED and
drives them the way quant_train.py:validate() does (quant_train.py:264-299, 625-674).
Used to (a) pin oracle/hawq_oracle.c + oracle/oracle.py and (b) generate the golden
fixtures under tests/golden/ (tests/golden/make_golden.py).  /root/reference does not
exist on the GPU box: nothing that runs there may import this module.

Recipe:
  * ``utils`` is registered as an empty package shell so utils/__init__.py (torchvision
    imports) is skipped;
  * ``pytorchcv`` is stubbed (imported by q_resnet.py:10-11, unused by the ResNet classes);
  * ``torch.Tensor.cuda`` becomes the identity on CPU-only hosts (quant_utils.py:212-213,
    251, 299 call ``.cuda()`` unconditionally).
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF_ROOT = os.environ.get("HAWQ_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "utils", "quantization_utils", "quant_modules.py"))


_loaded = None


def load_reference():
    """Return (q_resnet module, quant_modules module, quant_utils module) of the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    shell = types.ModuleType("utils")
    shell.__path__ = [os.path.join(REF_ROOT, "utils")]
    sys.modules["utils"] = shell
    for name in ("utils.models", "utils.quantization_utils"):
        sub = types.ModuleType(name)
        sub.__path__ = [os.path.join(REF_ROOT, *name.split("."))]
        sys.modules[name] = sub
    for name, attrs in (
        ("pytorchcv", ()),
        ("pytorchcv.models", ()),
        ("pytorchcv.models.common", ("ConvBlock",)),
        ("pytorchcv.models.shufflenetv2", ("ShuffleUnit", "ShuffleInitBlock")),
    ):
        stub = types.ModuleType(name)
        for a in attrs:
            setattr(stub, a, type(a, (), {}))
        sys.modules[name] = stub
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    import importlib

    qu = importlib.import_module("utils.quantization_utils.quant_utils")
    qm = importlib.import_module("utils.quantization_utils.quant_modules")
    qr = importlib.import_module("utils.models.q_resnet")
    _loaded = (qr, qm, qu)
    return _loaded


def apply_bit_config(model, cfg: dict):
    """The loop of quant_train.py:267-299 with the CLI defaults (quant_train.py:26-152)."""
    matched = 0
    for name, m in model.named_modules():
        if name in cfg:
            matched += 1
            m.quant_mode = "symmetric"
            m.bias_bit = 32
            m.quantize_bias = True
            m.per_channel = True
            m.act_percentile = 0
            m.act_range_momentum = 0.99
            m.weight_percentile = 0
            m.fix_flag = False
            m.fix_BN = True
            m.fix_BN_threshold = None
            m.training_BN_mode = True
            m.checkpoint_iter_threshold = -1
            m.save_path = ""
            m.fixed_point_quantization = False
            bits = cfg[name]
            if hasattr(m, "activation_bit"):
                m.activation_bit = bits
                if bits == 4:
                    m.quant_mode = "asymmetric"
            else:
                m.weight_bit = bits
    assert matched == len(cfg), (matched, len(cfg))
    return model


def build_reference_model(arch: str, scheme: str, seed: int = 0):
    """Float skeleton (seeded) -> reference Q_ResNet with the schedule applied, eval mode."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hawq_amd.bit_schedules import get_bit_config
    from hawq_amd.skeleton import build_float_resnet, init_synthetic

    qr, qm, _ = load_reference()
    if arch == "mobilenetv2_w1":   # utils/models/q_mobilenetv2.py, unmodified (quant_train.py:158)
        import importlib
        from hawq_amd.skeleton import build_float_mobilenetv2
        qmb = importlib.import_module("utils.models.q_mobilenetv2")
        q = qmb.q_mobilenetv2_w1(init_synthetic(build_float_mobilenetv2(), seed))
        cfg = {k: (v[0] if isinstance(v, tuple) else v) for k, v in get_bit_config(arch, scheme).items()}
        apply_bit_config(q, cfg)
        q.eval()
        return q
    fl = init_synthetic(build_float_resnet(arch), seed)
    # quant_train.py:155-158 (quantize_arch_dict): resnet50b shares q_resnet50
    q = {"resnet18": qr.q_resnet18, "resnet50": qr.q_resnet50, "resnet50b": qr.q_resnet50,
         "resnet101": qr.q_resnet101}[arch](fl)
    apply_bit_config(q, get_bit_config(arch, scheme))
    q.eval()
    return q


class ConvTap:
    """Capture the raw fp32 outputs of F.conv2d / F.linear inside the reference modules
    (= the int32 accumulators incl. bias, before the scale multiply; quant_modules.py:130, 493)."""

    def __init__(self, qm):
        self.qm = qm
        self.conv, self.linear = [], []

    def __enter__(self):
        F = self.qm.F
        self._c, self._l = F.conv2d, F.linear

        def conv2d(*a, **k):
            y = self._c(*a, **k)
            self.conv.append(y.detach().clone())
            return y

        def linear(*a, **k):
            y = self._l(*a, **k)
            self.linear.append(y.detach().clone())
            return y

        F.conv2d, F.linear = conv2d, linear
        return self

    def __exit__(self, *exc):
        self.qm.F.conv2d, self.qm.F.linear = self._c, self._l


def calibrate_and_freeze(q, x_cal):
    """One un-frozen forward (ranges initialise to batch min/max, quant_modules.py:247-250),
    then freeze_model (quant_modules.py:739-758) as validate() does (quant_train.py:636)."""
    _, qm, _ = load_reference()
    with torch.no_grad():
        q(x_cal)
    qm.freeze_model(q)
    q.eval()
    return q


def forward_with_taps(q, x):
    _, qm, _ = load_reference()
    with torch.no_grad(), ConvTap(qm) as tap:
        y = q(x)
    return y, tap.conv, tap.linear
