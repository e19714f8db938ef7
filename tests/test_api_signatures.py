"""Drop-in boundary pinned at SIGNATURE level.  The reference has no FFI: its operator
boundary is the Python nn.Module API of utils/quantization_utils/quant_modules.py (:34-42, 157-165, 205-206, 334-343, 358,
571-574, 627-634, 652).  Build container only (needs /root/reference):

  * inspect.signature of __init__ / forward / set_param / fix / unfix of every class hawq_amd.quant_modules exports equals the
    reference's (names, order, kinds, defaults); freeze_model / unfreeze_model likewise;
  * the reference's OWN graph code (utils/models/q_resnet.py, unmodified) is executed over hawq_amd.quant_modules (the import
    `from ..quantization_utils.quant_modules import *` resolved to the drop-in): its Q_ResNet18 / Q_ResNet50 constructors must
    run, and the resulting module tree (names + class names) and state_dict keys / shapes must equal those of the same graph
    over the reference's modules AND those of hawq_amd.q_resnet - i.e. INTEGRATION.md option A (swap one import) really is a
    drop-in with the same checkpoint key layout.
"""
import importlib.util
import inspect
import os
import sys

import pytest

pytestmark = pytest.mark.reference

CLASSES = ("QuantLinear", "QuantAct", "QuantBnConv2d", "QuantMaxPool2d", "QuantDropout", "QuantAveragePool2d", "QuantConv2d")
METHODS = ("__init__", "forward", "set_param", "fix", "unfix")


def _sig(fn):
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


def test_module_signatures_equal_the_reference():
    from hawq_amd import quant_modules as ours
    from oracle import ref_live
    _, qm, _ = ref_live.load_reference()
    checked = 0
    for cname in CLASSES:
        ref_cls, our_cls = getattr(qm, cname), getattr(ours, cname)
        for meth in METHODS:
            if not hasattr(ref_cls, meth):
                assert meth in ("set_param", "fix", "unfix") and not hasattr(our_cls, meth) or hasattr(our_cls, meth), (cname, meth)
                continue
            assert hasattr(our_cls, meth), f"{cname}.{meth} missing"
            assert _sig(getattr(our_cls, meth)) == _sig(getattr(ref_cls, meth)), \
                f"{cname}.{meth}: {inspect.signature(getattr(our_cls, meth))} vs reference {inspect.signature(getattr(ref_cls, meth))}"
            checked += 1
    assert checked >= 24
    for fn in ("freeze_model", "unfreeze_model"):
        assert _sig(getattr(ours, fn)) == _sig(getattr(qm, fn))


def _reference_graph_over(modules_pkg, fname="q_resnet.py"):
    """utils/models/<fname> of the reference, unmodified, with `..quantization_utils.quant_modules` resolved to `modules_pkg`."""
    from oracle import ref_live
    ref_live.load_reference()   # registers the `utils` package shell and the pytorchcv stubs
    key = "utils.quantization_utils.quant_modules"
    saved = sys.modules[key]
    sys.modules[key] = modules_pkg
    try:
        spec = importlib.util.spec_from_file_location("utils.models._" + fname[:-3] + "_over_" + modules_pkg.__name__.replace(".", "_"),
                                                      os.path.join(ref_live.REF_ROOT, "utils", "models", fname))
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "utils.models"
        spec.loader.exec_module(mod)
    finally:
        sys.modules[key] = saved
    return mod


@pytest.mark.parametrize("arch,cls", [("resnet18", "Q_ResNet18"), ("resnet50", "Q_ResNet50")])
def test_the_references_own_graph_runs_over_the_drop_in_modules(arch, cls):
    from hawq_amd import q_resnet as our_graph
    from hawq_amd import quant_modules as ours
    from hawq_amd.skeleton import build_float_resnet, init_synthetic
    from oracle import ref_live
    _, qm, _ = ref_live.load_reference()

    def skeleton():
        fl = build_float_resnet(arch)
        init_synthetic(fl, 0)
        return fl

    over_ours = getattr(_reference_graph_over(ours), cls)(skeleton())      # reference graph code, drop-in modules
    over_ref = getattr(_reference_graph_over(qm), cls)(skeleton())         # reference graph code, reference modules
    native = our_graph.quantize_arch_dict[arch](skeleton())                 # hawq_amd's own graph builder

    def tree(m):
        return [(n, type(x).__name__) for n, x in m.named_modules() if n]

    def keys(m):
        return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]

    assert all(type(x).__module__ == "hawq_amd.quant_modules" for _, x in over_ours.named_modules()
               if type(x).__name__.startswith("Quant"))
    assert tree(over_ours) == tree(over_ref)
    assert keys(over_ours) == keys(over_ref)
    # hawq_amd's builder produces the same quantized leaves under the same names (its unit container class is its own)
    leaves = lambda m: [(n, c) for n, c in tree(m) if c.startswith("Quant")]
    assert leaves(native) == leaves(over_ref)
    assert keys(native) == keys(over_ref)
    # (nn.Module.load_state_dict cannot address the dotted child names "stage1.unit1" that the reference registers with setattr -
    # neither its classes nor this mirror strict-load through torch; checkpoints go through hawq_amd.api.load_checkpoint /
    # load_quantized_checkpoint, pinned to files the live reference writes in tests/test_checkpoint.py)


def test_mobilenetv2_builder_mirrors_the_references_graph():
    """hawq_amd.q_mobilenetv2 walks the float network instead of spelling the graph out (it is not a transcription of
    q_mobilenetv2.py); what it builds must still be the reference's network: same quantized leaves under the same names in the
    same registration order (= named_modules order the fixtures and bit schedules rely on), same state_dict keys and shapes,
    same residual / expansion decisions per unit."""
    from hawq_amd import quant_modules as ours
    from hawq_amd.q_mobilenetv2 import q_mobilenetv2_w1
    from hawq_amd.skeleton import build_float_mobilenetv2, init_synthetic

    def skeleton():
        return init_synthetic(build_float_mobilenetv2(), 0)

    ref_graph = _reference_graph_over(ours, "q_mobilenetv2.py").q_mobilenetv2_w1(skeleton())
    native = q_mobilenetv2_w1(skeleton())
    leaves = lambda m: [(n, type(x).__name__) for n, x in m.named_modules() if type(x).__name__.startswith("Quant")]
    assert leaves(native) == leaves(ref_graph) and len(leaves(native)) == 3 + 17 * 7 + 6
    assert [(k, tuple(v.shape)) for k, v in native.state_dict().items()] == [(k, tuple(v.shape)) for k, v in ref_graph.state_dict().items()]
    ref_units = [u for st in ref_graph.features.children() if isinstance(st, type(ref_graph.features)) for u in st.children()]
    assert [u.residual for u in native.units()] == [u.residual for u in ref_units] and len(ref_units) == 17
    assert native.channels == ref_graph.channels
