"""Convenience entry points mirroring the body of quant_train.py's evaluate path
(quant_train.py:224-299 model build + bit config, :625-674 validate)."""
from __future__ import annotations

import torch

from .bit_schedules import get_bit_config
from .q_resnet import apply_bit_config, quantize_arch_dict
from .quant_modules import freeze_model, trust_integer_buffers
from .skeleton import build_float_resnet, init_synthetic


def build_quantized_resnet(arch: str, scheme: str, seed: int | None = 0, float_model=None):
    """Float skeleton (synthetic weights unless ``float_model`` is given) -> Q_ResNet (or, ``mobilenetv2_w1``,
    Q_MobileNetV2) with the ``bit_config_<arch>_<scheme>`` schedule applied; eval mode, un-frozen."""
    if arch == "mobilenetv2_w1":
        from .q_mobilenetv2 import q_mobilenetv2_w1
        from .skeleton import build_float_mobilenetv2
        build, quantize = build_float_mobilenetv2, q_mobilenetv2_w1
    else:
        build, quantize = (lambda: build_float_resnet(arch)), quantize_arch_dict[arch]
    fl = float_model if float_model is not None else build()
    if float_model is None and seed is not None:
        init_synthetic(fl, seed)
    q = quantize(fl)
    apply_bit_config(q, get_bit_config(arch, scheme))
    q.eval()
    return q


def calibrate(model, images: torch.Tensor):
    """One un-frozen forward to initialise every QuantAct range from ``images`` (the reference
    gets its ranges from QAT checkpoints; with synthetic weights one batch plays that role),
    then freeze_model as validate() does (quant_train.py:636)."""
    model.eval()
    with torch.no_grad():
        model.forward_modules(images)
    freeze_model(model)
    model.invalidate_engine()
    # ranges and integer buffers now come from the float parameters again, not from a quantized checkpoint
    trust_integer_buffers(model, False)
    model.engine_defaults = dict(getattr(model, "engine_defaults", {}), from_buffers=False)
    return model


# ------------------------------------------------------------------ quantized checkpoints (quant_train.py:665-670)
_QCKPT_GROUPS = ("convbn_scaling_factor", "fc_scaling_factor", "weight_integer", "bias_integer", "act_scaling_factor")
# The reference's five groups omit QuantConv2d's own scale buffer (MobileNetV2's classifier, quant_modules.py:627-634): a file it
# writes cannot restore that network by itself.  Ours adds the group when the model has such buffers (the reference's loader
# ignores unknown groups); loading a file without it into such a model is refused instead of running on a placeholder scale.
_QCKPT_EXTRA = ("conv_scaling_factor",)


def save_quantized_checkpoint(model, path):
    """Write ``quantized_checkpoint.pth.tar`` exactly as the reference's ``validate()`` does
    (quant_train.py:665-670): five dicts of the frozen model's integer weights / biases and scales, keyed by
    the ``state_dict`` names.  Call after a frozen forward (the buffers are filled by it)."""
    sd = model.state_dict()
    out = {g: {k: v.detach().cpu() for k, v in sd.items() if g in k} for g in _QCKPT_GROUPS}
    for g in _QCKPT_EXTRA:
        extra = {k: v.detach().cpu() for k, v in sd.items() if k.rpartition(".")[2] == g}
        if extra:
            out[g] = extra
    torch.save(out, path)


def load_quantized_checkpoint(model, ckpt, strict: bool = True):
    """Load a reference ``quantized_checkpoint.pth.tar`` (path or the already-loaded dict) into ``model`` (a
    ``Q_ResNet*`` with the matching bit configuration applied), freeze it and make its fused engine trust the
    integer buffers (``IntegerEngine(from_buffers=True)``): no float weights, BN statistics or calibration are
    needed.  Returns the model.  A file that LACKS a tensor the integer path needs is always refused (running on placeholder
    buffers would return garbage logits silently); ``strict=False`` only tolerates EXTRA entries the model has no use for."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    missing = [g for g in _QCKPT_GROUPS if g not in ckpt]
    if missing:
        raise KeyError(f"not a HAWQ quantized checkpoint: missing {missing}")
    flat = {}
    for g in _QCKPT_GROUPS + tuple(g for g in _QCKPT_EXTRA if g in ckpt):
        # validate() saves the state_dict of the DataParallel-wrapped model (quant_train.py:358, 665-670): real HAWQ
        # files carry 'module.'-prefixed keys; the TVM loader strips them the same way (hawq_utils_resnet50.py:479-485)
        flat.update({(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt[g].items()})
    own = dict(model.named_buffers())
    own.update(dict(model.named_parameters()))
    unexpected = [k for k in flat if k not in own]
    wanted = [k for k in own if any(g in k for g in _QCKPT_GROUPS) or k.rpartition(".")[2] in _QCKPT_EXTRA]
    absent = [k for k in wanted if k not in flat]
    if strict and (unexpected or absent):
        raise KeyError(f"quantized checkpoint does not match the model: unexpected {unexpected[:3]}..., missing {absent[:3]}...")
    if absent:
        # an engine that trusts integer buffers must find ALL of them: running on placeholder buffers would return
        # garbage logits without any error
        raise KeyError(f"quantized checkpoint lacks {len(absent)} of the {len(wanted)} tensors the integer engine needs "
                       f"(first: {absent[:3]}); nothing was loaded")
    with torch.no_grad():
        for k, v in flat.items():
            if k not in own:
                continue
            dst = own[k]
            v = v.to(device=dst.device, dtype=dst.dtype)
            if dst.shape == v.shape:
                dst.copy_(v)
            else:  # buffers are registered with placeholder shapes until the first frozen forward fills them
                mod_name, _, attr = k.rpartition(".")
                setattr(model.get_submodule(mod_name), attr, v.clone())
    freeze_model(model)
    model.eval()
    model.invalidate_engine()
    # both consumers of the loaded integers: the per-module path (forward_modules, Q_MobileNetV2) and the fused engine
    trust_integer_buffers(model, True)
    model.engine_defaults = dict(getattr(model, "engine_defaults", {}), from_buffers=True)
    return model


# ------------------------------------------------------------------ float QAT checkpoints (quant_train.py:303-314)
def load_checkpoint(model, ckpt, freeze: bool = True):
    """Load a HAWQ ``checkpoint.pth.tar`` / ``model_best.pth.tar`` (path or loaded dict) the way
    ``quant_train.py --resume --resume-quantize`` does (quant_train.py:303-314): take ``['state_dict']``, drop
    ``num_batches_tracked`` / ``weight_integer`` / ``bias_integer`` entries, strip the DataParallel ``module.`` prefix,
    ``load_state_dict(strict=False)``.  The checkpoint carries float weights, BN statistics and the activation ranges
    (``x_min`` / ``x_max``) of QAT, so no calibration is needed; with ``freeze`` the model is frozen for inference as
    ``validate()`` does (quant_train.py:636).  Returns (model, missing_keys, unexpected_keys)."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    kept = {}
    for key, value in sd.items():
        if "num_batches_tracked" in key or "weight_integer" in key or "bias_integer" in key:
            continue
        kept[key.replace("module.", "")] = value
    own = model.state_dict()
    # buffers whose placeholder shape differs (filled by the first frozen forward) cannot go through load_state_dict
    direct = {k: v for k, v in kept.items() if k in own and own[k].shape == v.shape}
    result = model.load_state_dict(direct, strict=False)
    with torch.no_grad():
        for k, v in kept.items():
            if k in own and own[k].shape != v.shape:
                mod_name, _, attr = k.rpartition(".")
                setattr(model.get_submodule(mod_name), attr, v.to(device=own[k].device, dtype=own[k].dtype).clone())
    unexpected = [k for k in kept if k not in own]
    missing = [k for k in result.missing_keys
               if not any(t in k for t in ("num_batches_tracked", "weight_integer", "bias_integer"))
               and k not in kept]
    if freeze:
        freeze_model(model)
        model.eval()
    model.invalidate_engine()
    # the integer buffers are stale now (quant_train.py:309-316 drops them too): re-derive from the float parameters
    trust_integer_buffers(model, False)
    model.engine_defaults = dict(getattr(model, "engine_defaults", {}), from_buffers=False)
    return model, missing, unexpected


# ------------------------------------------------------------------ evaluation loop (quant_train.py:625-674)
def validate(model, loader, uint8: bool = False, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), device="cuda"):
    """Top-1 / top-5 accuracy (percent) of a frozen ``Q_ResNet*`` over ``loader`` = iterable of (images, target):
    the body of the reference's ``validate()`` (freeze, eval, no_grad, ``accuracy(output, target, topk=(1, 5))``,
    sample-weighted averages).  ``images`` are normalised fp32 NCHW batches as the reference's pipeline produces, or -
    ``uint8`` - raw uint8 NHWC batches that go through the look-up-table input quantiser (``forward_uint8``);
    ``hawq_amd.image.preprocess_batch`` turns decoded images of any size into such batches (Resize(256) + CenterCrop(224)).
    Returns (top1, top5, n_images)."""
    freeze_model(model)
    model.eval()
    top1 = top5 = 0.0
    n = 0
    with torch.no_grad():
        for images, target in loader:
            images = images.to(device, non_blocking=True)
            target = target.to(device, non_blocking=True)
            if uint8 and not hasattr(model, "engine"):
                raise NotImplementedError(f"uint8 image input needs the fused ResNet engine; {type(model).__name__} has none")
            output = model.engine().forward_uint8(images, mean, std) if uint8 else model(images)
            _, pred = output.topk(5, 1, True, True)
            correct = pred.t().eq(target.view(1, -1).expand(5, -1))
            top1 += float(correct[:1].reshape(-1).float().sum())
            top5 += float(correct[:5].reshape(-1).float().sum())
            n += int(target.size(0))
    return 100.0 * top1 / max(n, 1), 100.0 * top5 / max(n, 1), n


build_quantized_model = build_quantized_resnet   # the name says ResNet for history; MobileNetV2 goes through it too
