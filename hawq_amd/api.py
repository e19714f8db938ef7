"""Convenience entry points mirroring the body of quant_train.py's evaluate path
(quant_train.py:224-299 model build + bit config, :625-674 validate)."""
from __future__ import annotations

import torch

from .bit_schedules import get_bit_config
from .q_resnet import apply_bit_config, quantize_arch_dict
from .quant_modules import freeze_model
from .skeleton import build_float_resnet, init_synthetic


def build_quantized_resnet(arch: str, scheme: str, seed: int | None = 0, float_model=None):
    """Float skeleton (synthetic weights unless ``float_model`` is given) -> Q_ResNet with the
    ``bit_config_<arch>_<scheme>`` schedule applied; eval mode, un-frozen."""
    fl = float_model if float_model is not None else build_float_resnet(arch)
    if float_model is None and seed is not None:
        init_synthetic(fl, seed)
    q = quantize_arch_dict[arch](fl)
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
    return model
