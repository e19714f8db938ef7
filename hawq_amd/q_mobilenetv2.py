"""Quantised MobileNetV2 over the drop-in modules: a table-driven builder, not a transcription of the reference's file.

The reference (utils/models/q_mobilenetv2.py:12-209) spells the network out class by class; with INTEGRATION.md option A its
own file runs unchanged on ``hawq_amd.quant_modules``.  This module exists for option B (no reference tree around): it WALKS
the float network it is given - ``features.init_block``, every ``features.stageN.unitM`` with whatever of ``conv1`` (1x1
expansion), ``conv2`` (3x3 depthwise), ``conv3`` (1x1 linear projection) the unit has, ``features.final_block``,
``features.final_pool``, ``output`` - and mirrors each float layer with the quantized module of the same role under the
reference's attribute names, in the reference's registration order, so that ``state_dict`` keys, ``named_modules()`` order and
the ``bit_config_mobilenetv2_w1_*`` schedules (bit_config.py) line up with checkpoints and fixtures of the reference.

The forward is one rule applied along the layer tables below: ``conv -> ReLU6 -> QuantAct`` (quant_modules.py:205-305 case 0)
for every activated layer, the bare ``conv`` for a unit's projection, and the unit-closing ``quant_act_int32`` in its
residual form (fixedpoint_fn case 1, quant_utils.py:416-456) when the float unit keeps its input shape.

Execution: a frozen, eval-mode network called on a CUDA tensor runs the FUSED INTEGER PLAN of
``hawq_amd.engine_mbv2.MobileNetV2Engine`` (int8 tensors between the convs of a unit, int32 for the signed 16-bit values between
units, three launches per unit, hipGraph replay).  Otherwise (un-frozen = range calibration, or ``fused = False``) it steps module by
module through the same HIP library in the reference's fp32-tuple convention: 1x1 convs on the MFMA implicit-GEMM kernel, the
3x3 depthwise convs on ``hawq_depthwise3x3`` (adapters.hip), every QuantAct on ``hawq_fixedpoint_f32``.  Either way the
QuantConv2d classifier - whose reference forward runs an fp32 conv on un-rounded ``x / S_a`` (quant_modules.py:727-736) - is
reproduced to 2 ulp with identical int32 accumulators, not bit for bit."""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F

from .quant_modules import QuantAct, QuantAveragePool2d, QuantBnConv2d, QuantConv2d

# (float child, QuantAct that consumes its ReLU6'd output) of a unit's activated layers, in execution order; the projection
# `conv3` has no activation of its own and feeds the unit-closing `quant_act_int32`
UNIT_ACTIVATED = (("conv1", "quant_act1"), ("conv2", "quant_act2"))
UNIT_PROJECTION = "conv3"


def _fold(float_block) -> QuantBnConv2d:
    """conv + BN block of the float network -> QuantBnConv2d holding the same parameters"""
    q = QuantBnConv2d()
    q.set_param(float_block.conv, float_block.bn)
    return q


def _activated(conv, act, x, scale):
    """conv -> ReLU6 -> QuantAct (case 0): the pattern of every layer that is not a linear projection"""
    x, w_scale = conv(x, scale)
    return act(F.relu6(x), scale, w_scale)


class Q_LinearBottleneck(nn.Module):
    """One inverted-residual unit, built from the float unit's own children (role of q_mobilenetv2.py:12-93)."""

    def __init__(self, float_unit, residual: bool):
        super().__init__()
        self.residual = bool(residual)
        self.quant_act = QuantAct()
        self.activated = tuple((c, a) for c, a in UNIT_ACTIVATED if hasattr(float_unit, c))   # a unit may lack the expansion conv
        for conv, act in self.activated:
            setattr(self, conv, _fold(getattr(float_unit, conv)))
            setattr(self, act, QuantAct())
        setattr(self, UNIT_PROJECTION, _fold(getattr(float_unit, UNIT_PROJECTION)))
        self.quant_act_int32 = QuantAct()

    def forward(self, x, scaling_factor_int32=None):
        skip = x if self.residual else None
        x, scale = self.quant_act(x, scaling_factor_int32)
        for conv, act in self.activated:
            x, scale = _activated(getattr(self, conv), getattr(self, act), x, scale)
        x, w_scale = getattr(self, UNIT_PROJECTION)(x, scale)
        if skip is None:
            return self.quant_act_int32(x, scale, w_scale)
        # both branches are requantised separately inside (the block input still carries the previous unit's scale)
        return self.quant_act_int32(x + skip, scale, w_scale, skip, scaling_factor_int32, None)


class Q_MobileNetV2(nn.Module):
    """Quantised mirror of a pytorchcv-style float MobileNetV2 (role of q_mobilenetv2.py:96-209)."""

    def __init__(self, model):
        super().__init__()
        f = model.features
        self.quant_input = QuantAct()
        self.init_block = _fold(f.init_block)
        self.quant_act_int32 = QuantAct()
        self.features = nn.Sequential()
        width = f.init_block.conv.out_channels
        self.channels = []
        for sname, stage in f.named_children():
            if not sname.startswith("stage"):
                continue
            qstage, widths = nn.Sequential(), []
            for uname, unit in stage.named_children():
                proj, dw = getattr(unit, UNIT_PROJECTION).conv, unit.conv2.conv
                qstage.add_module(uname, Q_LinearBottleneck(unit, residual=(proj.out_channels == width and dw.stride[0] == 1)))
                width = proj.out_channels
                widths.append(width)
            self.features.add_module(sname, qstage)
            self.channels.append(widths)
        self.quant_act_before_final_block = QuantAct()
        self.features.add_module("final_block", _fold(f.final_block))
        self.quant_act_int32_final = QuantAct()
        pool = QuantAveragePool2d()
        pool.set_param(f.final_pool)
        self.features.add_module("final_pool", pool)
        self.quant_act_output = QuantAct()
        self.output = QuantConv2d()
        self.output.set_param(model.output)
        self.fused = True          # use the integer plan when frozen + eval + CUDA
        self._engine = None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._on_state_dict_loaded())

    def units(self):
        for sname, stage in self.features.named_children():
            if sname.startswith("stage"):
                yield from stage.children()

    def forward(self, x):
        if self.fused and x.is_cuda and not self.training and self.is_frozen():
            from .engine_mbv2 import PlanNotApplicable
            try:
                return self.engine()(x)
            except PlanNotApplicable as exc:
                # a configuration the fused plan does not take (other input / unit-output QuantAct widths, a classifier bias, a
                # ReLU6 that does not fold): the module-by-module path computes it, as it did before the plan existed
                import warnings
                warnings.warn(f"Q_MobileNetV2: fused integer plan not applicable ({exc}); using the module-by-module path")
                self.fused = False
                self._engine = None   # a half-built executor is not kept
        return self.forward_modules(x)

    def forward_modules(self, x):
        """Module-by-module forward (q_mobilenetv2.py:176-209)."""
        x, scale = self.quant_input(x)
        x, scale = _activated(self.init_block, self.quant_act_int32, x, scale)
        for unit in self.units():
            x, scale = unit(x, scale)
        x, scale = self.quant_act_before_final_block(x, scale)
        x, scale = _activated(self.features.final_block, self.quant_act_int32_final, x, scale)
        x = self.features.final_pool(x, scale)
        x, scale = self.quant_act_output(x, scale)
        x, _ = self.output(x, scale)
        return x.flatten(1)

    def is_frozen(self):
        acts = [m for m in self.modules() if isinstance(m, QuantAct)]
        convs = [m for m in self.modules() if isinstance(m, (QuantBnConv2d, QuantConv2d))]
        return all((not m.running_stat) for m in acts) and all(m.fix_flag for m in convs)

    def engine(self, **kw):
        """Build (or return the cached) fused integer executor for this frozen network."""
        from .engine_mbv2 import MobileNetV2Engine
        if self._engine is None or kw:
            opts = {k: v for k, v in getattr(self, "engine_defaults", {}).items() if k == "from_buffers"}
            self._engine = MobileNetV2Engine(self, **{**opts, **kw})
        return self._engine

    def invalidate_engine(self):
        self._engine = None

    def _on_state_dict_loaded(self):
        from .quant_modules import trust_integer_buffers
        self.invalidate_engine()
        trust_integer_buffers(self, False)
        if getattr(self, "engine_defaults", None):
            self.engine_defaults = dict(self.engine_defaults, from_buffers=False)


def q_get_mobilenetv2(model, width_scale, remove_exp_conv=False):
    """The reference's generic entry point (q_mobilenetv2.py:212-236: ``q_get_mobilenetv2(model, width_scale, remove_exp_conv)``).
    Its two extra arguments describe the float network handed in; this builder reads the same facts off the network itself, so they
    are only checked: a width other than the float model's, or ``remove_exp_conv`` that contradicts its first unit, is an error."""
    width = {"mobilenetv2_w1": 1.0, "mobilenetv2_w3d4": 0.75, "mobilenetv2_wd2": 0.5, "mobilenetv2_wd4": 0.25}.get(getattr(model, "arch", "mobilenetv2_w1"))
    if width is not None and abs(float(width_scale) - width) > 1e-9:
        raise ValueError(f"width_scale={width_scale} but the float network is {getattr(model, 'arch', 'mobilenetv2_w1')}")
    first = next(iter(next(s for n, s in model.features.named_children() if n.startswith("stage")).children()))
    has_exp = getattr(first, "use_exp_conv", hasattr(first, "conv1"))
    if bool(remove_exp_conv) == bool(has_exp):
        raise ValueError(f"remove_exp_conv={remove_exp_conv} contradicts the float network (first unit {'has' if has_exp else 'has no'} expansion conv)")
    return Q_MobileNetV2(model)


def q_mobilenetv2_w1(model):
    """Entry point named like the reference's (q_mobilenetv2.py:252): the width-1.0 network its bit schedules cover."""
    if getattr(model, "arch", "mobilenetv2_w1") != "mobilenetv2_w1":
        raise NotImplementedError("only mobilenetv2_w1 has bit schedules (bit_config.py)")
    return Q_MobileNetV2(model)
