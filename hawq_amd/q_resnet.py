"""Quantised ResNet graphs with the reference's structure and names.

Same classes, attribute names (``quant_input``, ``quant_init_convbn`` / ``quant_init_block_convbn``,
``stageN.unitM`` registered with a dot in the name, ``quant_act_int32`` ...) and call order as
utils/models/q_resnet.py:16-331, so state_dict keys and bit_config schedules line up.

Execution: a frozen, eval-mode network called on a CUDA tensor runs the FUSED INTEGER PLAN
(hawq_amd.engine.IntegerEngine: int8/int4 tensors between kernels, ~3 launches per residual
unit, hipGraph replay).  Otherwise (un-frozen = range calibration, or ``fused=False``) it steps
module by module through the same HIP library in the reference's fp32-tuple convention.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .quant_modules import QuantAct, QuantAveragePool2d, QuantBnConv2d, QuantLinear


class _QUnit(nn.Module):
    """Common part of Q_ResUnitBn (bottleneck) and Q_ResBlockBn (basic)."""
    n_body = 0

    def set_param(self, unit):
        self.resize_identity = unit.resize_identity
        self.quant_act = QuantAct()
        for i in range(1, self.n_body + 1):
            convbn = getattr(unit.body, f"conv{i}")
            q = QuantBnConv2d()
            q.set_param(convbn.conv, convbn.bn)
            setattr(self, f"quant_convbn{i}", q)
            if i < self.n_body:
                setattr(self, f"quant_act{i}", QuantAct())
        if self.resize_identity:
            self.quant_identity_convbn = QuantBnConv2d()
            self.quant_identity_convbn.set_param(unit.identity_conv.conv, unit.identity_conv.bn)
        self.quant_act_int32 = QuantAct()

    def forward(self, x, scaling_factor_int32=None):
        # q_resnet.py:231-260 (bottleneck) / :291-316 (basic block)
        if self.resize_identity:
            x, act_scaling_factor = self.quant_act(x, scaling_factor_int32)
            identity_act_scaling_factor = act_scaling_factor.clone()
            identity, identity_weight_scaling_factor = self.quant_identity_convbn(x, act_scaling_factor)
        else:
            identity = x
            x, act_scaling_factor = self.quant_act(x, scaling_factor_int32)
        for i in range(1, self.n_body):
            x, weight_scaling_factor = getattr(self, f"quant_convbn{i}")(x, act_scaling_factor)
            x = torch.relu(x)
            x, act_scaling_factor = getattr(self, f"quant_act{i}")(x, act_scaling_factor, weight_scaling_factor)
        x, weight_scaling_factor = getattr(self, f"quant_convbn{self.n_body}")(x, act_scaling_factor)
        x = x + identity
        if self.resize_identity:
            x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor, identity,
                                                         identity_act_scaling_factor, identity_weight_scaling_factor)
        else:
            x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor, identity,
                                                         scaling_factor_int32, None)
        x = torch.relu(x)
        return x, act_scaling_factor


class Q_ResUnitBn(_QUnit):
    """Quantised bottleneck unit (reference: q_resnet.py:199-260)."""
    n_body = 3


class Q_ResBlockBn(_QUnit):
    """Quantised basic block (reference: q_resnet.py:263-316)."""
    n_body = 2


class _QResNet(nn.Module):
    channel = ()
    unit_cls = Q_ResUnitBn
    stem_name = "quant_init_convbn"

    def __init__(self, model):
        super().__init__()
        features = getattr(model, 'features')
        init_block = getattr(features, 'init_block')
        self.quant_input = QuantAct()
        stem = QuantBnConv2d()
        stem.set_param(init_block.conv.conv, init_block.conv.bn)
        setattr(self, self.stem_name, stem)
        self.quant_act_int32 = QuantAct()
        self.pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.act = nn.ReLU()
        for stage_num in range(4):
            stage = getattr(features, "stage{}".format(stage_num + 1))
            for unit_num in range(self.channel[stage_num]):
                unit = getattr(stage, "unit{}".format(unit_num + 1))
                quant_unit = self.unit_cls()
                quant_unit.set_param(unit)
                setattr(self, f"stage{stage_num + 1}.unit{unit_num + 1}", quant_unit)
        self.final_pool = QuantAveragePool2d(kernel_size=7, stride=1)
        self.quant_act_output = QuantAct()
        self.quant_output = QuantLinear()
        self.quant_output.set_param(getattr(model, 'output'))
        self.fused = True          # use the integer plan when frozen + eval + CUDA
        self._engine = None
        # new parameters / ranges make a cached plan (packed weights, tables, hipGraph) stale
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._on_state_dict_loaded())

    # -- structure helpers -------------------------------------------------------------------
    @property
    def stem(self):
        return getattr(self, self.stem_name)

    def units(self):
        for s in range(4):
            for u in range(self.channel[s]):
                yield f"stage{s + 1}.unit{u + 1}", getattr(self, f"stage{s + 1}.unit{u + 1}")

    def is_frozen(self):
        from .quant_modules import QuantBnConv2d as _C
        acts = [m for m in self.modules() if isinstance(m, QuantAct)]
        convs = [m for m in self.modules() if isinstance(m, (_C, QuantLinear))]
        return all((not m.running_stat) for m in acts) and all(m.fix_flag for m in convs)

    def engine(self, **kw):
        """Build (or return the cached) fused integer executor for this frozen network."""
        from .engine import IntegerEngine
        if self._engine is None or kw:
            # engine_defaults: set by hawq_amd.api.load_quantized_checkpoint (from_buffers=True)
            self._engine = IntegerEngine(self, **{**getattr(self, "engine_defaults", {}), **kw})
        return self._engine

    def invalidate_engine(self):
        self._engine = None

    def _on_state_dict_loaded(self):
        """load_state_dict brought new float parameters / ranges: the cached plan is stale, and so is any trust in integer
        buffers loaded earlier from a quantized checkpoint (hawq_amd.api.load_quantized_checkpoint sets it again itself)."""
        from .quant_modules import trust_integer_buffers
        self.invalidate_engine()
        trust_integer_buffers(self, False)
        if getattr(self, "engine_defaults", None):
            self.engine_defaults = dict(self.engine_defaults, from_buffers=False)

    # -- forward -------------------------------------------------------------------------------
    def forward_modules(self, x):
        """Module-by-module forward (q_resnet.py:53-74 / 114-135)."""
        x, act_scaling_factor = self.quant_input(x)
        x, weight_scaling_factor = self.stem(x, act_scaling_factor)
        x = self.pool(x)
        x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor)
        x = self.act(x)
        for _, unit in self.units():
            x, act_scaling_factor = unit(x, act_scaling_factor)
        x = self.final_pool(x, act_scaling_factor)
        x, act_scaling_factor = self.quant_act_output(x, act_scaling_factor)
        x = x.view(x.size(0), -1)
        return self.quant_output(x, act_scaling_factor)

    def forward(self, x):
        if self.fused and x.is_cuda and not self.training and self.is_frozen():
            return self.engine()(x)
        return self.forward_modules(x)


class Q_ResNet18(_QResNet):
    """Quantised ResNet18 (reference: q_resnet.py:16-74)."""
    channel = [2, 2, 2, 2]
    unit_cls = Q_ResBlockBn
    stem_name = "quant_init_block_convbn"


class Q_ResNet50(_QResNet):
    """Quantised ResNet50 (reference: q_resnet.py:77-135)."""
    channel = [3, 4, 6, 3]


class Q_ResNet101(_QResNet):
    """Quantised ResNet101 (reference: q_resnet.py:138-196)."""
    channel = [3, 4, 23, 3]


def q_resnet18(model):
    return Q_ResNet18(model)


def q_resnet50(model):
    return Q_ResNet50(model)


def q_resnet101(model):
    return Q_ResNet101(model)


quantize_arch_dict = {'resnet18': q_resnet18, 'resnet50': q_resnet50, 'resnet50b': q_resnet50,
                      'resnet101': q_resnet101}


def apply_bit_config(model, bit_config, bias_bit=32, channel_wise=True, act_percentile=0, act_range_momentum=0.99,
                     weight_percentile=0, fix_BN=True, fix_BN_threshold=None, fixed_point_quantization=False):
    """The per-module configuration loop of quant_train.py:264-299 with its CLI defaults
    (quant_train.py:26-152): sets quant_mode/bias_bit/per_channel/... and the bit-widths;
    4-bit activations become 'asymmetric' (unsigned, zero-point unused)."""
    matched = 0
    for name, m in model.named_modules():
        if name in bit_config:
            matched += 1
            m.quant_mode = 'symmetric'
            m.bias_bit = bias_bit
            m.quantize_bias = (bias_bit != 0)
            m.per_channel = channel_wise
            m.act_percentile = act_percentile
            m.act_range_momentum = act_range_momentum
            m.weight_percentile = weight_percentile
            m.fix_flag = False
            m.fix_BN = fix_BN
            m.fix_BN_threshold = fix_BN_threshold
            m.training_BN_mode = fix_BN
            m.fixed_point_quantization = fixed_point_quantization
            bits = bit_config[name][0] if type(bit_config[name]) is tuple else bit_config[name]
            if hasattr(m, 'activation_bit'):
                m.activation_bit = bits
                if bits == 4:
                    m.quant_mode = 'asymmetric'
            else:
                m.weight_bit = bits
    if matched != len(bit_config):
        raise ValueError(f"bit_config names matched {matched} of {len(bit_config)} modules")
    if isinstance(model, _QResNet):
        model.invalidate_engine()
    return model
