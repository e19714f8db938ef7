"""Quantised MobileNetV2 with the reference's structure and names (utils/models/q_mobilenetv2.py:12-262): same classes,
attribute names (``quant_input``, ``init_block``, ``features.stageN.unitM.{quant_act, conv1, quant_act1, conv2, quant_act2,
conv3, quant_act_int32}``, ``quant_act_before_final_block``, ``features.final_block``, ``quant_act_int32_final``,
``features.final_pool``, ``quant_act_output``, ``output``) and call order, so state_dict keys and the
``bit_config_mobilenetv2_w1_*`` schedules line up.

Execution is module by module through the HIP library (the fp32-tuple convention of the reference): 1x1 convs on the MFMA
implicit-GEMM kernel, 3x3 depthwise convs on hawq_conv2d_grouped, every QuantAct on hawq_fixedpoint_f32.  The fused integer
plan (hawq_amd.engine) covers the ResNets only - SURVEY.md 8(f).3 lists MobileNetV2 as the next widening."""
from __future__ import annotations

import torch.nn as nn

from .quant_modules import QuantAct, QuantAveragePool2d, QuantBnConv2d, QuantConv2d


class Q_LinearBottleneck(nn.Module):
    """Quantised MobileNetV2 unit (reference: q_mobilenetv2.py:12-93)."""

    def __init__(self, model, in_channels, out_channels, stride, expansion, remove_exp_conv):
        super().__init__()
        self.residual = (in_channels == out_channels) and (stride == 1)
        self.use_exp_conv = (expansion or (not remove_exp_conv))
        self.activatition_func = nn.ReLU6()
        self.quant_act = QuantAct()
        if self.use_exp_conv:
            self.conv1 = QuantBnConv2d()
            self.conv1.set_param(model.conv1.conv, model.conv1.bn)
            self.quant_act1 = QuantAct()
        self.conv2 = QuantBnConv2d()
        self.conv2.set_param(model.conv2.conv, model.conv2.bn)
        self.quant_act2 = QuantAct()
        self.conv3 = QuantBnConv2d()
        self.conv3.set_param(model.conv3.conv, model.conv3.bn)
        self.quant_act_int32 = QuantAct()

    def forward(self, x, scaling_factor_int32=None):
        if self.residual:
            identity = x
        x, act_scaling_factor = self.quant_act(x, scaling_factor_int32, None, None, None, None)
        if self.use_exp_conv:
            x, weight_scaling_factor = self.conv1(x, act_scaling_factor)
            x = self.activatition_func(x)
            x, act_scaling_factor = self.quant_act1(x, act_scaling_factor, weight_scaling_factor, None, None)
        x, weight_scaling_factor = self.conv2(x, act_scaling_factor)
        x = self.activatition_func(x)
        x, act_scaling_factor = self.quant_act2(x, act_scaling_factor, weight_scaling_factor, None, None)
        x, weight_scaling_factor = self.conv3(x, act_scaling_factor)   # linear bottleneck: no activation
        if self.residual:
            x = x + identity
            x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor, identity,
                                                         scaling_factor_int32, None)
        else:
            x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor, None, None, None)
        return x, act_scaling_factor


class Q_MobileNetV2(nn.Module):
    """Quantised MobileNetV2 (reference: q_mobilenetv2.py:96-209)."""

    def __init__(self, model, channels, init_block_channels, final_block_channels, remove_exp_conv, in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super().__init__()
        self.in_size, self.num_classes, self.channels = in_size, num_classes, channels
        self.activatition_func = nn.ReLU6()
        self.quant_input = QuantAct()
        self.add_module("init_block", QuantBnConv2d())
        self.init_block.set_param(model.features.init_block.conv, model.features.init_block.bn)
        self.quant_act_int32 = QuantAct()
        self.features = nn.Sequential()
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            cur_stage = getattr(model.features, f'stage{i + 1}')
            for j, out_channels in enumerate(channels_per_stage):
                cur_unit = getattr(cur_stage, f'unit{j + 1}')
                stride = 2 if (j == 0) and (i != 0) else 1
                expansion = (i != 0) or (j != 0)
                stage.add_module("unit{}".format(j + 1), Q_LinearBottleneck(cur_unit, in_channels=in_channels, out_channels=out_channels,
                                                                            stride=stride, expansion=expansion,
                                                                            remove_exp_conv=remove_exp_conv))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.quant_act_before_final_block = QuantAct()
        self.features.add_module("final_block", QuantBnConv2d())
        self.features.final_block.set_param(model.features.final_block.conv, model.features.final_block.bn)
        self.quant_act_int32_final = QuantAct()
        self.features.add_module("final_pool", QuantAveragePool2d())
        self.features.final_pool.set_param(model.features.final_pool)
        self.quant_act_output = QuantAct()
        self.output = QuantConv2d()
        self.output.set_param(model.output)

    def forward(self, x):
        x, act_scaling_factor = self.quant_input(x)
        x, weight_scaling_factor = self.init_block(x, act_scaling_factor)
        x = self.activatition_func(x)
        x, act_scaling_factor = self.quant_act_int32(x, act_scaling_factor, weight_scaling_factor, None, None)
        for i, channels_per_stage in enumerate(self.channels):
            cur_stage = getattr(self.features, f'stage{i + 1}')
            for j, _ in enumerate(channels_per_stage):
                x, act_scaling_factor = getattr(cur_stage, f'unit{j + 1}')(x, act_scaling_factor)
        x, act_scaling_factor = self.quant_act_before_final_block(x, act_scaling_factor, None, None, None, None)
        x, weight_scaling_factor = self.features.final_block(x, act_scaling_factor)
        x = self.activatition_func(x)
        x, act_scaling_factor = self.quant_act_int32_final(x, act_scaling_factor, weight_scaling_factor, None, None, None)
        x = self.features.final_pool(x, act_scaling_factor)
        x, act_scaling_factor = self.quant_act_output(x, act_scaling_factor, None, None, None, None)
        x, act_scaling_factor = self.output(x, act_scaling_factor)
        return x.view(x.size(0), -1)

    forward_modules = forward

    def is_frozen(self):
        acts = [m for m in self.modules() if isinstance(m, QuantAct)]
        convs = [m for m in self.modules() if isinstance(m, (QuantBnConv2d, QuantConv2d))]
        return all((not m.running_stat) for m in acts) and all(m.fix_flag for m in convs)

    def invalidate_engine(self):   # API symmetry with the ResNets (there is no fused plan to drop)
        pass


def q_get_mobilenetv2(model, width_scale, remove_exp_conv=False):
    """q_mobilenetv2.py:212-249 (width_scale 1.0 only: the one the reference's schedules cover)."""
    if width_scale != 1.0:
        raise NotImplementedError("only mobilenetv2_w1 has bit schedules (bit_config.py)")
    return Q_MobileNetV2(model, channels=model.channels, init_block_channels=32, final_block_channels=1280,
                         remove_exp_conv=remove_exp_conv)


def q_mobilenetv2_w1(model):
    return q_get_mobilenetv2(model, width_scale=1.0)
