"""Float ResNet skeletons with the attribute tree HAWQ's quantized graphs dereference.

The reference builds its float model with ``pytorchcv.model_provider.get_model``
(quant_train.py:227,233) and then walks ``features.init_block.conv.{conv,bn}``,
``features.stageN.unitM.{resize_identity, body.conv1..3.{conv,bn}, identity_conv.{conv,bn}}``
and ``output`` (utils/models/q_resnet.py:22-51, 84-112, 206-229, 270-289).
pytorchcv is not installable here, so this module provides stand-ins with exactly that
attribute tree.  ``resnet50`` puts the stride on the first 1x1 conv of a bottleneck
(pytorchcv ``conv1_stride=True``), ``resnet50b`` on the 3x3.

Only shapes/names matter: weights come from a checkpoint or from ``init_synthetic``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

ARCH = {
    #            units per stage, stage widths,            bottleneck, stride on conv1
    "resnet18": ((2, 2, 2, 2), (64, 128, 256, 512), False, True),
    "resnet50": ((3, 4, 6, 3), (256, 512, 1024, 2048), True, True),
    "resnet50b": ((3, 4, 6, 3), (256, 512, 1024, 2048), True, False),
    "resnet101": ((3, 4, 23, 3), (256, 512, 1024, 2048), True, True),
}


def _conv_bn(cin, cout, k, stride, pad):
    blk = nn.Module()
    blk.conv = nn.Conv2d(cin, cout, k, stride, pad, bias=False)
    blk.bn = nn.BatchNorm2d(cout)
    return blk


def _unit(cin, cout, stride, bottleneck, conv1_stride):
    u = nn.Module()
    u.resize_identity = (cin != cout) or (stride != 1)
    u.body = nn.Module()
    if bottleneck:
        mid = cout // 4
        s1, s2 = (stride, 1) if conv1_stride else (1, stride)
        u.body.conv1 = _conv_bn(cin, mid, 1, s1, 0)
        u.body.conv2 = _conv_bn(mid, mid, 3, s2, 1)
        u.body.conv3 = _conv_bn(mid, cout, 1, 1, 0)
    else:
        u.body.conv1 = _conv_bn(cin, cout, 3, stride, 1)
        u.body.conv2 = _conv_bn(cout, cout, 3, 1, 1)
    if u.resize_identity:
        u.identity_conv = _conv_bn(cin, cout, 1, stride, 0)
    return u


def build_float_resnet(arch: str, num_classes: int = 1000) -> nn.Module:
    units, widths, bottleneck, conv1_stride = ARCH[arch]
    net = nn.Module()
    net.arch = arch
    net.features = nn.Module()
    net.features.init_block = nn.Module()
    net.features.init_block.conv = _conv_bn(3, 64, 7, 2, 3)
    cin = 64
    for si, (n, w) in enumerate(zip(units, widths)):
        stage = nn.Module()
        for ui in range(n):
            stride = 2 if (ui == 0 and si > 0) else 1
            setattr(stage, f"unit{ui + 1}", _unit(cin, w, stride, bottleneck, conv1_stride))
            cin = w
        setattr(net.features, f"stage{si + 1}", stage)
    net.output = nn.Linear(cin, num_classes)
    return net


def build_float_mobilenetv2(num_classes: int = 1000) -> nn.Module:
    """pytorchcv ``mobilenetv2_w1`` attribute tree as utils/models/q_mobilenetv2.py:120-172 dereferences it:
    ``features.init_block.{conv,bn}`` (3x3/2, 3->32), ``features.stageN.unitM.{conv1,conv2,conv3}.{conv,bn}`` (1x1 expand x6 -
    x1 in the very first unit -, 3x3 depthwise, 1x1 linear projection), ``features.final_block.{conv,bn}`` (1x1 320->1280),
    ``features.final_pool`` (AvgPool2d(7)), ``output`` (1x1 conv, no bias)."""
    layers, downsample, widths = [1, 2, 3, 4, 3, 3, 1], [0, 1, 1, 1, 0, 1, 0], [16, 24, 32, 64, 96, 160, 320]
    channels = [[]]
    for c, n, d in zip(widths, layers, downsample):   # q_get_mobilenetv2's grouping (q_mobilenetv2.py:224-233)
        if d:
            channels.append([c] * n)
        else:
            channels[-1] += [c] * n
    net = nn.Module()
    net.arch = "mobilenetv2_w1"
    net.features = nn.Module()
    net.features.init_block = _conv_bn(3, 32, 3, 2, 1)
    cin = 32
    for si, per_stage in enumerate(channels):
        stage = nn.Module()
        for ui, cout in enumerate(per_stage):
            stride = 2 if (ui == 0 and si != 0) else 1
            mid = cin * 6 if (si != 0 or ui != 0) else cin
            u = nn.Module()
            u.conv1 = _conv_bn(cin, mid, 1, 1, 0)
            u.conv2 = _conv_bn(mid, mid, 3, stride, 1)
            u.conv2.conv = nn.Conv2d(mid, mid, 3, stride, 1, groups=mid, bias=False)
            u.conv3 = _conv_bn(mid, cout, 1, 1, 0)
            setattr(stage, f"unit{ui + 1}", u)
            cin = cout
        setattr(net.features, f"stage{si + 1}", stage)
    net.features.final_block = _conv_bn(cin, 1280, 1, 1, 0)
    net.features.final_pool = nn.AvgPool2d(kernel_size=7, stride=1)
    net.output = nn.Conv2d(1280, num_classes, 1, bias=False)
    net.channels = channels
    return net


def init_synthetic(net: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic synthetic weights (no zoo checkpoints are reachable here).

    Conv/linear keep PyTorch's default init drawn under ``manual_seed(seed)``; BN
    statistics are randomised so that BN folding (quant_modules.py:441-449) is not a
    no-op: running_mean~N(0,.1), running_var~U(.5,1.5), weight~U(.5,1.5), bias~N(0,.1).
    Module construction order fixes the RNG stream, so the same seed gives the same
    tensors in every process that uses the same torch build.
    """
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fan_in = m.weight[0].numel()
                bound = (1.0 / fan_in) ** 0.5
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 2 - 1) * bound)
                if m.bias is not None:
                    m.bias.copy_((torch.rand(m.bias.shape, generator=g) * 2 - 1) * bound)
            elif isinstance(m, nn.BatchNorm2d):
                c = m.num_features
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(c, generator=g) + 0.5)
                m.weight.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.copy_(torch.randn(c, generator=g) * 0.1)
    return net


def synthetic_images(batch: int, seed: int = 0, size: int = 224) -> torch.Tensor:
    """Normalised-image-like synthetic input: seeded N(0,1)."""
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randn(batch, 3, size, size, generator=g)
