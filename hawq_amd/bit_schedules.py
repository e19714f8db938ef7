"""Per-layer bit-width schedules for the quantized ResNets.

Same data as the reference's ``bit_config_dict`` (bit_config.py:1-4204, keys
``bit_config_<arch>_<scheme>``; consumed by quant_train.py:264-299), stored compactly:
one character per quantized module in graph order ('4', '8', 'g' = 16 bit).  The table
was produced by tools/gen_bit_schedules.py.  ``bit_config_dict`` is rebuilt at import
with the reference's key names, so ``bit_config_dict["bit_config_resnet50_bops_0.5"]
["stage1.unit1.quant_convbn1"]`` works unchanged.
"""
from __future__ import annotations

from .skeleton import ARCH


def module_names(arch: str) -> list[str]:
    """Names of the quantized modules of Q_ResNet* in the order the schedules list them."""
    units, widths, bottleneck, _ = ARCH[arch]
    stem = "quant_init_convbn" if bottleneck else "quant_init_block_convbn"
    names = ["quant_input", stem, "quant_act_int32"]
    cin = 64
    for si, (n, w) in enumerate(zip(units, widths)):
        for ui in range(n):
            p = f"stage{si + 1}.unit{ui + 1}."
            resize = (cin != w) or (ui == 0 and si > 0)
            names += [p + "quant_act", p + "quant_convbn1", p + "quant_act1", p + "quant_convbn2"]
            if bottleneck:
                names += [p + "quant_act2", p + "quant_convbn3"]
            if resize:
                names.append(p + "quant_identity_convbn")
            names.append(p + "quant_act_int32")
            cin = w
    names += ["quant_act_output", "quant_output"]
    return names


MOBILENETV2_UNITS = (1, 2, 3, 7, 4)   # units per stage of pytorchcv's mobilenetv2_w1 as q_get_mobilenetv2 groups them (q_mobilenetv2.py:224-233)


def mobilenetv2_module_names(stray: bool = False) -> list[str]:
    """Names of the quantized modules of Q_MobileNetV2 in schedule order (q_mobilenetv2.py:120-172).  Three of the
    reference's four schedules carry two stray entries (`...stage4.unit5.conv1.conv` / `.bn`, bit_config.py:3602-4204) that
    name the nn.Conv2d / BatchNorm2d children of a QuantBnConv2d; they are kept so that the dicts are the reference's."""
    names = ["quant_input", "init_block", "quant_act_int32"]
    for si, n in enumerate(MOBILENETV2_UNITS):
        for ui in range(n):
            p = f"features.stage{si + 1}.unit{ui + 1}."
            names += [p + "quant_act", p + "conv1"]
            if stray and (si, ui) == (3, 4):
                names += [p + "conv1.conv", p + "conv1.bn"]
            names += [p + "quant_act1", p + "conv2", p + "quant_act2", p + "conv3", p + "quant_act_int32"]
    return names + ["quant_act_before_final_block", "features.final_block", "quant_act_int32_final", "quant_act_output", "output"]


_MOBILENET_TABLE = {   # tools/gen_bit_schedules.py
    "mobilenetv2_w1_uniform8": "88g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g88888888g888888g888888g888888g888888g888888g888888g88g88",
    "mobilenetv2_w1_modelsize_0.5": "88g888888g888888g888888g888888g448888g448888g888888g448844g448844g448844g888888g448844g448844g888888g448844g888888g888888g88g88",
    "mobilenetv2_w1_bops_0.5": "88g888888g448888g448888g448888g448888g448888g888888g448888g448844g448844g88888888g448844g448888g448888g448888g448888g444488g88g88",
    "mobilenetv2_w1_uniform4": "88g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g44444444g444444g444444g444444g444444g444444g444444g88g88",
}

_TABLE = {
    "resnet18_uniform8": "88g8888g8888g88888g8888g88888g8888g88888g8888g88",
    "resnet18_uniform4": "88g4444g4444g44444g4444g44444g4444g44444g4444g88",
    "resnet18_modelsize_0.75": "88g8888g8888g88888g8888g88888g8844g88888g8844g88",
    "resnet18_modelsize_0.5": "88g8888g8888g88888g8888g88888g8888g88448g4444g88",
    "resnet18_modelsize_0.25": "88g8888g8888g88888g8888g88888g8888g44444g4444g88",
    "resnet18_bops_0.75": "88g8888g8888g88888g8844g88888g8844g88448g8844g88",
    "resnet18_bops_0.5": "88g8888g8844g88888g8844g88888g4444g44444g4444g88",
    "resnet18_bops_0.25": "88g8888g8844g88448g4444g44444g4444g44444g4444g88",
    "resnet18_latency_0.75": "88g8888g8888g88888g8844g88888g8844g88448g8844g88",
    "resnet18_latency_0.5": "88g8888g8888g88448g8844g88888g4444g88448g4444g88",
    "resnet18_latency_0.25": "88g8888g8844g88448g4444g44444g4444g44444g4444g88",
    "resnet50_uniform8": "88g8888888g888888g888888g8888888g888888g888888g888888g8888888g888888g888888g888888g888888g888888g8888888g888888g888888g88",
    "resnet50_uniform4": "88g4444444g444444g444444g4444444g444444g444444g444444g4444444g444444g444444g444444g444444g444444g4444444g444444g444444g88",
    "resnet50_modelsize_0.75": "88g8888888g888888g888888g8888888g888888g888888g888888g8888888g888888g888888g888888g888888g888888g8844888g884488g884488g88",
    "resnet50_modelsize_0.5": "88g8888888g888888g888888g8888888g888888g888888g888888g8844888g888888g884444g888888g884488g884488g8844888g884444g884444g88",
    "resnet50_modelsize_0.25": "88g8888888g888888g888888g8888888g888888g888888g888888g8844888g884444g884444g884444g884444g444444g4444884g444444g884444g88",
    "resnet50_bops_0.75": "88g8888888g888888g884488g8888888g888888g888888g884488g8844888g888888g884444g884488g884488g884488g8844888g888888g888888g88",
    "resnet50_bops_0.5": "88g8888888g888888g884488g8844888g884488g884444g884488g8844888g884444g884444g884444g884444g884488g8844888g884488g884488g88",
    "resnet50_bops_0.25": "88g8844888g444488g884444g8844888g444444g444444g884444g8844888g444444g444444g444444g444444g444444g8844888g884444g884444g88",
    "resnet50_latency_0.75": "88g8888888g888888g884488g8844888g888888g888888g884488g8844888g888888g884488g888888g884488g884488g8844888g888888g888888g88",
    "resnet50_latency_0.5": "88g8888888g888888g884444g8844888g888888g884488g884488g8844888g444444g884444g884444g884444g884444g8844888g884488g884488g88",
    "resnet50_latency_0.25": "88g8888888g884488g884444g8844888g444444g444444g884444g4444884g444444g444444g444444g444444g884444g8844888g884444g884444g88",
    "resnet50b_uniform8": "88g8888888g888888g888888g8888888g888888g888888g888888g8888888g888888g888888g888888g888888g888888g8888888g888888g888888g88",
    "resnet50b_uniform4": "88g4444444g444444g444444g4444444g444444g444444g444444g4444444g444444g444444g444444g444444g444444g4444444g444444g444444g88",
    "resnet101_uniform8": "88g8888888g888888g888888g8888888g888888g888888g888888g8888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g888888g8888888g888888g888888g88",
    "resnet101_uniform4": "88g4444444g444444g444444g4444444g444444g444444g444444g4444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g444444g4444444g444444g444444g88",
}

_BITS = {"4": 4, "8": 8, "g": 16}

bit_config_dict = {}
for _name, _s in _TABLE.items():
    _arch = _name.split("_", 1)[0]
    _names = module_names(_arch)
    assert len(_names) == len(_s), _name
    bit_config_dict["bit_config_" + _name] = {n: _BITS[c] for n, c in zip(_names, _s)}


for _name, _s in _MOBILENET_TABLE.items():
    _names = mobilenetv2_module_names(stray=len(_s) == 129)
    assert len(_names) == len(_s), _name
    bit_config_dict["bit_config_" + _name] = {n: _BITS[c] for n, c in zip(_names, _s)}


def get_bit_config(arch: str, scheme: str) -> dict:
    return bit_config_dict[f"bit_config_{arch}_{scheme}"]
