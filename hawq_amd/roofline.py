"""Algorithmic work model of the frozen integer forward.

Canonical layer-fused byte model: every tensor once at its schedule bit-width; each conv reads
its input once + weights once (+12 B/channel of bias/m/e) and writes its epilogue output once;
residual tensors are 16-bit, identity-conv accumulators 32-bit; the block-input QuantAct is a
read of the 16-bit residual + a write at b/8; the input quantiser reads fp32 and writes int8;
the stem writes 16-bit after pooling; weights are counted once per batch.
``roofline.achieved`` in bench.py = algorithmic_bytes(batch) / measured forward time.
"""
from __future__ import annotations

from .bit_schedules import get_bit_config
from .skeleton import ARCH

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 MFMA (2x the 2.5 PF bf16 figure); ubench 4404 (32x32x32)


def layer_table(arch: str, scheme: str, size: int = 224):
    """Per-image list of dicts(name, macs, act_bytes, weight_bytes) in graph order."""
    cfg = get_bit_config(arch, scheme)
    units, widths, bottleneck, conv1_stride = ARCH[arch]
    rows = []
    hw = size * size
    rows.append(dict(name="quant_input", macs=0, act_bytes=3 * hw * 4 + 3 * hw, weight_bytes=0))
    h = size // 2
    stem = "quant_init_convbn" if bottleneck else "quant_init_block_convbn"
    hp = h // 2
    rows.append(dict(name=stem, macs=147 * 64 * h * h, act_bytes=3 * hw + 64 * hp * hp * 2,
                     weight_bytes=147 * 64 * cfg[stem] // 8 + 64 * 12))
    h, cin = hp, 64
    for si, (n, wd) in enumerate(zip(units, widths)):
        for ui in range(n):
            p = f"stage{si + 1}.unit{ui + 1}."
            stride = 2 if (ui == 0 and si > 0) else 1
            resize = (cin != wd) or stride != 1
            ab = cfg[p + "quant_act"]
            rows.append(dict(name=p + "quant_act", macs=0, act_bytes=cin * h * h * 2 + cin * h * h * ab // 8,
                             weight_bytes=0))
            ho = h // stride
            if bottleneck:
                mid = wd // 4
                s1, s2 = (stride, 1) if conv1_stride else (1, stride)
                h1 = h // s1
                convs = [("quant_convbn1", cin, mid, 1, h, h1, ab, cfg[p + "quant_act1"]),
                         ("quant_convbn2", mid, mid, 3, h1, ho, cfg[p + "quant_act1"], cfg[p + "quant_act2"]),
                         ("quant_convbn3", mid, wd, 1, ho, ho, cfg[p + "quant_act2"], None)]
            else:
                convs = [("quant_convbn1", cin, wd, 3, h, ho, ab, cfg[p + "quant_act1"]),
                         ("quant_convbn2", wd, wd, 3, ho, ho, cfg[p + "quant_act1"], None)]
            for cname, ci, co, k, hin, hout, in_b, out_b in convs:
                wb = cfg[p + cname]
                act = ci * hin * hin * in_b // 8
                if out_b is not None:
                    act += co * hout * hout * out_b // 8
                else:  # last conv: identity read (16-bit residual or int32 identity accumulators) + 16-bit out
                    act += co * hout * hout * (4 if resize else 2) + co * hout * hout * 2
                rows.append(dict(name=p + cname, macs=ci * k * k * co * hout * hout, act_bytes=act,
                                 weight_bytes=ci * k * k * co * wb // 8 + co * 12))
            if resize:
                wb = cfg[p + "quant_identity_convbn"]
                rows.append(dict(name=p + "quant_identity_convbn", macs=cin * wd * ho * ho,
                                 act_bytes=cin * h * h * ab // 8 + wd * ho * ho * 4,
                                 weight_bytes=cin * wd * wb // 8 + wd * 12))
            cin, h = wd, ho
    rows.append(dict(name="final_pool+quant_act_output", macs=0, act_bytes=cin * h * h * 2 + cin, weight_bytes=0))
    rows.append(dict(name="quant_output", macs=cin * 1000, act_bytes=cin + 1000 * 4,
                     weight_bytes=cin * 1000 + 1000 * 12))
    return rows


def algorithmic_bytes(arch: str, scheme: str, batch: int) -> int:
    rows = layer_table(arch, scheme)
    return batch * sum(r["act_bytes"] for r in rows) + sum(r["weight_bytes"] for r in rows)


def macs(arch: str, scheme: str, batch: int) -> int:
    return batch * sum(r["macs"] for r in layer_table(arch, scheme))
