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


def fused_plan_table(arch: str, scheme: str, size: int = 224):
    """Per-image bytes the FUSED plan of hawq_amd.engine has to move, launch by launch (dicts: name, read, write,
    macs, weight_bytes).  Differences to the canonical model above: the stem reads the fp32 image and writes the
    pooled uint16 residual + the first unit's int8/int4 input directly; the block-input QuantAct is the expand
    conv's second output (no separate 16-bit read); an identity conv runs inside the expand conv's launch and its
    int32 accumulators never exist; strided 1x1 convs only touch the pixels they use.  This is the traffic floor of
    the implementation as launched (profiles/traffic.json holds the measured counterpart)."""
    cfg = get_bit_config(arch, scheme)
    units, widths, bottleneck, conv1_stride = ARCH[arch]
    rows = []
    hw = size * size
    h = size // 4
    first = cfg["stage1.unit1.quant_act"]
    stem = "quant_init_convbn" if bottleneck else "quant_init_block_convbn"
    rows.append(dict(name="hawq_stem_fused", read=3 * hw * 4, write=64 * h * h * 2 + 64 * h * h * first // 8,
                     macs=147 * 64 * (size // 2) ** 2, weight_bytes=147 * 64 + 64 * 12))
    cin = 64
    n_units = sum(units)
    seen = 0
    for si, (n, wd) in enumerate(zip(units, widths)):
        for ui in range(n):
            seen += 1
            p = f"stage{si + 1}.unit{ui + 1}."
            stride = 2 if (ui == 0 and si > 0) else 1
            resize = (cin != wd) or stride != 1
            ab = cfg[p + "quant_act"]
            ho = h // stride
            if bottleneck:
                mid = wd // 4
                s1, s2 = (stride, 1) if conv1_stride else (1, stride)
                h1 = h // s1
                convs = [("quant_convbn1", cin, mid, 1, s1, h, h1, ab, cfg[p + "quant_act1"]),
                         ("quant_convbn2", mid, mid, 3, s2, h1, ho, cfg[p + "quant_act1"], cfg[p + "quant_act2"]),
                         ("quant_convbn3", mid, wd, 1, 1, ho, ho, cfg[p + "quant_act2"], None)]
            else:
                convs = [("quant_convbn1", cin, wd, 3, stride, h, ho, ab, cfg[p + "quant_act1"]),
                         ("quant_convbn2", wd, wd, 3, 1, ho, ho, cfg[p + "quant_act1"], None)]
            for cname, ci, co, k, st, hin, hout, in_b, out_b in convs:
                # a strided 1x1 conv reads only the pixels it uses; a strided 3x3 reads (almost) all of them
                pix_in = hout * hout if (k == 1 and st > 1) else hin * hin
                rd = ci * pix_in * in_b // 8
                wbytes = ci * k * k * co * cfg[p + cname] // 8 + co * 16
                mac = ci * k * k * co * hout * hout
                name = p + cname
                if out_b is not None:
                    wr = co * hout * hout * out_b // 8
                else:
                    last_unit = seen == n_units
                    nxt_resize = (not last_unit) and (ui == n - 1)  # the next unit opens a stage: it has an identity conv
                    nxt_ab = None if last_unit else cfg[(f"stage{si + 2}.unit1." if ui == n - 1 else f"stage{si + 1}.unit{ui + 2}.") + "quant_act"]
                    if resize:  # identity conv inside this launch: reads the block input (strided), its own weights
                        rd += cin * ho * ho * ab // 8
                        wbytes += cin * wd * cfg[p + "quant_identity_convbn"] // 8 + wd * 16
                        mac += cin * wd * ho * ho
                        name += "+identity"
                    else:
                        rd += co * hout * hout * 2          # uint16 residual in
                    wr = 0 if nxt_resize else co * hout * hout * 2   # uint16 residual out (not needed before a resize unit)
                    if nxt_ab is not None:
                        wr += co * hout * hout * nxt_ab // 8         # the next unit's block-input QuantAct output
                rows.append(dict(name=name, read=rd, write=wr, macs=mac, weight_bytes=wbytes))
            cin, h = wd, ho
    rows.append(dict(name="hawq_avgpool_requant", read=cin * h * h * 2, write=cin, macs=0, weight_bytes=0))
    rows.append(dict(name="quant_output", read=cin, write=1000 * 4, macs=cin * 1000, weight_bytes=cin * 1024 + 1024 * 8))
    return rows


def fused_plan_bytes(arch: str, scheme: str, batch: int, fused_pairs=()) -> int:
    """Bytes per launch of the plan.  `fused_pairs`: names of the expand convs ("stage1.unit2.quant_convbn3") whose launch also
    runs the next unit's reduce conv (hawq_conv_expand_reduce): that unit's 8-bit block input is then neither written nor read."""
    rows = fused_plan_table(arch, scheme)
    total = batch * sum(r["read"] + r["write"] for r in rows) + sum(r["weight_bytes"] for r in rows)
    names = [r["name"] for r in rows]
    for pair in fused_pairs:
        i = next(k for k, n in enumerate(names) if n.split("+")[0] == pair)
        total -= 2 * batch * rows[i + 1]["read"]   # the row after an expand conv is the next unit's quant_convbn1: its input is that tensor
    return total
