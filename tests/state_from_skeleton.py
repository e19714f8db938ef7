"""Build the oracle's float-state dict straight from a float skeleton + a bit schedule, i.e.
without any quantized-module implementation (neither the reference's nor hawq_amd's)."""
import numpy as np

from hawq_amd.skeleton import ARCH

f32 = np.float32


def _np(t):
    return t.detach().cpu().numpy()


def state_from_skeleton(fl, arch, cfg):
    units, widths, bottleneck, _ = ARCH[arch]

    def act(name):
        bits = cfg[name]
        return dict(bits=bits, mode="asymmetric" if bits == 4 else "symmetric",
                    x_min=np.zeros(1, f32), x_max=np.zeros(1, f32))

    def convbn(blk, name):
        c, b = blk.conv, blk.bn
        return dict(bits=cfg[name], w=_np(c.weight), gamma=_np(b.weight), beta=_np(b.bias), mean=_np(b.running_mean),
                    var=_np(b.running_var), eps=float(b.eps), stride=int(c.stride[0]), pad=int(c.padding[0]))

    st = dict(bottleneck=bottleneck, quant_input=act("quant_input"))
    st["stem"] = convbn(fl.features.init_block.conv, "quant_init_convbn" if bottleneck else "quant_init_block_convbn")
    st["quant_act_int32"] = act("quant_act_int32")
    st["units"] = []
    for si, n in enumerate(units):
        for ui in range(n):
            u = getattr(getattr(fl.features, f"stage{si + 1}"), f"unit{ui + 1}")
            p = f"stage{si + 1}.unit{ui + 1}"
            d = dict(name=p, resize=bool(u.resize_identity), quant_act=act(p + ".quant_act"),
                     convbn1=convbn(u.body.conv1, p + ".quant_convbn1"), quant_act1=act(p + ".quant_act1"),
                     convbn2=convbn(u.body.conv2, p + ".quant_convbn2"), quant_act_int32=act(p + ".quant_act_int32"))
            if bottleneck:
                d["quant_act2"] = act(p + ".quant_act2")
                d["convbn3"] = convbn(u.body.conv3, p + ".quant_convbn3")
            if d["resize"]:
                d["identity"] = convbn(u.identity_conv, p + ".quant_identity_convbn")
            st["units"].append(d)
    st["quant_act_output"] = act("quant_act_output")
    st["fc"] = dict(bits=cfg["quant_output"], w=_np(fl.output.weight), b=_np(fl.output.bias))
    return st
