"""MobileNetV2 (w1, W8A8) fused integer plan, launch by launch: microseconds of every launch of ONE chain over the whole batch (HIP events
around eager launches, 20 repetitions) beside the bytes it moves as stored (channels padded to 64) and at the network's true widths.
usage (GPU box): python tools/mbv2_perop.py [batch]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd.api import build_quantized_model, calibrate
from hawq_amd.engine_mbv2 import MobileNetV2Engine
from hawq_amd.skeleton import synthetic_images

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = build_quantized_model("mobilenetv2_w1", "uniform8", seed=0).cuda()
calibrate(model, synthetic_images(8, seed=0).cuda())
eng = MobileNetV2Engine(model, chains=1)
x = synthetic_images(N, seed=1).cuda()
eng(x)
torch.cuda.synchronize()
rows, total = [], 0.0
with torch.cuda.stream(eng.stream):
    for i, op in enumerate(eng._ops):
        name = op.args[0] if hasattr(op, "args") else "op"
        op()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        for _ in range(20):
            op()
        e1.record(eng.stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        total += us
        extra = ""
        if name in ("hawq_conv2d",):
            a = op.args[1]._obj
            extra = f"M={a.N * a.H * a.W // (a.stride * a.stride)} Cin={a.Cin} Cout={a.Cout} k={a.KH} s={a.stride} epi={a.epilogue} tile={a.tile} fast={a.fast_tables} n_valid={a.n_valid}"
        elif name == "hawq_linear_bottleneck":
            b = op.args[1]._obj
            extra = (f"N={b.expand.N} {b.expand.H}x{b.expand.W} in_pitch={b.expand.in_pitch or 64} hidden={b.c_mid} stride={b.dw_stride} "
                     f"out_pitch={b.project.out_pitch or 64} identity={int(bool(b.project.res_in))} carrier={int(bool(b.project.res_out))}")
        elif name == "hawq_depthwise3x3_requant":
            _, _x, _w, _b, _m, _e, n, h, w, c, cv, s = op.args[:12]
            extra = f"N={n} {h}x{w} C={c} (valid {cv}) stride={s}"
        rows.append((i, name, us, extra))
if "--preshift" in sys.argv:   # how many per-channel requant tables of each unit carry a pre-shift k > 0 (conv1 / depthwise / conv3)
    for ui, u in enumerate(eng.P['units']):
        def nk(t):
            return "-" if t is None else int(((t.view(-1, 4)[:, 1] >> 8) > 0).sum().item())
        e1, e2 = u['layers'][0], u['layers'][1]
        print(f"unit {ui + 1}: k > 0 in conv1 {nk(e1.get('ctab'))}, depthwise {nk(e2.get('dw_ctab'))}, conv3 {nk(u['proj']['fast']['ctab'] if u['proj']['fast'] else None)}; "
              f"scalars q_fast {u['q_fast']}, id_fast {u.get('id_fast')}")
for i, name, us, extra in rows:
    print(f"{i:3d} {name:28s} {us:7.1f} us  {extra}")
print(f"sum of launches {total:.1f} us for batch {N} (one chain); {eng.n_fused_units} one-launch units; tiles {eng.tile_choice}")
