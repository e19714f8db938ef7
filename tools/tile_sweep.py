"""Whole-network tile sweep: for every conv launch of the fused plan and every tile id the library accepts for it,
run the network with only that launch's tile changed and compare the logits with the all-default plan (bit-exact).
usage (GPU box): python tools/tile_sweep.py [arch scheme batch ...]"""
import os
import sys
import torch
os.environ["HAWQ_KEEP_PACKED"] = "1"   # tiles are switched after the build: keep every layer's packed weight streams
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd import _lib
from hawq_amd.api import build_quantized_resnet, calibrate
from hawq_amd.engine import IntegerEngine
from hawq_amd.skeleton import synthetic_images

args = sys.argv[1:] or ["resnet18", "uniform8", "6"]
arch, scheme, batches = args[0], args[1], [int(b) for b in args[2:]]
model = build_quantized_resnet(arch, scheme, seed=0).cuda()
calibrate(model, synthetic_images(2, seed=0).cuda())
ntiles = _lib.load().hawq_conv2d_num_tiles()
bad = 0
for b in batches:
    x = (synthetic_images(b, seed=3) * 1.3).cuda()
    eng = IntegerEngine(model, use_graph=False, autotune=False, chains=1)
    ref = eng(x).clone()
    for i, (name, a) in enumerate(zip(eng._conv_names, eng._conv_args)):
        for t in range(1, ntiles + 1):
            a.tile = t
            try:
                y = eng(x)
            except RuntimeError:
                continue  # tile does not apply to this layer
            if not torch.equal(y, ref):
                bad += 1
                print(f"MISMATCH batch {b} {name} tile {t}: max |diff| {float((y - ref).abs().max()):.4g}")
        a.tile = 0
    # every variant of both fused kernel families for every expand(-> reduce) launch of the plan, and its unfused form
    import ctypes as C
    nv_total = 0
    for name, pair in zip(eng._er_names, eng._er_args):
        nvar = _lib.load().hawq_conv_expand_reduce_variants(C.byref(pair.er))
        for v in range(0, nvar + 1):
            pair.er.tile, pair.fused = v, True
            y = eng(x)
            nv_total += 1
            if not torch.equal(y, ref):
                bad += 1
                print(f"MISMATCH batch {b} {name} fused variant {v}: max |diff| {float((y - ref).abs().max()):.4g}")
        pair.fused = False
        if not torch.equal(eng(x), ref):
            bad += 1
            print(f"MISMATCH batch {b} {name} as separate launches")
        pair.er.tile, pair.fused = 0, True
    print(f"batch {b}: swept {len(eng._conv_args)} launches x {ntiles} tiles, {len(eng._er_args)} expand(-> reduce) launches x their variants ({nv_total} runs)")
print("mismatches:", bad)
