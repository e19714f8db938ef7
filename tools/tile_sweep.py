"""Whole-network tile sweep: for every conv launch of the fused plan and every tile id the library accepts for it,
run the network with only that launch's tile changed and compare the logits with the all-default plan (bit-exact).
usage (GPU box): python tools/tile_sweep.py [arch scheme batch ...]"""
import sys
import torch
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
    print(f"batch {b}: swept {len(eng._conv_args)} launches x {ntiles} tiles")
print("mismatches:", bad)
