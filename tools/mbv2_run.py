"""MobileNetV2 w1 / uniform8 fused integer plan: build, tune, then `steps` graph replays of one batch (for rocprofv3 runs: the kernel
trace and the FETCH_SIZE / WRITE_SIZE passes of tools/mbv2_profile.sh).   usage (GPU box): python tools/mbv2_run.py [batch] [steps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd.api import build_quantized_model, calibrate
from hawq_amd.engine_mbv2 import MobileNetV2Engine
from hawq_amd.skeleton import synthetic_images

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = build_quantized_model("mobilenetv2_w1", "uniform8", seed=0).cuda()
calibrate(model, synthetic_images(8, seed=0).cuda())
eng = MobileNetV2Engine(model, chains=1)
eng(synthetic_images(N, seed=1).cuda())
torch.cuda.synchronize()
with torch.cuda.stream(eng.stream):
    for _ in range(steps):
        eng.run_resident()
torch.cuda.synchronize()
print(f"batch {N}: {steps} replays; {eng.n_launches} launches per forward, {eng.n_fused_units} one-launch units, plan bytes per image {eng.total_plan_bytes // N}")
