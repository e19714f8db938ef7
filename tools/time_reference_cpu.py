"""Wall time of the UNMODIFIED reference's frozen fake-quant forward on this host's CPU cores (build container only: needs
/root/reference) - SURVEY.md 8(d)'s CPU baseline protocol: ResNet50 uniform8, batch 128, 1 warm-up + 3 timed forwards.
The number that bench.py reports on the GPU box comes from oracle/fakequant_port.py (the reference cannot travel); this
script records what the real thing does here, beside the port on the same host.   usage: python tools/time_reference_cpu.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawq_amd.skeleton import synthetic_images
from oracle import fakequant_port, oracle, ref_live

arch, scheme, batch = "resnet50", "uniform8", 128
q = ref_live.build_reference_model(arch, scheme, seed=0)
ref_live.calibrate_and_freeze(q, synthetic_images(8, seed=0))
x = synthetic_images(batch, seed=1)
rows = []
with torch.no_grad():
    q(x[:2])
    for i in range(4):
        t0 = time.perf_counter(); y = q(x); dt = time.perf_counter() - t0
        rows.append(("reference (unmodified /root/reference modules)", "warm-up" if i == 0 else f"run {i}", dt))
from hawq_amd.api import build_quantized_resnet
st = oracle.extract_float_state(build_quantized_resnet(arch, scheme, seed=0))
oracle.forward_int(st, synthetic_images(8, seed=0).numpy(), calibrate=True)
fakequant_port.forward(st, x[:2])
for i in range(4):
    t0 = time.perf_counter(); y2 = fakequant_port.forward(st, x); dt = time.perf_counter() - t0
    rows.append(("port (oracle/fakequant_port.py)", "warm-up" if i == 0 else f"run {i}", dt))
print(f"# CPU fake-quant forward, {arch} {scheme}, batch {batch}, {torch.get_num_threads()} threads ({os.cpu_count()} cores), torch {torch.__version__}\n")
print("| path | run | seconds | images/s |\n|---|---|---|---|")
for p, r, dt in rows:
    print(f"| {p} | {r} | {dt:.2f} | {batch / dt:.2f} |")
print(f"\nlogits of the two paths bit-equal: {bool(torch.equal(y, y2))}")
