"""The fused stem launch alone (fp32 and uint8 input), microseconds at the given batches; with the probe library and HAWQ_DBG bits
(16 = no input loads, 32 = one of three window rows, 64 = no epilogue) it bounds what each phase costs.
usage (GPU box): [HAWQ_LIB=.../libhawq_mi355_ablate.so HAWQ_DBG=16] python tools/stemprobe.py [batch ...]"""
import sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_stem_weight
from hawq_amd.quant_utils import requant_table
lib.load()
rng = np.random.default_rng(0)
w = torch.from_numpy(pack_stem_weight(rng.integers(-127, 128, (64, 3, 7, 7)).astype(np.int64))).cuda()
bias = torch.from_numpy(rng.integers(-2000, 2000, 64).astype(np.int32)).cuda()
m, e = requant_table(torch.tensor([0.02]), torch.from_numpy(rng.uniform(2e-3, 2e-2, 64).astype(np.float32)), torch.tensor([0.003]), vbits=24)
md, ed = torch.from_numpy(m).cuda(), torch.from_numpy(e).cuda()
mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
lut = torch.from_numpy(rng.integers(-128, 128, (3, 256)).astype(np.int8)).cuda()
sp = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N in [int(v) for v in sys.argv[1:]] or [64, 128]:
    x = torch.randn(N, 3, 224, 224, device="cuda")
    xu8 = torch.randint(0, 256, (N, 224, 224, 3), dtype=torch.uint8, device="cuda")
    res = torch.zeros(N * 56 * 56 * 64, dtype=torch.uint16, device="cuda")
    q = torch.zeros(N * 56 * 56 * 64, dtype=torch.uint8, device="cuda")
    f32 = lambda: lib.call("hawq_stem_fused", x.data_ptr(), N, 3, 224, 224, 50.0, -128, 127, w.data_ptr(), bias.data_ptr(), md.data_ptr(), ed.data_ptr(),
                           -32768, 32767, res.data_ptr(), q.data_ptr(), 8, int(mq[0]), int(eq[0]), 0, 127, 1, sp)
    u8 = lambda: lib.call("hawq_stem_fused_u8", xu8.data_ptr(), lut.data_ptr(), N, 3, 224, 224, w.data_ptr(), bias.data_ptr(), md.data_ptr(), ed.data_ptr(),
                          -32768, 32767, res.data_ptr(), q.data_ptr(), 8, int(mq[0]), int(eq[0]), 0, 127, 1, sp)
    print(f"N={N}: stem fp32 input {timeit(f32):.1f} us, uint8 input {timeit(u8):.1f} us", flush=True)
