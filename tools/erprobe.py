"""Fused expand->reduce launches at ResNet50 shapes (batch 128): microseconds per variant (HIP events, 20 launches) and,
with HAWQ_DBG=128, per-phase cycle stamps of one wave.   usage (GPU box): [HAWQ_DBG=128] python tools/erprobe.py [batch]"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_conv_weight, pack_ctab
from hawq_amd.quant_utils import requant_table
lib.load()
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ONLY = int(os.environ.get("ERPROBE_ONLY", 0))   # only the shape with this feature-map size, and only its fused variants (PMC runs)
SHAPES = ((56, 64, 256), (28, 128, 512), (14, 256, 1024), (7, 512, 2048))
if os.environ.get("ERPROBE_C3"):   # slope / intercept runs: the stage-3 pair with other numbers of 64-channel slices (C3 = 128 .. 2048)
    SHAPES = tuple((14, 256, int(v)) for v in os.environ["ERPROBE_C3"].split(","))
for (h, c, c3) in SHAPES:
    if ONLY and h != ONLY:
        continue
    M = N * h * h
    x2 = torch.from_numpy(rng.integers(0, 128, (M, c)).astype(np.int8)).cuda()
    w3 = rng.integers(-127, 128, (c3, c, 1, 1)).astype(np.int64); b3 = rng.integers(-2000, 2000, c3).astype(np.int64)
    w1 = rng.integers(-127, 128, (c, c3, 1, 1)).astype(np.int64); b1 = rng.integers(-2000, 2000, c).astype(np.int64)
    m3, e3 = requant_table(torch.ones(1), torch.from_numpy((rng.uniform(2e-3, 2e-2, c3) * 0.7).astype(np.float32)), torch.tensor([0.7]))
    m1, e1 = requant_table(torch.ones(1), torch.from_numpy((rng.uniform(2e-5, 3e-4, c) * 0.7).astype(np.float32)), torch.tensor([0.7]))
    mi, ei = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    keep = [torch.from_numpy(pack_conv_weight(w3, 8)).cuda(), torch.from_numpy(pack_conv_weight(w1, 8)).cuda(),
            torch.from_numpy(pack_ctab(b3, m3, e3)).cuda(), torch.from_numpy(pack_ctab(b1, m1, e1)).cuda(),
            torch.from_numpy(rng.integers(0, 20000, M * c3).astype(np.uint16)).cuda(), torch.zeros(M * c3, dtype=torch.uint16, device='cuda'),
            torch.zeros(M * c, dtype=torch.uint8, device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda'),
            torch.from_numpy(b3.astype(np.int32)).cuda(), torch.from_numpy(m3).cuda(), torch.from_numpy(e3).cuda()]
    a = lib.ExpandReduceArgs()
    ex, rd = a.expand, a.reduce
    ex.in_, ex.wgt, ex.bias, ex.m, ex.e = x2.data_ptr(), keep[0].data_ptr(), keep[8].data_ptr(), keep[9].data_ptr(), keep[10].data_ptr()
    ex.N, ex.H, ex.W, ex.Cin, ex.Cout, ex.KH, ex.KW, ex.stride, ex.pad = N, h, h, c, c3, 1, 1, 1, 0
    ex.in_bits = ex.w_bits = 8; ex.epilogue = lib.EPI_RESIDUAL; ex.ctab, ex.flags, ex.fast_tables = keep[2].data_ptr(), keep[7].data_ptr(), 1
    ex.res_in, ex.res_in_bits, ex.m_id_scalar, ex.e_id_scalar = keep[4].data_ptr(), 16, int(mi[0]), int(ei[0])
    ex.res_out, ex.res_out_bits, ex.out_bits, ex.q_lo, ex.q_hi, ex.mq, ex.eq = keep[5].data_ptr(), 16, 8, 0, 127, int(mq[0]), int(eq[0])
    rd.wgt, rd.bias, rd.m, rd.e = keep[1].data_ptr(), keep[8].data_ptr(), keep[9].data_ptr(), keep[10].data_ptr()
    rd.N, rd.H, rd.W, rd.Cin, rd.Cout, rd.KH, rd.KW, rd.stride, rd.pad = N, h, h, c3, c, 1, 1, 1, 0
    rd.in_bits = rd.w_bits = 8; rd.epilogue, rd.relu, rd.ctab, rd.fast_tables = lib.EPI_REQUANT, 1, keep[3].data_ptr(), 1
    rd.out_q, rd.out_bits, rd.q_lo, rd.q_hi = keep[6].data_ptr(), 8, -128, 127
    def timeit(fn, reps=20):
        fn()
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1_.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1_) / reps * 1e3
    sp = torch.cuda.current_stream().cuda_stream
    if not os.environ.get("HAWQ_DBG") and not ONLY:
        # the same two layers as separate hawq_conv2d launches (best tile each), then the expand conv alone on the wave-private kernel
        qbuf = torch.zeros(M * c3, dtype=torch.uint8, device='cuda')
        ex.out_q = qbuf.data_ptr()
        best = {}
        for name, conv in (("expand", ex), ("reduce", rd)):
            if name == "reduce":
                rd.in_ = qbuf.data_ptr()
            for tile in range(1, lib.load().hawq_conv2d_num_tiles() + 1):
                conv.tile = tile
                if lib.load().hawq_conv2d(C.byref(conv), sp) != 0:
                    continue
                us = timeit(lambda: lib.load().hawq_conv2d(C.byref(conv), sp), 10)
                if name not in best or us < best[name][1]:
                    best[name] = (tile, us)
            conv.tile = 0
        print(f"N={N} {h}x{h} C={c} C3={c3} separate launches: expand tile {best['expand'][0]} {best['expand'][1]:.1f} us + reduce tile {best['reduce'][0]} {best['reduce'][1]:.1f} us", flush=True)
        solo = lib.ExpandReduceArgs()
        C.memmove(C.byref(solo.expand), C.byref(ex), C.sizeof(ex))
        for tile in range(1, lib.load().hawq_conv_expand_reduce_variants(C.byref(solo)) + 1):
            solo.tile = tile
            print(f"N={N} {h}x{h} C={c} C3={c3} expand alone, wave-private variant {tile}: {timeit(lambda: lib.call('hawq_conv_expand_reduce', C.byref(solo), sp)):.1f} us", flush=True)
    for tile in range(1, lib.load().hawq_conv_expand_reduce_variants(C.byref(a)) + 1):
        a.tile = tile
        lib.call("hawq_conv_expand_reduce", C.byref(a), None)
        if os.environ.get("HAWQ_DBG") and not os.environ.get("ERPROBE_TIME"):   # stamp runs print from inside the library
            continue
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.call("hawq_conv_expand_reduce", C.byref(a), torch.cuda.current_stream().cuda_stream)
        e1_.record(); torch.cuda.synchronize()
        print(f"N={N} {h}x{h} C={c} C3={c3} variant {tile}: {e0.elapsed_time(e1_) / 20 * 1e3:.1f} us")
