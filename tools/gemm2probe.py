"""Round 5: the streaming 1x1 kernels of gemm_v2.hip against the general tiles of conv_kernel on ResNet50's long-K 1x1 layers (reduce
convs, stride 1 / 2; REQUANT): one launch each, outputs compared byte for byte, HIP events over 20 launches; HAWQ_DBG=128 + the probe
library print the s_memtime stamps.  Usage: python tools/gemm2probe.py [batch ...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_conv_weight, pack_ctab, pack_w1x1_k128
from hawq_amd.quant_utils import requant_table

L = lib.load()
rng = np.random.default_rng(0)
batches = [int(v) for v in sys.argv[1:]] or [64]
n_all, n_special, n_g2 = L.hawq_conv2d_num_tiles(), L.hawq_conv2d_num_band_tiles(), L.hawq_conv2d_num_gemm2_tiles()
tiles = list(range(1, n_all - n_special + 1)) + list(range(n_all - n_g2 + 1, n_all + 1))
# (input map, Cin, Cout, stride): stage2.u1 / stage3.u1 / stage4.u1 conv1 (strided), stage4.u2 conv1, stage3.u2 conv1
shapes = [(56, 256, 128, 2), (28, 512, 256, 2), (14, 1024, 512, 2), (7, 2048, 512, 1), (14, 1024, 256, 1)]
shapes = [shapes[int(i)] for i in os.environ.get("SHAPES", "0,1,2,3,4").split(",")]
for n in batches:
    for (h, cin, cout, s) in shapes:
        ho = (h - 1) // s + 1
        M = n * ho * ho
        x = torch.from_numpy(rng.integers(0, 128, (n * h * h, cin)).astype(np.int8)).cuda()
        wt = rng.integers(-127, 128, (cout, cin, 1, 1)).astype(np.int64)
        b = rng.integers(-2000, 2000, cout).astype(np.int64)
        r = torch.from_numpy((rng.uniform(2e-5, 3e-4, cout) * 0.7).astype(np.float32))
        m, e = requant_table(torch.ones(1), r, torch.tensor([0.7]))
        w8 = pack_conv_weight(wt, 8)
        keep = [torch.from_numpy(w8).cuda(), torch.from_numpy(b.astype(np.int32)).cuda(), torch.from_numpy(pack_ctab(b, m, e)).cuda(),
                torch.from_numpy(m).cuda(), torch.from_numpy(e).cuda(), torch.from_numpy(pack_w1x1_k128(w8, cout, cin)).cuda()]
        out = torch.zeros(M * cout, dtype=torch.uint8, device='cuda')
        a = lib.ConvArgs()
        a.in_, a.wgt, a.bias, a.wgt_k128 = x.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[5].data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, h, h, cin, cout, 1, 1, s, 0
        a.in_bits = a.w_bits = 8
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = 1, 1, keep[3].data_ptr(), keep[4].data_ptr(), keep[2].data_ptr(), 1
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
        ref, res = None, []
        for tile in tiles:
            a.tile = tile
            out.zero_()
            if L.hawq_conv2d(C.byref(a), None) != 0:
                continue
            torch.cuda.synchronize()
            got = out.clone()
            if ref is None:
                ref = got
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                L.hawq_conv2d(C.byref(a), None)
            e0.record()
            for _ in range(20):
                L.hawq_conv2d(C.byref(a), None)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 50, tile, bool(torch.equal(got, ref))))
        old = min(r for r in res if r[1] <= n_all - n_special)
        print(f"B={n} {h}x{h}/{s} {cin}->{cout}: best general tile {old[1]}: {old[0]:6.1f} us | " +
              " ".join(f"g2 tile {t}: {us:6.1f} us same={sm}" for us, t, sm in res if t > n_all - n_special), flush=True)
