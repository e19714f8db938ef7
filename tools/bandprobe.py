"""Time every 3x3 band tile on the ResNet50 conv2 shapes (one launch each, HIP events over 20 launches), then one
HAWQ_DBG=128 launch per (shape, tile) for the per-phase cycle stamps.  Usage: python tools/bandprobe.py [batch ...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_conv_weight, pack_ctab
from hawq_amd.quant_utils import requant_table

L = lib.load()
rng = np.random.default_rng(0)
batches = [int(v) for v in sys.argv[1:]] or [64, 128]
nplain = L.hawq_conv2d_num_tiles() - L.hawq_conv2d_num_band_tiles()
for n in batches:
    for (h, cin) in [(56, 64), (28, 128), (14, 256), (7, 512)][:int(os.environ.get('NSHAPES', '4'))]:
        cout = cin
        x = torch.from_numpy(rng.integers(0, 128, (n, h, h, cin)).astype(np.int8)).cuda()
        wt = rng.integers(-127, 128, (cout, cin, 3, 3)).astype(np.int64)
        b = rng.integers(-2000, 2000, cout).astype(np.int64)
        r = torch.from_numpy((rng.uniform(2e-5, 3e-4, cout) * 0.7).astype(np.float32))
        m, e = requant_table(torch.ones(1), r, torch.tensor([0.7]))
        keep = [torch.from_numpy(pack_conv_weight(wt, 8)).cuda(), torch.from_numpy(b.astype(np.int32)).cuda(),
                torch.from_numpy(pack_ctab(b, m, e)).cuda(), torch.from_numpy(m).cuda(), torch.from_numpy(e).cuda()]
        out = torch.zeros(n * h * h * cout, dtype=torch.uint8, device='cuda')
        a = lib.ConvArgs()
        a.in_, a.wgt, a.bias = x.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, h, h, cin, cout, 3, 3, 1, 1
        a.in_bits = a.w_bits = 8
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = 1, 1, keep[3].data_ptr(), keep[4].data_ptr(), keep[2].data_ptr(), 1
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
        ref = None
        for bt in range(L.hawq_conv2d_num_tiles() - nplain):
            a.tile = nplain + bt + 1
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
            us = e0.elapsed_time(e1) * 50
            gmac = n * h * h * cout * cin * 9 / 1e9
            print(f"B={n} {h}x{h} C={cin} band tile {bt}: {us:7.1f} us  {gmac / us / 2.2 * 100:5.1f} % of 2.2 PMAC/s  same={bool(torch.equal(got, ref))}", flush=True)
