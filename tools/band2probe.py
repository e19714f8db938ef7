"""Round 5: the 3x3 kernels of band_v2.hip against the band kernels of rounds 1-4 on ResNet50's conv2 shapes (= ResNet18's 3x3
shapes): one launch each, planar input for both, outputs compared bit for bit (the reference is the first older tile that takes
the layer; the GPU tests compare with the CPU oracle), HIP events over 20 launches; HAWQ_DBG=128 + the probe library print the
s_memtime stamps.  Usage: python tools/band2probe.py [batch ...]   (PLANAR_OUT=1: planar output as well)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_conv_weight, pack_ctab, pack_w3x3_band
from hawq_amd.quant_utils import requant_table

L = lib.load()
rng = np.random.default_rng(0)
batches = [int(v) for v in sys.argv[1:]] or [64, 128]
n_g2 = L.hawq_conv2d_num_gemm2_tiles()
n_all, n_b1, n_b2 = L.hawq_conv2d_num_tiles() - n_g2, L.hawq_conv2d_num_band_tiles() - n_g2, L.hawq_conv2d_num_band2_tiles()
first_band = n_all - n_b1
planar_out = int(os.environ.get("PLANAR_OUT", "0"))
shapes = [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]
shapes = [shapes[int(i)] for i in os.environ.get("SHAPES", "0,1,2,3").split(",")]
# SWEEP_CIN="128,256,512,1024": slope / intercept probe - the selected maps with Cout fixed and Cin (= the number of K steps, 3 per 64 channels) swept
if os.environ.get("SWEEP_CIN"):
    shapes = [(h, int(c), cout) for (h, _, cout) in shapes for c in os.environ["SWEEP_CIN"].split(",")]
# SWEEP_COUT="64,128,256": the selected maps with Cin fixed and the number of 64-channel output tiles (= workgroups per pixel tile) swept - what one
# pixel tile's workgroups cost when 1, 2, 4 of them share the chip (profiles/r06_unit_fusion_analysis.md)
if os.environ.get("SWEEP_COUT"):
    shapes = [(h, cin, int(c)) for (h, cin, _) in shapes for c in os.environ["SWEEP_COUT"].split(",")]
for n in batches:
    for (h, cin, cout) in shapes:
        M = n * h * h
        x = rng.integers(0, 128, (M, cin)).astype(np.int8)
        xp = torch.from_numpy(np.ascontiguousarray(x.reshape(M, cin // 16, 16).transpose(1, 0, 2))).cuda()
        wt = rng.integers(-127, 128, (cout, cin, 3, 3)).astype(np.int64)
        b = rng.integers(-2000, 2000, cout).astype(np.int64)
        r = torch.from_numpy((rng.uniform(2e-5, 3e-4, cout) * 0.7).astype(np.float32))
        m, e = requant_table(torch.ones(1), r, torch.tensor([0.7]))
        w8 = pack_conv_weight(wt, 8)
        keep = [torch.from_numpy(w8).cuda(), torch.from_numpy(b.astype(np.int32)).cuda(),
                torch.from_numpy(pack_ctab(b, m, e)).cuda(), torch.from_numpy(m).cuda(), torch.from_numpy(e).cuda(),
                torch.from_numpy(pack_w3x3_band(w8, cout, cin)).cuda()]
        out = torch.zeros(M * cout, dtype=torch.uint8, device='cuda')
        a = lib.ConvArgs()
        a.in_, a.wgt, a.bias, a.wgt_band = xp.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[5].data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, h, h, cin, cout, 3, 3, 1, 1
        a.in_bits = a.w_bits = 8
        a.in_planar = 1
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = 1, 1, keep[3].data_ptr(), keep[4].data_ptr(), keep[2].data_ptr(), 1
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
        ref = None
        for bt in range(first_band, n_all):
            a.tile = bt + 1
            is_v2 = bt >= n_all - n_b2
            a.out_planar = planar_out if is_v2 else 0
            out.zero_()
            if L.hawq_conv2d(C.byref(a), None) != 0:
                continue
            torch.cuda.synchronize()
            got = out.clone()
            if a.out_planar:
                got = got.view(cout // 16, M, 16).permute(1, 0, 2).reshape(-1).contiguous()
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
            gmac = M * cout * cin * 9 / 1e9
            same = bool(torch.equal(got, ref))
            nbad = 0 if same else int((got != ref).sum())
            print(f"B={n} {h}x{h} C={cin}{'' if cin == cout else '->' + str(cout)} {'v2  ' if is_v2 else 'band'} tile {bt - first_band}: {us:7.1f} us  {gmac / us / 2.2 * 100:5.1f} % of 2.2 PMAC/s  same={same}"
                  + ("" if same else f" ({nbad} of {got.numel()} bytes differ)"), flush=True)
