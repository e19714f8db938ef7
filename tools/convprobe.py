"""HAWQ_DBG=128 cycle stamps (prologue | K loop | epilogue of one wave) of single conv launches at ResNet50 shapes.
usage (GPU box): HAWQ_DBG=128 python tools/convprobe.py"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd import _lib as lib
from hawq_amd.packing import pack_conv_weight, pack_ctab
from hawq_amd.quant_utils import requant_table
lib.load()
rng = np.random.default_rng(0)
# n, h, w, cin, cout, k, stride, epilogue (1 REQUANT, 2 RESIDUAL), tiles to try
CASES = [(128, 14, 14, 1024, 256, 1, 1, 1, (14, 8, 1, 12)), (128, 28, 28, 512, 128, 1, 1, 1, (14, 12, 10)),
         (128, 7, 7, 2048, 512, 1, 1, 1, (9, 14, 12)), (128, 14, 14, 256, 1024, 1, 1, 2, (12, 13, 1)),
         (128, 28, 28, 128, 512, 1, 1, 2, (12, 13)), (128, 56, 56, 64, 256, 1, 1, 2, (12, 13))]
for (n, h, w, cin, cout, k, st, epi, tiles) in CASES:
    x = torch.from_numpy(rng.integers(0, 128, (n, h, w, cin)).astype(np.int8)).cuda()
    wt = rng.integers(-127, 128, (cout, cin, k, k)).astype(np.int64)
    b = rng.integers(-2000, 2000, cout).astype(np.int64)
    r = torch.from_numpy((rng.uniform(2e-5, 3e-4, cout) * 0.7).astype(np.float32))
    m, e = requant_table(torch.ones(1), r, torch.tensor([0.7]))
    wd = torch.from_numpy(pack_conv_weight(wt, 8)).cuda(); bd = torch.from_numpy(b.astype(np.int32)).cuda()
    ct = torch.from_numpy(pack_ctab(b, m, e)).cuda(); md = torch.from_numpy(m).cuda(); ed = torch.from_numpy(e).cuda()
    M = n * h * w
    out = torch.zeros(M * cout, dtype=torch.uint8, device='cuda')
    res_in = torch.from_numpy(rng.integers(0, 30000, M * cout).astype(np.uint16)).cuda()
    res_out = torch.zeros(M * cout, dtype=torch.uint16, device='cuda')
    flags = torch.zeros(1, dtype=torch.int32, device='cuda')
    m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    for tile in tiles:
        a = lib.ConvArgs()
        a.in_, a.wgt, a.bias = x.data_ptr(), wd.data_ptr(), bd.data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, h, w, cin, cout, k, k, st, 0
        a.in_bits = a.w_bits = 8; a.tile = tile; a.epilogue = epi; a.relu = 1
        a.m, a.e, a.ctab, a.fast_tables, a.flags = md.data_ptr(), ed.data_ptr(), ct.data_ptr(), 1, flags.data_ptr()
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, (-128 if epi == 1 else 0), 127
        if epi == 2:
            a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = res_in.data_ptr(), 16, int(m1[0]), int(e1[0])
            a.res_out, a.res_out_bits, a.mq, a.eq = res_out.data_ptr(), 16, int(mq[0]), int(eq[0])
        for _ in range(2):
            lib.call("hawq_conv2d", C.byref(a), None)
        torch.cuda.synchronize()
