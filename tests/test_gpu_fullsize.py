"""Full-size layer cases (SURVEY.md App. B shapes at batch 128) of every conv kernel class against the C oracle:
the int32 PRE-REQUANT accumulators (EPI_RAW through the same operand pipeline: int8 x int8 launches take the
asynchronous global_load_lds K-pipeline whatever the epilogue) and the fused requantised / residual outputs of the
kernels the benchmark actually launches (every applicable tile id, 3x3 band kernels included).
Bit-exact: np.array_equal on integers.  Reference lines: quant_modules.py:489-494, quant_utils.py:390-456."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from tests.test_gpu_kernels import (conv_args, dev, from_planar, lib, make_conv, nhwc, odyadic, orc, pack_act,  # noqa: F401
                                    rand_tables, stream, to_planar, unpack_q)

pytestmark = pytest.mark.gpu


def _tiles(lib):
    n, nb = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_band_tiles()
    return list(range(1, n - nb + 1)), list(range(n - nb + 1, n + 1))


def _try(lib, a):
    """Launch; False if this tile id does not apply to the layer (refused by the launcher, never mis-computed)."""
    return lib.load().hawq_conv2d(C.byref(a), stream()) == 0


@pytest.mark.parametrize("name,shape", [
    ("reduce 1x1 256->64 @56^2 (M = 401 408)", (128, 56, 56, 256, 64, 1, 1, 0)),
    ("3x3 64->64 @56^2 (M = 401 408, K = 576)", (128, 56, 56, 64, 64, 3, 1, 1)),
    ("3x3 256->256 @14^2 (K = 2304)", (128, 14, 14, 256, 256, 3, 1, 1)),
    ("3x3 512->512 @7^2 (K = 4608)", (128, 7, 7, 512, 512, 3, 1, 1)),
    ("strided reduce 1x1 1024->512 /2 @14^2", (128, 14, 14, 1024, 512, 1, 2, 0)),
])
@pytest.mark.parametrize("bits", [8, 4])
def test_full_size_requant_layers(lib, orc, name, shape, bits):
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast, tables_fit_fast
    n, h, w, cin, cout, k, stride, pad = shape
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x, wt, b = make_conv(rng, n, h, w, cin, cout, k, bits, bits)
    acc = orc.conv2d(x, wt, b, stride, pad)
    ho, wo = acc.shape[2], acc.shape[3]
    lo, hi = (0, 127) if bits == 8 else (0, 15)
    sd = float(acc.std())  # per-channel ratios that spread the outputs over the clamp range instead of saturating them
    m, e = rand_tables(rng, cout, 0.2 * hi / sd, 0.7 * hi / sd)
    vb = int(np.abs(acc).max()).bit_length() + 1
    assert tables_fit_fast(m, e, vb)
    mode = 1 if tables_are_fast(m, e, vb) else 5
    ref_q = odyadic(orc, np.maximum(acc, 0), m, e, (lo, hi))
    generic, band = _tiles(lib)
    # 1. raw accumulators (heuristic tile + two autotuner favourites)
    for tile in (0, 11, 14):
        a, keep = conv_args(lib, x, wt, b, stride, pad, bits, bits, tile=tile)
        out = torch.full((acc.size,), -7, dtype=torch.int32, device='cuda')
        a.epilogue, a.out_acc = lib.EPI_RAW, out.data_ptr()
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out.cpu().numpy().reshape(n, ho, wo, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, acc), f"{name}: raw accumulators, tile {tile}"
        del out, keep
    # 2. fused requant epilogue, every tile id that takes the layer
    ran = 0
    for tile in [0] + generic + band:
        a, keep = conv_args(lib, x, wt, b, stride, pad, bits, bits, tile=tile)
        keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
        out = torch.zeros(acc.size * bits // 8, dtype=torch.uint8, device='cuda')
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = (lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(),
                                                                keep['ctab'].data_ptr(), mode)
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), bits, lo, hi
        if not _try(lib, a):
            assert tile in band, f"{name}: generic tile {tile} refused"
            continue
        assert np.array_equal(unpack_q(out, (n, ho, wo, cout), bits), ref_q), f"{name}: requant, tile {tile}"
        if tile in band:  # as the engine launches it: activations in channel-group planes
            keep['xp'] = dev(to_planar(pack_act(x, bits)))
            a.in_, a.in_planar = keep['xp'].data_ptr(), 1
            out.zero_()
            lib.call("hawq_conv2d", C.byref(a), stream())
            assert np.array_equal(unpack_q(out, (n, ho, wo, cout), bits), ref_q), f"{name}: planar input, tile {tile}"
        elif tile in (11, 12, 14):  # conv1 -> conv2 tensors are written as planes
            a.out_planar = 1
            out.zero_()
            lib.call("hawq_conv2d", C.byref(a), stream())
            assert np.array_equal(from_planar(out, (n, ho, wo, cout), bits), ref_q), f"{name}: planar output, tile {tile}"
        ran += 1
        del out, keep
    assert ran >= len(generic) + 1 + (1 if k == 3 and (bits == 8 or cin % 128 == 0) else 0)
    frac_sat = float((ref_q == hi).mean())
    assert frac_sat < 0.35, frac_sat


@pytest.mark.parametrize("name,shape,dual", [
    ("expand 1x1 64->256 @56^2 + uint16 residual", (128, 56, 56, 64, 256), None),
    ("expand 1x1 256->1024 @14^2 + uint16 residual", (128, 14, 14, 256, 1024), None),
    ("expand 1x1 512->2048 @7^2 + identity conv 1024->2048 /2 in the same launch", (128, 7, 7, 512, 2048), 1024),
])
def test_full_size_residual_layers(lib, orc, name, shape, dual):
    from hawq_amd.packing import pack_conv_weight, pack_ctab
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    acc = orc.conv2d(x, wt, b, 1, 0)
    sd = float(acc.std())
    m2, e2 = rand_tables(rng, cout, 500 / sd, 4000 / sd)
    if dual:
        x2, w2, b2 = make_conv(rng, n, 2 * h, 2 * w, dual, cout, 1, 8, 8)
        acc_id = orc.conv2d(x2, w2, b2, 2, 0)
        sd2 = float(acc_id.std())
        m1, e1 = rand_tables(rng, cout, 500 / sd2, 4000 / sd2)
        idq = odyadic(orc, acc_id, m1, e1)
        fast_id = tables_are_fast(m1, e1, int(np.abs(acc_id).max()).bit_length() + 1)
    else:
        res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
        m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
        idq = odyadic(orc, res, m1, e1)
        fast_id = True
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + idq, 0)
    assert ref_res.max() < 65536
    mode = 1 if fast_id and tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1) else 5  # 5 = exact-tie kernels
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    generic, _ = _tiles(lib)
    for tile in [0] + generic:
        a, keep = conv_args(lib, x, wt, b, 1, 0, 8, 8, tile=tile)
        keep.update(ctab=dev(pack_ctab(b, m2, e2)), m=dev(m2), e=dev(e2))
        if dual:
            keep.update(x2=dev(nhwc(x2).astype(np.int8).view(np.uint8)), w2=dev(pack_conv_weight(w2, 8)),
                        b2=dev(b2.astype(np.int32)), m1=dev(m1), e1=dev(e1), ctab_id=dev(pack_ctab(b2, m1, e1)))
            a.in2, a.wgt2, a.bias2 = keep['x2'].data_ptr(), keep['w2'].data_ptr(), keep['b2'].data_ptr()
            a.H2, a.W2, a.Cin2, a.stride2, a.in2_bits, a.w2_bits = 2 * h, 2 * w, dual, 2, 8, 8
            a.m_id, a.e_id, a.ctab_id = keep['m1'].data_ptr(), keep['e1'].data_ptr(), keep['ctab_id'].data_ptr()
        else:
            keep['res'] = dev(nhwc(res).astype(np.uint16))
            a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 16, int(m1[0]), int(e1[0])
        flags = torch.zeros(1, dtype=torch.int32, device='cuda')
        out_res = torch.zeros(ref_res.size, dtype=torch.uint16, device='cuda')
        out_q = torch.zeros(ref_res.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.m, a.e, a.ctab, a.flags = lib.EPI_RESIDUAL, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), flags.data_ptr()
        a.res_out, a.res_out_bits = out_res.data_ptr(), 16
        a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, 0, 127, int(mq[0]), int(eq[0])
        a.fast_tables = mode
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref_res), f"{name}: residual, tile {tile}"
        assert np.array_equal(unpack_q(out_q, (n, h, w, cout), 8), ref_q), f"{name}: next QuantAct, tile {tile}"
        assert flags.item() == 0
        del out_res, out_q, keep
