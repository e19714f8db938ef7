"""Round 5: the 3x3 kernels of band_v2.hip (the last hawq_conv2d_num_band2_tiles() tile ids) against the CPU oracle, through
the C ABI.  Bit-exact: int32 accumulators are never exposed by these kernels, so the checks are on the requantised int8 outputs
and the un-clamped uint16 residuals, which the oracle derives from its exact accumulators (quant_modules.py:489-494,
quant_utils.py:390-456; q_resnet.py:241-243, 300-316)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_gpu_kernels import (conv_args, dev, from_planar, lib, make_conv, nhwc, odyadic, orc, pack_act,  # noqa: F401
                                    rand_tables, stream, to_planar, unpack_q)

pytestmark = pytest.mark.gpu

# (pixels per workgroup, band pixels per LDS stage) of the kernels, in tile-id order
GEOM2 = [(128, 256), (256, 384)]


def _ids(lib):
    n, nb2, ng2 = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_band2_tiles(), lib.load().hawq_conv2d_num_gemm2_tiles()
    assert nb2 == len(GEOM2)
    return list(range(n - ng2 - nb2 + 1, n - ng2 + 1))


def _applies(bm, band_px, w, cin, bits=8):
    return bm + 2 * w + 9 <= band_px - 4 and cin * bits // 8 >= 128 and (bits == 8 or cin % 128 == 0)


def _args(lib, x, wt, b, tile, bits=8):
    from hawq_amd.packing import pack_conv_weight, pack_w3x3_band
    a, keep = conv_args(lib, x, wt, b, 1, 1, bits, bits, tile=tile)
    cout, cin = wt.shape[0], wt.shape[1]
    keep['wb'] = dev(pack_w3x3_band(pack_conv_weight(wt, bits), cout, cin * bits // 8))
    keep['xp'] = dev(to_planar(pack_act(x, bits)))
    a.wgt_band, a.in_, a.in_planar = keep['wb'].data_ptr(), keep['xp'].data_ptr(), 1
    return a, keep


# all ResNet 3x3 map sizes with Cin >= 128; several images per workgroup; ragged last pixel tile (M % 128 != 0); rectangular maps;
# Cin / Cout that differ; one pixel tile only; maps narrower than a wave
SHAPES = [(3, 28, 28, 128, 128), (5, 14, 14, 256, 256), (9, 7, 7, 512, 128), (2, 9, 30, 192, 128), (1, 14, 20, 128, 192),
          (7, 7, 7, 128, 64), (1, 3, 5, 128, 64), (2, 56, 56, 128, 64)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", [1, 5])
def test_band2_requant(lib, orc, shape, mode):
    """REQUANT epilogue: both ReLU settings, NHWC and planar output, tie-free and exact-tie instantiations."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(h * 1000 + w + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, 8, 8)
    acc = orc.conv2d(x, wt, b, 1, 1)
    m, e = rand_tables(rng, cout, 2e-5, 3e-4)
    assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
    ran = 0
    for tile, (bm, band_px) in zip(_ids(lib), GEOM2):
        a, keep = _args(lib, x, wt, b, tile)
        keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
        out = torch.zeros(acc.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), mode
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
        if not _applies(bm, band_px, w, cin):
            assert lib.load().hawq_conv2d(C.byref(a), None) != 0   # refused, not mis-computed
            continue
        for relu in (1, 0):
            ref = odyadic(orc, np.maximum(acc, 0) if relu else acc, m, e, (-128, 127))
            for outp in (0, 1):
                a.relu, a.out_planar = relu, outp
                out.zero_()
                lib.call("hawq_conv2d", C.byref(a), stream())
                got = from_planar(out, (n, h, w, cout), 8) if outp else unpack_q(out, (n, h, w, cout), 8)
                assert np.array_equal(got, ref), (tile, relu, outp)
        ran += 1
        # what these kernels do not take is refused: NHWC input, missing packed weights, 4-bit output
        a.in_, a.in_planar = keep['x'].data_ptr(), 0
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
        a.in_, a.in_planar, a.wgt_band = keep['xp'].data_ptr(), 1, None
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
        a.wgt_band, a.out_bits = keep['wb'].data_ptr(), 4
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
    assert ran >= 1


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (5, 14, 14, 256, 256), (9, 7, 7, 512, 256), (2, 9, 30, 192, 64)])
@pytest.mark.parametrize("mode", [1, 5])
def test_band2_residual(lib, orc, shape, mode):
    """RESIDUAL epilogue (second conv of a basic block): uint16 residual in and out, the next QuantAct's int8 output (NHWC and planar),
    either output alone, and the sticky overflow flag."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(7 * h + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, 8, 8)
    acc = orc.conv2d(x, wt, b, 1, 1)
    m2, e2 = rand_tables(rng, cout, 2e-5, 3e-4)
    assert tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1)
    res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
    m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + odyadic(orc, res, m1, e1), 0)
    assert ref_res.max() < 65536
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    ran = 0
    for tile, (bm, band_px) in zip(_ids(lib), GEOM2):
        if not _applies(bm, band_px, w, cin):
            continue
        a, keep = _args(lib, x, wt, b, tile)
        keep.update(ctab=dev(pack_ctab(b, m2, e2)), m=dev(m2), e=dev(e2), res=dev(nhwc(res).astype(np.uint16)))
        flags = torch.zeros(1, dtype=torch.int32, device='cuda')
        out_res = torch.zeros(ref_res.size, dtype=torch.uint16, device='cuda')
        out_q = torch.zeros(ref_res.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.m, a.e, a.ctab, a.flags = lib.EPI_RESIDUAL, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), flags.data_ptr()
        a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 16, int(m1[0]), int(e1[0])
        a.res_out, a.res_out_bits = out_res.data_ptr(), 16
        a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, 0, 127, int(mq[0]), int(eq[0])
        a.fast_tables = mode
        for outp in (0, 1):
            a.out_planar = outp
            out_res.zero_(), out_q.zero_()
            lib.call("hawq_conv2d", C.byref(a), stream())
            got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
            assert np.array_equal(got, ref_res), (tile, outp)
            gq = from_planar(out_q, (n, h, w, cout), 8) if outp else unpack_q(out_q, (n, h, w, cout), 8)
            assert np.array_equal(gq, ref_q), (tile, outp)
            assert flags.item() == 0
        # either output alone
        a.out_planar, a.out_q = 0, None
        out_res.zero_()
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2), ref_res)
        a.out_q, a.res_out = out_q.data_ptr(), None
        out_q.zero_()
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(unpack_q(out_q, (n, h, w, cout), 8), ref_q)
        # a residual that leaves uint16 raises the sticky flag (and only then)
        big = res.copy()
        big[0, 0, 0, 0] = 65535
        keep['big'] = dev(nhwc(big).astype(np.uint16))
        m1b, e1b = requant_table(torch.tensor([1.5 * 0.7]), torch.ones(1), torch.tensor([0.7]))
        a.res_in, a.res_out, a.m_id_scalar, a.e_id_scalar = keep['big'].data_ptr(), out_res.data_ptr(), int(m1b[0]), int(e1b[0])
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert flags.item() == 1
        ran += 1
    assert ran >= 1


@pytest.mark.parametrize("name,shape", [("3x3 128->128 @28^2", (128, 28, 28, 128, 128)), ("3x3 256->256 @14^2", (128, 14, 14, 256, 256)),
                                        ("3x3 512->512 @7^2", (128, 7, 7, 512, 512))])
def test_band2_full_size_equals_the_band_kernels(lib, name, shape):
    """Benchmark-sized layers (batch 128): every round-5 kernel reproduces, byte for byte, what the first band kernel of rounds 1-3
    that takes the layer writes (those are pinned to the oracle at full size by tests/test_gpu_fullsize.py)."""
    from hawq_amd.packing import pack_ctab
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, 8, 8)
    m, e = rand_tables(rng, cout, 2e-5, 3e-4)
    nt, nb, nb2 = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_band_tiles(), lib.load().hawq_conv2d_num_band2_tiles()
    a, keep = _args(lib, x, wt, b, 0)
    keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
    out = torch.zeros(n * h * w * cout, dtype=torch.uint8, device='cuda')
    a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), 1
    a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
    ref = None
    for tile in range(nt - nb + 1, nt - nb + 9):   # the band kernels of rounds 1-3 and the weight-stationary kernel
        a.tile = tile
        if lib.load().hawq_conv2d(C.byref(a), stream()) == 0:
            torch.cuda.synchronize()
            ref = out.clone()
            break
    assert ref is not None
    ran = 0
    for tile in _ids(lib):
        a.tile = tile
        out.zero_()
        if lib.load().hawq_conv2d(C.byref(a), stream()) != 0:
            continue
        torch.cuda.synchronize()
        assert torch.equal(out, ref), tile
        ran += 1
    assert ran >= 1


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (5, 14, 14, 256, 256), (9, 7, 7, 512, 128), (2, 9, 30, 256, 64), (2, 14, 14, 128, 64)])
@pytest.mark.parametrize("mode", [1, 5])
def test_band2_hawq4_operands_and_outputs(lib, orc, shape, mode):
    """W4A4 layers (both operands hawq4 nibbles, Cin % 128 == 0; Cin = 128 is a single 64-byte slice and is refused: the kernels want
    two): int8 and hawq4 outputs, NHWC and planar; and a hawq4 output from int8 operands (mixed schedules)."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(h * 100 + w + cin + 4)
    for bits in (4, 8):
        x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, bits, bits)
        acc = orc.conv2d(x, wt, b, 1, 1)
        m, e = rand_tables(rng, cout, 2e-5 if bits == 8 else 2e-3, 3e-4 if bits == 8 else 2e-2)
        assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
        for tile, (bm, band_px) in zip(_ids(lib), GEOM2):
            a, keep = _args(lib, x, wt, b, tile, bits)
            keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
            a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), mode
            for out_bits, (lo, hi) in ((8, (-128, 127)), (4, (0, 15))):
                if bits == 8 and out_bits == 8:
                    continue   # test_band2_requant
                out = torch.zeros(acc.size * out_bits // 8, dtype=torch.uint8, device='cuda')
                a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), out_bits, lo, hi
                if not _applies(bm, band_px, w, cin, bits):
                    a.out_planar = 0
                    assert lib.load().hawq_conv2d(C.byref(a), None) != 0
                    continue
                ref = odyadic(orc, np.maximum(acc, 0), m, e, (lo, hi))
                for outp in (0, 1):
                    a.out_planar = outp
                    out.zero_()
                    lib.call("hawq_conv2d", C.byref(a), stream())
                    got = from_planar(out, (n, h, w, cout), out_bits) if outp else unpack_q(out, (n, h, w, cout), out_bits)
                    assert np.array_equal(got, ref), (bits, tile, out_bits, outp)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bits", [8, 4])
def test_band2_raw_accumulators(lib, orc, shape, bits):
    """Round 6: north_star asks for bit-exactness "on the int32 pre-requant accumulators".  HAWQ_EPI_RAW instantiations of
    the round-5 3x3 kernels expose exactly those (accumulators + bias, dense [M][Cout] int32) and are compared with the oracle's exact
    sums (oracle.conv2d = quant_modules.py:489-494 restated in integers) - int8 and hawq4 operands, every geometry of this file."""
    n, h, w, cin, cout = shape
    if bits == 4 and (cin % 128 or cin < 256):
        pytest.skip("hawq4 operands need Cin % 128 == 0 and a 128-byte pixel row")
    rng = np.random.default_rng(31 * h + w + cin + bits)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, bits, bits)
    ref = orc.conv2d(x, wt, b, 1, 1)
    ran = 0
    for tile, (bm, band_px) in zip(_ids(lib), GEOM2):
        a, keep = _args(lib, x, wt, b, tile, bits)
        out = torch.full((ref.size,), -7, dtype=torch.int32, device='cuda')
        a.epilogue, a.out_acc = lib.EPI_RAW, out.data_ptr()
        if not _applies(bm, band_px, w, cin, bits):
            assert lib.load().hawq_conv2d(C.byref(a), None) != 0   # refused, not mis-computed
            continue
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out.cpu().numpy().reshape(n, h, w, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref), f"tile {tile}"
        ran += 1
    assert ran >= 1
