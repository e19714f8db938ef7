"""Round 5: the streaming 1x1 kernels of gemm_v2.hip (the last hawq_conv2d_num_gemm2_tiles() tile ids) against the CPU oracle,
through the C ABI.  Bit-exact on the requantised int8 outputs and the un-clamped uint16 residuals (quant_modules.py:489-494,
quant_utils.py:390-456; q_resnet.py:231-260: reduce conv, expand conv + stored residual, expand conv + identity conv)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_gpu_kernels import (conv_args, dev, from_planar, lib, make_conv, nhwc, odyadic, orc, pack_act,  # noqa: F401
                                    rand_tables, stream, unpack_q)

pytestmark = pytest.mark.gpu

BN2 = [64, 64, 128]   # channel tile of the kernels, in tile-id order


def _ids(lib):
    n, ng2 = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_gemm2_tiles()
    assert ng2 == len(BN2) and lib.load().hawq_conv2d_gemm2_first() == n - ng2 + 1
    return list(range(n - ng2 + 1, n + 1))


def _args(lib, x, wt, b, stride, tile):
    from hawq_amd.packing import pack_conv_weight, pack_w1x1_k128
    a, keep = conv_args(lib, x, wt, b, stride, 0, 8, 8, tile=tile)
    cout, cin = wt.shape[0], wt.shape[1]
    keep['wk'] = dev(pack_w1x1_k128(pack_conv_weight(wt, 8), cout, cin))
    a.wgt_k128 = keep['wk'].data_ptr()
    return a, keep


# n, h, w, cin, cout, stride: K = 1 .. 16 chunks (shorter and longer than the ring), ragged last pixel tile, one tile only, strided rows
SHAPES = [(2, 14, 14, 128, 64, 1), (3, 14, 14, 1024, 128, 1), (5, 7, 7, 2048, 128, 1), (1, 28, 28, 256, 128, 2), (2, 9, 5, 384, 192, 1),
          (1, 3, 3, 512, 64, 1), (2, 56, 56, 256, 128, 2), (1, 15, 13, 640, 64, 2)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", [1, 5])
def test_gemm2_requant(lib, orc, shape, mode):
    """REQUANT epilogue (the reduce conv of a bottleneck, stride 1 and 2): both ReLU settings, NHWC and planar output."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast
    n, h, w, cin, cout, stride = shape
    rng = np.random.default_rng(h * 1000 + w + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    acc = orc.conv2d(x, wt, b, stride, 0)
    ho, wo = acc.shape[2], acc.shape[3]
    m, e = rand_tables(rng, cout, 2e-5, 3e-4)
    assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
    ran = 0
    for tile, bn in zip(_ids(lib), BN2):
        a, keep = _args(lib, x, wt, b, stride, tile)
        keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
        out = torch.zeros(acc.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), mode
        a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
        if cout % bn:
            assert lib.load().hawq_conv2d(C.byref(a), None) != 0   # refused, not mis-computed
            continue
        for relu in (1, 0):
            ref = odyadic(orc, np.maximum(acc, 0) if relu else acc, m, e, (-128, 127))
            for outp in (0, 1):
                a.relu, a.out_planar = relu, outp
                out.zero_()
                lib.call("hawq_conv2d", C.byref(a), stream())
                got = from_planar(out, (n, ho, wo, cout), 8) if outp else unpack_q(out, (n, ho, wo, cout), 8)
                assert np.array_equal(got, ref), (tile, relu, outp)
        ran += 1
        a.wgt_k128 = None   # missing packed weights: refused
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
    assert ran >= 1


@pytest.mark.parametrize("shape", [(2, 14, 14, 256, 128), (3, 7, 7, 512, 256), (1, 28, 28, 128, 128), (2, 9, 5, 384, 64)])
@pytest.mark.parametrize("mode", [1, 5])
def test_gemm2_residual(lib, orc, shape, mode):
    """RESIDUAL epilogue on a stored uint16 residual (the expand conv of a bottleneck): residual and next-QuantAct outputs, either alone,
    the sticky overflow flag."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(11 * h + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    acc = orc.conv2d(x, wt, b, 1, 0)
    m2, e2 = rand_tables(rng, cout, 2e-5, 3e-4)
    assert tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1)
    res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
    m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + odyadic(orc, res, m1, e1), 0)
    assert ref_res.max() < 65536
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    ran = 0
    for tile, bn in zip(_ids(lib), BN2):
        if cout % bn:
            continue
        a, keep = _args(lib, x, wt, b, 1, tile)
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
        a.out_planar, a.out_q = 0, None
        out_res.zero_()
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2), ref_res)
        a.out_q, a.res_out = out_q.data_ptr(), None
        out_q.zero_()
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(unpack_q(out_q, (n, h, w, cout), 8), ref_q)
        big = res.copy()
        big[0, 0, 0, 0] = 65535
        keep['big'] = dev(nhwc(big).astype(np.uint16))
        m1b, e1b = requant_table(torch.tensor([1.5 * 0.7]), torch.ones(1), torch.tensor([0.7]))
        a.res_in, a.res_out, a.m_id_scalar, a.e_id_scalar = keep['big'].data_ptr(), out_res.data_ptr(), int(m1b[0]), int(e1b[0])
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert flags.item() == 1
        ran += 1
    assert ran >= 1


# n, h2, w2 (block input map), cin (expand conv's K), cin2 (identity conv's K), cout, stride of the identity conv
DUAL_SHAPES = [(2, 28, 28, 128, 256, 128, 2), (3, 14, 14, 256, 512, 256, 2), (1, 14, 14, 512, 1024, 128, 2), (2, 9, 9, 128, 128, 64, 1)]


@pytest.mark.parametrize("shape", DUAL_SHAPES)
@pytest.mark.parametrize("mode", [1, 5])
def test_gemm2_residual_with_identity_conv(lib, orc, shape, mode):
    """The closing launch of a resize unit (q_resnet.py:236-258): expand conv + strided identity conv as a second phase of the same ring,
    each branch requantised by its own per-channel table, summed un-clamped, ReLU, next QuantAct."""
    from hawq_amd.packing import pack_conv_weight, pack_ctab, pack_w1x1_k128
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    n, h2, w2, cin, cin2, cout, s2 = shape
    rng = np.random.default_rng(13 * h2 + cin2)
    ho, wo = (h2 - 1) // s2 + 1, (w2 - 1) // s2 + 1
    x, wt, b = make_conv(rng, n, ho, wo, cin, cout, 1, 8, 8)
    xi, wti, bi = make_conv(rng, n, h2, w2, cin2, cout, 1, 8, 8)
    acc, acci = orc.conv2d(x, wt, b, 1, 0), orc.conv2d(xi, wti, bi, s2, 0)
    assert acc.shape == acci.shape
    m2, e2 = rand_tables(rng, cout, 2e-4, 3e-3)
    mi, ei = rand_tables(rng, cout, 2e-4, 3e-3)
    assert tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1) and tables_are_fast(mi, ei, int(np.abs(acci).max()).bit_length() + 1)
    mq, eq = requant_table(torch.tensor([0.039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + odyadic(orc, acci, mi, ei), 0)
    assert 0 < ref_res.max() < 65536
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    ran = 0
    for tile, bn in zip(_ids(lib), BN2):
        if cout % bn:
            continue
        a, keep = _args(lib, x, wt, b, 1, tile)
        keep.update(ctab=dev(pack_ctab(b, m2, e2)), ctab_id=dev(pack_ctab(bi, mi, ei)), m=dev(m2), e=dev(e2), mi=dev(mi), ei=dev(ei),
                    x2=dev(pack_act(xi, 8)), w2=dev(pack_conv_weight(wti, 8)), w2k=dev(pack_w1x1_k128(pack_conv_weight(wti, 8), cout, cin2)),
                    b2=dev(bi.astype(np.int32)))
        flags = torch.zeros(1, dtype=torch.int32, device='cuda')
        out_res = torch.zeros(ref_res.size, dtype=torch.uint16, device='cuda')
        out_q = torch.zeros(ref_res.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.m, a.e, a.ctab, a.flags = lib.EPI_RESIDUAL, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), flags.data_ptr()
        a.in2, a.wgt2, a.bias2, a.wgt2_k128 = keep['x2'].data_ptr(), keep['w2'].data_ptr(), keep['b2'].data_ptr(), keep['w2k'].data_ptr()
        a.H2, a.W2, a.Cin2, a.stride2, a.in2_bits, a.w2_bits = h2, w2, cin2, s2, 8, 8
        a.m_id, a.e_id, a.ctab_id = keep['mi'].data_ptr(), keep['ei'].data_ptr(), keep['ctab_id'].data_ptr()
        a.res_out, a.res_out_bits = out_res.data_ptr(), 16
        a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, 0, 127, int(mq[0]), int(eq[0])
        a.fast_tables = mode
        for outp in (0, 1):
            a.out_planar = outp
            out_res.zero_(), out_q.zero_()
            lib.call("hawq_conv2d", C.byref(a), stream())
            got = out_res.cpu().numpy().astype(np.int64).reshape(n, ho, wo, cout).transpose(0, 3, 1, 2)
            assert np.array_equal(got, ref_res), (tile, outp)
            gq = from_planar(out_q, (n, ho, wo, cout), 8) if outp else unpack_q(out_q, (n, ho, wo, cout), 8)
            assert np.array_equal(gq, ref_q), (tile, outp)
            assert flags.item() == 0
        a.wgt2_k128 = None   # second branch without its packed weights: refused
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
        ran += 1
    assert ran >= 1


# ---------------------------------------------------------------------------------------------------------------- round 6
def _args_bits(lib, x, wt, b, stride, tile, bits):
    """hawq4 operands (bits 4): activations and weights nibble-packed, the weight stream in 128-BYTE K chunks (= 256 channels)."""
    from hawq_amd.packing import pack_conv_weight, pack_w1x1_k128
    a, keep = conv_args(lib, x, wt, b, stride, 0, bits, bits, tile=tile)
    cout, cin = wt.shape[0], wt.shape[1]
    keep['wk'] = dev(pack_w1x1_k128(pack_conv_weight(wt, bits), cout, cin * bits // 8))
    a.wgt_k128 = keep['wk'].data_ptr()
    return a, keep


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bits", [8, 4])
def test_gemm2_raw_accumulators(lib, orc, shape, bits):
    """Round 6: the int32 accumulators (+ bias) of the streaming 1x1 kernels themselves against the oracle's exact sums
    (HAWQ_EPI_RAW instantiations; quant_modules.py:489-494), int8 and hawq4 operands."""
    n, h, w, cin, cout, stride = shape
    if bits == 4 and cin % 256:
        pytest.skip("hawq4 operands need Cin % 256 == 0 (128-byte K chunks)")
    rng = np.random.default_rng(17 * h + w + cin + bits)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, bits, bits)
    ref = orc.conv2d(x, wt, b, stride, 0)
    ho, wo = ref.shape[2], ref.shape[3]
    ran = 0
    for tile, bn in zip(_ids(lib), BN2):
        a, keep = _args_bits(lib, x, wt, b, stride, tile, bits)
        out = torch.full((ref.size,), -7, dtype=torch.int32, device='cuda')
        a.epilogue, a.out_acc = lib.EPI_RAW, out.data_ptr()
        if cout % bn:
            assert lib.load().hawq_conv2d(C.byref(a), None) != 0
            continue
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out.cpu().numpy().reshape(n, ho, wo, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref), f"tile {tile}"
        ran += 1
    assert ran >= 1


@pytest.mark.parametrize("shape", [(3, 14, 14, 1024, 128, 1), (5, 7, 7, 2048, 128, 1), (1, 28, 28, 256, 128, 2), (2, 56, 56, 256, 64, 1), (1, 3, 3, 512, 64, 1)])
@pytest.mark.parametrize("mode", [1, 5])
def test_gemm2_hawq4_operands_and_outputs(lib, orc, shape, mode):
    """Round 6: the reduce convs of the 4-bit schedules (bit_config.py:806, 1512) on the streaming kernel - hawq4
    activations and weights in, int8 or hawq4 out, NHWC rows or channel-group planes; REQUANT with both ReLU settings."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast
    n, h, w, cin, cout, stride = shape
    rng = np.random.default_rng(h * 77 + w + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 4, 4)
    acc = orc.conv2d(x, wt, b, stride, 0)
    ho, wo = acc.shape[2], acc.shape[3]
    m, e = rand_tables(rng, cout, 2e-4, 3e-3)
    if mode == 5:
        m[0], e[0] = 1 << 30, 33 | (1 << 8)   # ratio 1/4: exact .5 ties
        assert not tables_are_fast(m, e, 20)
    else:
        assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
    for tile, bn in zip(_ids(lib), BN2):
        if cout % bn:
            continue
        a, keep = _args_bits(lib, x, wt, b, stride, tile, 4)
        keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
        a.epilogue, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), mode
        for ob, (lo, hi) in ((8, (-128, 127)), (4, (0, 15))):
            out = torch.zeros(acc.size * ob // 8, dtype=torch.uint8, device='cuda')
            a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), ob, lo, hi
            for relu in ((1, 0) if ob == 8 else (1,)):
                ref = odyadic(orc, np.maximum(acc, 0) if relu else acc, m, e, (lo, hi))
                for outp in (0, 1):
                    a.relu, a.out_planar = relu, outp
                    out.zero_()
                    lib.call("hawq_conv2d", C.byref(a), stream())
                    got = from_planar(out, (n, ho, wo, cout), ob) if outp else unpack_q(out, (n, ho, wo, cout), ob)
                    assert np.array_equal(got, ref), (tile, ob, relu, outp)
        # mixed widths are refused, not mis-computed
        a.in_bits = 8
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0


def test_gemm2_hawq4_residual_and_identity(lib, orc):
    """hawq4 operands in the RESIDUAL epilogues: stored uint16 residual, and the identity conv as a second K phase (both phases nibbles)."""
    from hawq_amd.packing import pack_conv_weight, pack_ctab, pack_w1x1_k128
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 3, 14, 14, 512, 128
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 4, 4)
    acc = orc.conv2d(x, wt, b, 1, 0)
    m2, e2 = rand_tables(rng, cout, 2e-4, 3e-3)
    assert tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1)
    res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
    m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    # identity conv: block input [n, 2h, 2w, 256] nibbles at stride 2
    cin2 = 256
    x2, wt2, b2 = make_conv(rng, n, 2 * h, 2 * w, cin2, cout, 1, 4, 4)
    acc_id = orc.conv2d(x2, wt2, b2, 2, 0)
    mi, ei = rand_tables(rng, cout, 2e-4, 3e-3)
    assert tables_are_fast(mi, ei, int(np.abs(acc_id).max()).bit_length() + 1)
    for tile, bn in zip(_ids(lib), BN2):
        for dual in (False, True):
            a, keep = _args_bits(lib, x, wt, b, 1, tile, 4)
            keep.update(ctab=dev(pack_ctab(b, m2, e2)), m=dev(m2), e=dev(e2), res=dev(nhwc(res).astype(np.uint16)))
            flags = torch.zeros(1, dtype=torch.int32, device='cuda')
            idreq = odyadic(orc, acc_id, mi, ei) if dual else odyadic(orc, res, m1, e1)
            ref_res = np.maximum(odyadic(orc, acc, m2, e2) + idreq, 0)
            assert ref_res.max() < 65536
            out_res = torch.zeros(ref_res.size, dtype=torch.uint16, device='cuda')
            a.epilogue, a.m, a.e, a.ctab, a.flags, a.fast_tables = lib.EPI_RESIDUAL, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), flags.data_ptr(), 1
            a.res_out, a.res_out_bits = out_res.data_ptr(), 16
            if dual:
                keep.update(x2=dev(pack_act(x2, 4)), w2=dev(pack_conv_weight(wt2, 4)), wk2=dev(pack_w1x1_k128(pack_conv_weight(wt2, 4), cout, cin2 // 2)),
                            b2=dev(b2.astype(np.int32)), ctab_id=dev(pack_ctab(b2, mi, ei)), mi=dev(mi), ei=dev(ei))
                a.in2, a.wgt2, a.bias2, a.wgt2_k128 = keep['x2'].data_ptr(), keep['w2'].data_ptr(), keep['b2'].data_ptr(), keep['wk2'].data_ptr()
                a.H2, a.W2, a.Cin2, a.stride2, a.in2_bits, a.w2_bits = 2 * h, 2 * w, cin2, 2, 4, 4
                a.m_id, a.e_id, a.ctab_id = keep['mi'].data_ptr(), keep['ei'].data_ptr(), keep['ctab_id'].data_ptr()
            else:
                a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 16, int(m1[0]), int(e1[0])
            for ob, hi in ((8, 127), (4, 15)):
                ref_q = odyadic(orc, ref_res, mq, eq, (0, hi))
                out_q = torch.zeros(ref_res.size * ob // 8, dtype=torch.uint8, device='cuda')
                a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), ob, 0, hi, int(mq[0]), int(eq[0])
                out_res.zero_()
                lib.call("hawq_conv2d", C.byref(a), stream())
                assert np.array_equal(out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2), ref_res), (tile, dual, ob)
                assert np.array_equal(unpack_q(out_q, (n, h, w, cout), ob), ref_q), (tile, dual, ob)
                assert int(flags.item()) == 0
