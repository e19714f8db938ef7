"""GPU parity tests of the individual HIP kernels (through the C ABI) against the CPU oracle.
Bit-exact: every comparison is np.array_equal on integers (or on fp32 produced by one exact
multiply)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def lib():
    from hawq_amd import _lib
    _lib.load()
    _lib.check(_lib.load().hawq_device_ok())
    return _lib


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x_nchw):
    return np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))


def pack_act(x_nchw, bits):
    from hawq_amd.packing import pack_hawq4
    v = nhwc(x_nchw)
    return v.astype(np.int8).view(np.uint8) if bits == 8 else pack_hawq4(v)


def unpack_q(t, shape_nhwc, bits):
    from hawq_amd.packing import unpack_hawq4
    raw = t.cpu().numpy()
    n, h, w, c = shape_nhwc
    if bits == 8:
        v = raw.view(np.int8).astype(np.int64).reshape(n, h, w, c)
    else:
        v = unpack_hawq4(raw.reshape(n, h, w, c // 2)).astype(np.int64)
    return v.transpose(0, 3, 1, 2)


def to_planar(packed):
    """NHWC activation bytes (int8, or hawq4-packed) -> channel-group planes [row_bytes/16][M][16 B]
    (include/hawq_mi355.h: in_planar / out_planar)."""
    raw = np.ascontiguousarray(packed).view(np.uint8)
    rb = raw.shape[-1]
    return np.ascontiguousarray(raw.reshape(-1, rb // 16, 16).transpose(1, 0, 2))


def from_planar(t, shape_nhwc, bits):
    """device planes -> NCHW int64 (inverse of to_planar + unpack_q)."""
    n, h, w, c = shape_nhwc
    rb = c * bits // 8
    raw = t.cpu().numpy().view(np.uint8).reshape(rb // 16, n * h * w, 16).transpose(1, 0, 2).reshape(n, h, w, rb)
    return unpack_q(torch.from_numpy(np.ascontiguousarray(raw)), shape_nhwc, bits)


def odyadic(orc, acc, m, ek, clamp=None):
    """oracle dyadic on a device-format (m, e|k<<8) table: (v*2^k*m)/2^e == v*m/2^(e-k)."""
    ek = np.asarray(ek, np.int64)
    return orc.dyadic(acc, np.asarray(m, np.int64), ((ek & 0xff) - (ek >> 8)).astype(np.int32), clamp)


def rand_tables(rng, cout, lo=2e-4, hi=3e-3):
    """random per-channel requant ratios -> device-contract (m, e).  Dividing by a non-trivial
    output scale gives m a full 31-bit mantissa, as in a real network."""
    from hawq_amd.quant_utils import requant_table
    r = torch.from_numpy((rng.uniform(lo, hi, cout) * 0.7).astype(f32))
    return requant_table(torch.ones(1), r, torch.tensor([0.7]))


def make_conv(rng, n, h, w, cin, cout, k, a_bits, w_bits):
    a_lo, a_hi = (-128, 127) if a_bits == 8 else (0, 15)
    w_lo, w_hi = (-127, 127) if w_bits == 8 else (-8, 7)
    x = rng.integers(a_lo, a_hi + 1, (n, cin, h, w)).astype(np.int64)
    wt = rng.integers(w_lo, w_hi + 1, (cout, cin, k, k)).astype(np.int64)
    b = rng.integers(-20000, 20000, cout).astype(np.int64)
    return x, wt, b


def conv_args(lib, x, wt, b, stride, pad, a_bits, w_bits, tile=0):
    from hawq_amd.packing import pack_conv_weight
    n, cin, h, w = x.shape
    cout, _, k, _ = wt.shape
    t = dict(x=dev(pack_act(x, a_bits)), w=dev(pack_conv_weight(wt, w_bits)), b=dev(b.astype(np.int32)))
    a = lib.ConvArgs()
    a.in_, a.wgt, a.bias = t['x'].data_ptr(), t['w'].data_ptr(), t['b'].data_ptr()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, h, w, cin, cout, k, k, stride, pad
    a.in_bits, a.w_bits, a.tile = a_bits, w_bits, tile
    return a, t


# (pixels per workgroup, channel tile, band pixels per LDS stage, needs a single 64-channel slice) of the 3x3 band
# tiles, in tile-id order (the last ids of hawq_conv2d_num_tiles())
BAND_GEOM = [(256, 64, 512, True), (256, 128, 512, False), (128, 128, 256, False), (256, 128, 384, False),
             (128, 128, 256, False), (128, 64, 256, False), (256, 64, 512, True), (256, 64, 512, True)]
PERSIST = len(BAND_GEOM) - 2   # the last two are the weight-stationary kernel (band_persist.hip; 1 / 2 workgroups per CU): Cin == Cout == 64, int8 in and out, REQUANT, NHWC output

SHAPES = [  # n, h, w, cin, cout, k, stride, pad
    (2, 14, 14, 64, 64, 1, 1, 0),
    (2, 14, 14, 64, 128, 3, 1, 1),
    (1, 14, 14, 128, 256, 1, 2, 0),
    (2, 15, 13, 64, 64, 3, 2, 1),
    (1, 7, 7, 256, 128, 3, 1, 1),      # M = 49: ragged pixel tile
    (3, 9, 9, 192, 64, 1, 1, 0),       # Cin not a power of two
    (1, 1, 1, 512, 192, 1, 1, 0),      # M = 1 (the FC shape)
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bits", [(8, 8), (4, 4), (8, 4), (4, 8)])
def test_conv_raw_accumulators(lib, orc, shape, bits):
    n, h, w, cin, cout, k, stride, pad = shape
    rng = np.random.default_rng(hash((shape, bits)) % 2 ** 32)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, k, *bits)
    ref = orc.conv2d(x, wt, b, stride, pad)
    for tile in range(0, lib.load().hawq_conv2d_num_tiles() + 1 - lib.load().hawq_conv2d_num_band_tiles()):  # the last ids are the 3x3 band kernels
        a, keep = conv_args(lib, x, wt, b, stride, pad, *bits, tile=tile)
        out = torch.full((ref.size,), -7, dtype=torch.int32, device='cuda')
        a.epilogue, a.out_acc = lib.EPI_RAW, out.data_ptr()
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out.cpu().numpy().reshape(n, ref.shape[2], ref.shape[3], cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref), f"tile {tile}"


def test_conv_identity_weights_asymmetric(lib):
    """A = I check with an asymmetric activation tensor: catches operand transposes/permutations."""
    n, h, w, c = 1, 8, 8, 64
    x = (np.arange(n * c * h * w).reshape(n, c, h, w) % 251 - 125).astype(np.int64)
    wt = np.zeros((c, c, 1, 1), np.int64)
    wt[np.arange(c), np.arange(c), 0, 0] = 1
    a, keep = conv_args(lib, x, wt, np.zeros(c, np.int64), 1, 0, 8, 8)
    out = torch.empty(n * h * w * c, dtype=torch.int32, device='cuda')
    a.epilogue, a.out_acc = lib.EPI_RAW, out.data_ptr()
    lib.call("hawq_conv2d", C.byref(a), stream())
    assert np.array_equal(out.cpu().numpy().reshape(n, h, w, c).transpose(0, 3, 1, 2), x)


@pytest.mark.parametrize("fast", [0, 1, 3, 5])
@pytest.mark.parametrize("out_bits", [8, 4])
@pytest.mark.parametrize("bits", [(8, 8), (4, 4)])
def test_conv_requant_epilogue(lib, orc, bits, out_bits, fast):
    """fast=1: host-proved tie-free tables -> LDS-staged 2-instruction requant kernels; fast=3: additionally no
    pre-shift in any table (one instruction less); fast=5: fast kernels in exact-tie mode, with forced exact .5
    ties; fast=0: exact general kernels, with forced exact .5 ties."""
    from hawq_amd.quant_utils import tables_are_fast
    rng = np.random.default_rng(5)
    n, h, w, cin, cout, k = 2, 12, 12, 128, 128, 3
    x, wt, b = make_conv(rng, n, h, w, cin, cout, k, *bits)
    acc = orc.conv2d(x, wt, b, 1, 1)
    for c in (0, 1):  # channels 0 / 1 get the tie-prone tables below: centre their accumulators on zero
        b[c] += 2 - int(np.median(acc[:, c]))
    acc = orc.conv2d(x, wt, b, 1, 1)
    m, e = rand_tables(rng, cout, 2e-5 if bits[0] == 8 else 2e-3, 3e-4 if bits[0] == 8 else 2e-2)
    if fast in (1, 3):
        assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
        assert fast != 3 or (e >> 8 == 0).all()
    else:
        m[0], e[0] = 1 << 30, 33 | (1 << 8)  # ratio 1/4 (e=32 lifted by k=1): produces exact .5 ties
        m[1], e[1] = 3 << 29, 34             # ratio 3/16: ties whenever acc = 8 mod 16
        pos = acc[:, 0][acc[:, 0] > 0].astype(np.int64)
        assert (pos % 4 == 2).any()  # the data does contain exact .5 ties (on both parities of the quotient)
        assert not tables_are_fast(m, e, 20)
    lo, hi = (-128, 127) if out_bits == 8 else (0, 15)
    ref = odyadic(orc, np.maximum(acc, 0), m, e, (lo, hi))
    a, keep = conv_args(lib, x, wt, b, 1, 1, *bits)
    a.fast_tables = fast
    if fast:
        from hawq_amd.packing import pack_ctab
        keep['ctab'] = dev(pack_ctab(b, m, e))
        a.ctab = keep['ctab'].data_ptr()
    md, ed = dev(m), dev(e)
    out = torch.zeros(ref.size * out_bits // 8, dtype=torch.uint8, device='cuda')
    a.epilogue, a.relu, a.m, a.e = lib.EPI_REQUANT, 1, md.data_ptr(), ed.data_ptr()
    a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), out_bits, lo, hi
    lib.call("hawq_conv2d", C.byref(a), stream())
    assert np.array_equal(unpack_q(out, (n, h, w, cout), out_bits), ref)
    if fast:  # planar output of the staged epilogue, every generic tile; planar input is refused by those kernels
        for tile in range(0, lib.load().hawq_conv2d_num_tiles() - lib.load().hawq_conv2d_num_band_tiles() + 1):
            a.tile, a.out_planar = tile, 1
            out.zero_()
            lib.call("hawq_conv2d", C.byref(a), stream())
            assert np.array_equal(from_planar(out, (n, h, w, cout), out_bits), ref), tile
        a.tile, a.out_planar, a.in_planar = 1, 0, 1
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0
        a.tile, a.in_planar = 0, 0
    else:
        a.out_planar = 1
        assert lib.load().hawq_conv2d(C.byref(a), None) != 0  # the general kernels write NHWC only
        a.out_planar = 0
    # without ReLU, symmetric clamp
    if out_bits == 8:
        a.relu = 0
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(unpack_q(out, (n, h, w, cout), 8), odyadic(orc, acc, m, e, (lo, hi)))


@pytest.mark.parametrize("shape", [(2, 56, 56, 64, 64), (3, 28, 28, 128, 128), (5, 14, 14, 256, 256), (9, 7, 7, 128, 128),
                                   (1, 14, 20, 64, 192), (2, 9, 30, 192, 128), (7, 7, 7, 512, 128), (1, 3, 5, 64, 64)])
@pytest.mark.parametrize("bits", [8, 4])
def test_conv3x3_band_kernels(lib, orc, shape, bits):
    """The LDS-band 3x3 kernels (the last tile ids) vs the oracle: all ResNet50 spatial sizes, several images per
    workgroup, ragged last tile, rectangular maps; both ReLU settings; int8 and hawq4 outputs; int8 and hawq4
    (W4A4, Cin % 128 == 0) operands."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(h * 1000 + w + cin)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, bits, bits)
    acc = orc.conv2d(x, wt, b, 1, 1)
    m, e = rand_tables(rng, cout, 2e-5 if bits == 8 else 2e-3, 3e-4 if bits == 8 else 2e-2)
    assert tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1)
    ntiles, nband = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_band_tiles()
    # (pixels per workgroup, channel tile, band pixels per LDS stage, needs Cin == 64) of the band tiles, in id order
    # (the round-5 kernels behind them have their own tests: tests/test_gpu_band2.py)
    geom = BAND_GEOM
    assert nband - lib.load().hawq_conv2d_num_band2_tiles() - lib.load().hawq_conv2d_num_gemm2_tiles() == len(geom)
    ran = 0
    for gi, (tile, (bm, bn, band_px, cin64)) in enumerate(zip(range(ntiles - nband + 1, ntiles + 1), geom)):
        chunks = cin // 64 if bits == 8 else cin // 128
        applies = (cout % bn == 0 and ((bm + w - 1) // w + 3) * (w + 2) <= band_px - 4 and (chunks == 1 or not cin64)
                   and (bits == 8 or cin % 128 == 0))
        if gi >= PERSIST:
            applies = applies and cout == 64 and bits == 8
        if not applies:
            a, keep = conv_args(lib, x, wt, b, 1, 1, bits, bits, tile=tile)
            keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
            out = torch.zeros(acc.size, dtype=torch.uint8, device='cuda')
            a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, 1, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), 1
            a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), 8, -128, 127
            assert lib.load().hawq_conv2d(C.byref(a), None) != 0   # refused, not mis-computed
            continue
        for relu, out_bits, (lo, hi) in ((1, 8, (-128, 127)), (0, 8, (-128, 127)), (1, 4, (0, 15))):
            a, keep = conv_args(lib, x, wt, b, 1, 1, bits, bits, tile=tile)
            keep.update(ctab=dev(pack_ctab(b, m, e)), m=dev(m), e=dev(e))
            out = torch.zeros(acc.size * out_bits // 8, dtype=torch.uint8, device='cuda')
            a.epilogue, a.relu, a.m, a.e, a.ctab, a.fast_tables = lib.EPI_REQUANT, relu, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), 1
            a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), out_bits, lo, hi
            if gi >= PERSIST and out_bits != 8:
                assert lib.load().hawq_conv2d(C.byref(a), None) != 0
                continue
            lib.call("hawq_conv2d", C.byref(a), stream())
            ref = odyadic(orc, np.maximum(acc, 0) if relu else acc, m, e, (lo, hi))
            assert np.array_equal(unpack_q(out, (n, h, w, cout), out_bits), ref), (tile, relu, out_bits)
            # the same launch reading channel-group planes and / or writing them
            keep['xp'] = dev(to_planar(pack_act(x, bits)))
            for inp, outp in ((1, 0), (0, 1), (1, 1)):
                a.in_, a.in_planar, a.out_planar = (keep['xp'] if inp else keep['x']).data_ptr(), inp, outp   # (round 5: the weight-stationary kernel writes planes too)
                out.zero_()
                lib.call("hawq_conv2d", C.byref(a), stream())
                got = from_planar(out, (n, h, w, cout), out_bits) if outp else unpack_q(out, (n, h, w, cout), out_bits)
                assert np.array_equal(got, ref), (tile, relu, out_bits, inp, outp)
            ran += 1
    assert ran >= 3 or (bits == 4 and cin % 128)
    # a layer the band kernels cannot take is refused, not mis-computed
    a, keep = conv_args(lib, x, wt, b, 2, 1, bits, bits, tile=ntiles)
    assert lib.load().hawq_conv2d(C.byref(a), None) != 0


@pytest.mark.parametrize("shape", [(2, 56, 56, 64, 64), (3, 28, 28, 128, 128), (5, 14, 14, 256, 256), (9, 7, 7, 128, 256)])
@pytest.mark.parametrize("bits,mode", [(8, 1), (8, 5), (4, 1)])
def test_conv3x3_band_residual(lib, orc, shape, bits, mode):
    """Band kernels with the RESIDUAL epilogue (3x3 second conv of a basic block): uint16 residual in and out,
    fused next-QuantAct output; tie-free and exact-tie instantiations; int8 and hawq4 operands."""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import requant_table, tables_are_fast
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(7 * h + cin + bits)
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 3, bits, bits)
    acc = orc.conv2d(x, wt, b, 1, 1)
    m2, e2 = rand_tables(rng, cout, 2e-5 if bits == 8 else 2e-3, 3e-4 if bits == 8 else 2e-2)
    assert tables_are_fast(m2, e2, int(np.abs(acc).max()).bit_length() + 1)
    res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
    m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + odyadic(orc, res, m1, e1), 0)
    assert ref_res.max() < 65536
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    ntiles, nband = lib.load().hawq_conv2d_num_tiles(), lib.load().hawq_conv2d_num_band_tiles()
    geom = BAND_GEOM
    ran = 0
    for gi, (tile, (bm, bn, band_px, one_chunk)) in enumerate(zip(range(ntiles - nband + 1, ntiles + 1), geom)):
        chunks = cin // 64 if bits == 8 else cin // 128
        if not (cout % bn == 0 and ((bm + w - 1) // w + 3) * (w + 2) <= band_px - 4 and (chunks == 1 or not one_chunk)
                and (bits == 8 or cin % 128 == 0)):
            continue
        if gi >= PERSIST and not (cin == 64 and cout == 64 and bits == 8):   # the weight-stationary kernel (round 5: RESIDUAL epilogue too)
            continue
        a, keep = conv_args(lib, x, wt, b, 1, 1, bits, bits, tile=tile)
        keep.update(ctab=dev(pack_ctab(b, m2, e2)), m=dev(m2), e=dev(e2), res=dev(nhwc(res).astype(np.uint16)))
        flags = torch.zeros(1, dtype=torch.int32, device='cuda')
        out_res = torch.zeros(ref_res.size, dtype=torch.uint16, device='cuda')
        out_q = torch.zeros(ref_res.size, dtype=torch.uint8, device='cuda')
        a.epilogue, a.m, a.e, a.ctab, a.flags = lib.EPI_RESIDUAL, keep['m'].data_ptr(), keep['e'].data_ptr(), keep['ctab'].data_ptr(), flags.data_ptr()
        a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 16, int(m1[0]), int(e1[0])
        a.res_out, a.res_out_bits = out_res.data_ptr(), 16
        a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, 0, 127, int(mq[0]), int(eq[0])
        a.fast_tables = mode
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref_res), tile
        assert np.array_equal(unpack_q(out_q, (n, h, w, cout), 8), ref_q), tile
        assert flags.item() == 0
        keep['xp'] = dev(to_planar(pack_act(x, bits)))
        a.in_, a.in_planar, a.out_planar = keep['xp'].data_ptr(), 1, 1
        out_res.zero_(), out_q.zero_()
        lib.call("hawq_conv2d", C.byref(a), stream())
        got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref_res) and np.array_equal(from_planar(out_q, (n, h, w, cout), 8), ref_q), tile
        ran += 1
    assert ran >= 1 or (bits == 4 and cin % 128)


@pytest.mark.parametrize("fast", [0, 1, 3, 5])
@pytest.mark.parametrize("res_bits", [16, 32])
@pytest.mark.parametrize("dual", [False, True])
def test_conv_residual_epilogue(lib, orc, dual, res_bits, fast):
    from hawq_amd.quant_utils import tables_are_fast
    rng = np.random.default_rng(11 + dual)
    n, h, w, cin, cout = 2, 14, 14, 64, 256
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    acc = orc.conv2d(x, wt, b, 1, 0)
    m2, e2 = rand_tables(rng, cout, 1e-3, 4e-2)
    a, keep = conv_args(lib, x, wt, b, 1, 0, 8, 8)
    if dual:  # identity 1x1 stride-2 conv on a 2x larger grid
        x2, w2, b2 = make_conv(rng, n, 2 * h, 2 * w, 128, cout, 1, 8, 8)
        acc_id = orc.conv2d(x2, w2, b2, 2, 0)
        m1, e1 = rand_tables(rng, cout, 1e-3, 4e-2)
        idq = odyadic(orc, acc_id, m1, e1)
        from hawq_amd.packing import pack_conv_weight
        keep.update(x2=dev(pack_act(x2, 8)), w2=dev(pack_conv_weight(w2, 8)), b2=dev(b2.astype(np.int32)),
                    m1=dev(m1), e1=dev(e1))
        a.in2, a.wgt2, a.bias2 = keep['x2'].data_ptr(), keep['w2'].data_ptr(), keep['b2'].data_ptr()
        a.H2, a.W2, a.Cin2, a.stride2, a.in2_bits, a.w2_bits = 2 * h, 2 * w, 128, 2, 8, 8
        a.m_id, a.e_id = keep['m1'].data_ptr(), keep['e1'].data_ptr()
    else:
        res = rng.integers(0, 60000, (n, cout, h, w)).astype(np.int64)
        from hawq_amd.quant_utils import requant_table
        m1, e1 = requant_table(torch.tensor([0.37 * 0.7]), torch.ones(1), torch.tensor([0.7]))
        idq = odyadic(orc, res, m1, e1)
        keep['res'] = dev(nhwc(res).astype(np.uint16 if res_bits == 16 else np.int32))
        a.res_in, a.res_in_bits = keep['res'].data_ptr(), res_bits
        a.m_id_scalar, a.e_id_scalar = int(m1[0]), int(e1[0])
    ref_res = np.maximum(odyadic(orc, acc, m2, e2) + idq, 0)
    assert ref_res.max() < 65536
    from hawq_amd.quant_utils import requant_table
    mq, eq = requant_table(torch.tensor([0.0039 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    ref_q = odyadic(orc, ref_res, mq, eq, (0, 127))
    md, ed = dev(m2), dev(e2)
    flags = torch.zeros(1, dtype=torch.int32, device='cuda')
    out_res = torch.zeros(ref_res.size, dtype=torch.uint16 if res_bits == 16 else torch.int32, device='cuda')
    out_q = torch.zeros(ref_res.size, dtype=torch.uint8, device='cuda')
    a.epilogue, a.m, a.e, a.flags = lib.EPI_RESIDUAL, md.data_ptr(), ed.data_ptr(), flags.data_ptr()
    a.res_out, a.res_out_bits = out_res.data_ptr(), res_bits
    a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, 0, 127, int(mq[0]), int(eq[0])
    if fast:
        vb = int(np.abs(acc).max()).bit_length() + 1
        assert tables_are_fast(m2, e2, vb) and tables_are_fast(m1, e1, 22 if dual else 17, allow_shift=not dual)
        assert tables_are_fast(mq, eq, 17)
        # fast=3: no pre-shift in the conv tables and (mq, eq); the scalar identity table may keep its own
        assert fast != 3 or ((e2 >> 8 == 0).all() and (not dual or (e1 >> 8 == 0).all()) and int(eq[0]) >> 8 == 0)
    a.fast_tables = fast  # (32-bit residuals run the general kernels either way)
    if fast:
        from hawq_amd.packing import pack_ctab
        keep['ctab'] = dev(pack_ctab(b, m2, e2))
        a.ctab = keep['ctab'].data_ptr()
        if dual:
            keep['ctab_id'] = dev(pack_ctab(b2, m1, e1))
            a.ctab_id = keep['ctab_id'].data_ptr()
    lib.call("hawq_conv2d", C.byref(a), stream())
    got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
    assert np.array_equal(got, ref_res)
    assert np.array_equal(unpack_q(out_q, (n, h, w, cout), 8), ref_q)
    assert flags.item() == 0


def test_residual_uint16_overflow_sets_flag(lib, orc):
    rng = np.random.default_rng(3)
    n, h, w, cin, cout = 1, 8, 8, 64, 64
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    a, keep = conv_args(lib, x, wt, b, 1, 0, 8, 8)
    res = np.full((n, h, w, cout), 65000, np.uint16)
    keep['res'] = dev(res)
    m2 = np.full(cout, 1 << 30, np.int32)
    e2 = np.full(cout, 33 | (2 << 8), np.int32)  # ratio 1/2 (e=31 lifted by k=2)
    md, ed = dev(m2), dev(e2)
    flags = torch.zeros(1, dtype=torch.int32, device='cuda')
    out_res = torch.zeros(n * h * w * cout, dtype=torch.uint16, device='cuda')
    a.epilogue, a.m, a.e, a.flags = lib.EPI_RESIDUAL, md.data_ptr(), ed.data_ptr(), flags.data_ptr()
    a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 16, 1 << 30, 33 | (3 << 8)  # ratio 1
    a.res_out, a.res_out_bits = out_res.data_ptr(), 16
    lib.call("hawq_conv2d", C.byref(a), stream())
    acc = orc.conv2d(x, wt, b, 1, 0)
    expect_ovf = (np.maximum(odyadic(orc, acc, m2, e2) + 65000, 0) > 65535).any()
    assert bool(flags.item() & 1) == bool(expect_ovf) and expect_ovf


def test_bad_arguments_return_errors(lib):
    a = lib.ConvArgs()
    rc = lib.load().hawq_conv2d(C.byref(a), None)
    assert rc != 0 and b"hawq_conv2d" in lib.load().hawq_last_error()
    with pytest.raises(RuntimeError):
        lib.call("hawq_quantize_input", None, None, 1, 3, 8, 8, 8, 8, 0, 0, 1.0, -128, 127, None)


def test_quantize_input_and_stem(lib, orc):
    from hawq_amd.packing import pack_stem_weight
    rng = np.random.default_rng(9)
    n, hh, ww = 2, 38, 46
    x = rng.normal(0, 1.2, (n, 3, hh, ww)).astype(f32)
    x.reshape(-1)[:6] = [0.5, 1.5, 2.5, -0.5, 1e9, -1e9]
    scale = f32(1.0)
    q_ref = orc.quantize_f32(x, scale, 8)
    ho, wo = (hh + 6 - 7) // 2 + 1, (ww + 6 - 7) // 2 + 1
    hp, wp = max(2 * (ho - 1) + 8, hh + 3), max(2 * (wo - 1) + 8, ww + 4)
    wp += wp & 1
    xq = torch.zeros(n * hp * wp * 4, dtype=torch.int8, device='cuda')
    xd = dev(x)
    lib.call("hawq_quantize_input", xd.data_ptr(), xq.data_ptr(), n, 3, hh, ww, hp, wp, 3, 3, float(f32(1) / scale),
             -128, 127, stream())
    got = xq.cpu().numpy().reshape(n, hp, wp, 4)[:, 3:3 + hh, 3:3 + ww, :3].transpose(0, 3, 1, 2)
    assert np.array_equal(got, q_ref)
    wt = rng.integers(-127, 128, (64, 3, 7, 7)).astype(np.int64)
    b = rng.integers(-30000, 30000, 64).astype(np.int64)
    acc = orc.conv2d(q_ref, wt, b, 2, 3)
    m, e = rand_tables(rng, 64, 2e-3, 4e-2)
    ref16 = np.maximum(odyadic(orc, acc, m, e, (-32768, 32767)), 0)
    wd, bd, md, ed = dev(pack_stem_weight(wt)), dev(b.astype(np.int32)), dev(m), dev(e)
    out16 = torch.zeros(n * ho * wo * 64, dtype=torch.uint16, device='cuda')
    out_acc = torch.zeros(n * ho * wo * 64, dtype=torch.int32, device='cuda')
    lib.call("hawq_stem_conv7", xq.data_ptr(), wd.data_ptr(), bd.data_ptr(), md.data_ptr(), ed.data_ptr(), n, hp, wp,
             ho, wo, -32768, 32767, out16.data_ptr(), out_acc.data_ptr(), stream())
    assert np.array_equal(out_acc.cpu().numpy().reshape(n, ho, wo, 64).transpose(0, 3, 1, 2), acc)
    assert np.array_equal(out16.cpu().numpy().astype(np.int64).reshape(n, ho, wo, 64).transpose(0, 3, 1, 2), ref16)
    # max-pool + first QuantAct
    from hawq_amd.quant_utils import requant_table
    mq, eq = requant_table(torch.tensor([0.0041]), torch.ones(1), torch.ones(1))
    pooled = orc.maxpool(ref16, 3, 2, 1)
    h1, w1 = pooled.shape[2:]
    for bits, (lo, hi) in ((8, (-128, 127)), (4, (0, 15))):
        res = torch.zeros(pooled.size, dtype=torch.uint16, device='cuda')
        qo = torch.zeros(pooled.size * bits // 8, dtype=torch.uint8, device='cuda')
        lib.call("hawq_maxpool3s2_requant", out16.data_ptr(), n, ho, wo, 64, res.data_ptr(), qo.data_ptr(), bits,
                 int(mq[0]), int(eq[0]), lo, hi, stream())
        assert np.array_equal(res.cpu().numpy().astype(np.int64).reshape(n, h1, w1, 64).transpose(0, 3, 1, 2), pooled)
        assert np.array_equal(unpack_q(qo, (n, h1, w1, 64), bits), odyadic(orc, pooled, mq, eq, (lo, hi)))
        q2 = torch.zeros_like(qo)
        lib.call("hawq_requant_residual", res.data_ptr(), 16, pooled.size, q2.data_ptr(), bits, int(mq[0]), int(eq[0]),
                 lo, hi, stream())
        assert torch.equal(q2, qo)


@pytest.mark.parametrize("shape", [(3, 46, 38), (2, 64, 64), (1, 35, 51)])
def test_fused_stem_matches_unfused_semantics(lib, orc, shape):
    """hawq_stem_fused == quantize -> conv7x7/2 -> +bias -> max-pool(3,2,1) -> QuantAct16 -> ReLU -> QuantAct
    (q_resnet.py:115-122), incl. odd sizes, ragged 8x8 blocks and an odd batch."""
    from hawq_amd.packing import pack_stem_weight
    from hawq_amd.quant_utils import requant_table
    n, hh, ww = shape
    rng = np.random.default_rng(hh * 100 + ww)
    x = rng.normal(0, 1.3, (n, 3, hh, ww)).astype(f32)
    scale = f32(0.0213)
    q_ref = orc.quantize_f32(x, scale, 8)
    wt = rng.integers(-127, 128, (64, 3, 7, 7)).astype(np.int64)
    b = rng.integers(-30000, 30000, 64).astype(np.int64)
    acc = orc.maxpool(orc.conv2d(q_ref, wt, b, 2, 3), 3, 2, 1)
    m, e = rand_tables(rng, 64, 2e-3, 4e-2)
    r16 = np.maximum(odyadic(orc, acc, m, e, (-32768, 32767)), 0)
    mq, eq = requant_table(torch.tensor([0.0041 * 0.7]), torch.ones(1), torch.tensor([0.7]))
    hp, wp = acc.shape[2:]
    xd, wd, bd, md, ed = dev(x), dev(pack_stem_weight(wt)), dev(b.astype(np.int32)), dev(m), dev(e)
    from hawq_amd.quant_utils import tables_are_fast
    can_fast = tables_are_fast(m, e, int(np.abs(acc).max()).bit_length() + 1) and tables_are_fast(mq, eq, 17)
    for bits, (lo, hi), fast in ((8, (-128, 127), 0), (4, (0, 15), 0), (8, (-128, 127), int(can_fast))):
        res = torch.zeros(acc.size, dtype=torch.uint16, device='cuda')
        qo = torch.zeros(acc.size * bits // 8, dtype=torch.uint8, device='cuda')
        lib.call("hawq_stem_fused", xd.data_ptr(), n, 3, hh, ww, float(f32(1) / scale), -128, 127, wd.data_ptr(),
                 bd.data_ptr(), md.data_ptr(), ed.data_ptr(), -32768, 32767, res.data_ptr(), qo.data_ptr(), bits,
                 int(mq[0]), int(eq[0]), lo, hi, fast, stream())
        assert np.array_equal(res.cpu().numpy().astype(np.int64).reshape(n, hp, wp, 64).transpose(0, 3, 1, 2), r16)
        assert np.array_equal(unpack_q(qo, (n, hp, wp, 64), bits), odyadic(orc, r16, mq, eq, (lo, hi)))


@pytest.mark.parametrize("res_bits", [16, 32])
@pytest.mark.parametrize("c", [512, 320])
def test_avgpool_requant(lib, orc, res_bits, c):
    """c = 512 takes the 16-byte-load uint16 kernel, c = 320 / int32 residuals the general one."""
    from hawq_amd.quant_utils import requant_table
    rng = np.random.default_rng(21)
    n = 3
    x = rng.integers(0, 45000, (n, c, 7, 7)).astype(np.int64)
    x[1, 7] = 65535
    x[0, 0] = 3
    x[0, 1] = 5
    x[0, 1, 6, 6] = 4
    pooled = orc.avgpool_trunc(x)
    mq, eq = requant_table(torch.tensor([0.0038]), torch.ones(1), torch.ones(1))
    ref = odyadic(orc, pooled, mq, eq, (-128, 127))
    xin = dev(nhwc(x).astype(np.uint16 if res_bits == 16 else np.int32))
    out = torch.zeros(n * c, dtype=torch.int8, device='cuda')
    pd = torch.zeros(n * c, dtype=torch.int32, device='cuda')
    lib.call("hawq_avgpool_requant", xin.data_ptr(), res_bits, n, 49, c, out.data_ptr(), pd.data_ptr(), int(mq[0]),
             int(eq[0]), -128, 127, stream())
    assert np.array_equal(pd.cpu().numpy().reshape(n, c), pooled)
    assert np.array_equal(out.cpu().numpy().astype(np.int64).reshape(n, c), ref)


def test_dequant_epilogue_fc(lib, orc):
    rng = np.random.default_rng(2)
    bsz, k, nout = 5, 512, 1000
    x = rng.integers(-128, 128, (bsz, k)).astype(np.int64)
    wt = rng.integers(-127, 128, (nout, k)).astype(np.int64)
    b = rng.integers(-50000, 50000, nout).astype(np.int64)
    acc = orc.linear(x, wt, b)
    fs = rng.uniform(1e-5, 1e-4, nout).astype(f32)
    ref = (acc.astype(f32) * fs.reshape(1, -1)).astype(f32)
    from hawq_amd.packing import pack_conv_weight
    wp = dev(pack_conv_weight(wt.reshape(nout, k, 1, 1), 8, k, 1024))
    bp = np.zeros(1024, np.int32)
    bp[:nout] = b
    fsp = np.zeros(1024, f32)
    fsp[:nout] = fs
    xd, bd, fd = dev(x.astype(np.int8)), dev(bp), dev(fsp)
    out = torch.full((bsz, nout), float('nan'), device='cuda')
    a = lib.ConvArgs()
    a.in_, a.wgt, a.bias = xd.data_ptr(), wp.data_ptr(), bd.data_ptr()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = bsz, 1, 1, k, 1024, 1, 1, 1, 0
    a.in_bits, a.w_bits, a.epilogue = 8, 8, lib.EPI_DEQUANT
    a.out_f32, a.fscale, a.ldo, a.n_valid = out.data_ptr(), fd.data_ptr(), nout, nout
    lib.call("hawq_conv2d", C.byref(a), stream())
    assert np.array_equal(out.cpu().numpy(), ref)


# ---------------------------------------------------------------- fp32-convention adapters vs live-reference KATs
@pytest.mark.parametrize("n,k,nout", [(1, 512, 10), (37, 2048, 1000), (64, 2048, 1000), (128, 512, 1000), (64, 1280, 1000), (33, 128, 33)])
def test_fc_dequant_kernel_equals_the_conv_kernels_dequant_epilogue(lib, orc, n, k, nout):
    """QuantLinear's frozen path (quant_modules.py:112-130) through its own kernel (fc_dequant.hip, round 5): the int32 sums are the
    CPU oracle's, the fp32 logits are bit for bit numpy's float32 product and those of hawq_conv2d's DEQUANT epilogue."""
    from hawq_amd.packing import pack_conv_weight
    rng = np.random.default_rng(n * 7 + k)
    x = rng.integers(-128, 128, (n, k)).astype(np.int64)
    wt = rng.integers(-127, 128, (nout, k)).astype(np.int64)
    b = rng.integers(-3000, 3000, nout).astype(np.int64)
    nout_p = (nout + 63) // 64 * 64
    fsp = np.zeros(nout_p, f32)
    fsp[:nout] = rng.uniform(1e-5, 1e-3, nout).astype(f32)
    bp = np.zeros(nout_p, np.int32)
    bp[:nout] = b
    wp = dev(pack_conv_weight(wt.reshape(nout, k, 1, 1), 8, k, nout_p))
    xd, bd, fd = dev(x.astype(np.int8)), dev(bp), dev(fsp)
    assert lib.load().hawq_fc_dequant_ok(n, k, nout_p)
    out = torch.full((n, nout), -7.0, dtype=torch.float32, device="cuda")
    lib.call("hawq_fc_dequant", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), fd.data_ptr(), out.data_ptr(), n, k, nout_p, nout, nout, stream())
    torch.cuda.synchronize()
    ref = (orc.linear(x, wt, b).astype(f32) * fsp[:nout].reshape(1, -1)).astype(f32)
    assert np.array_equal(out.cpu().numpy(), ref)
    a = lib.ConvArgs()
    a.in_, a.wgt, a.bias = xd.data_ptr(), wp.data_ptr(), bd.data_ptr()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = n, 1, 1, k, nout_p, 1, 1, 1, 0
    a.in_bits, a.w_bits, a.epilogue = 8, 8, lib.EPI_DEQUANT
    via_conv = torch.full((n, nout), -9.0, dtype=torch.float32, device="cuda")
    a.out_f32, a.fscale, a.ldo, a.n_valid = via_conv.data_ptr(), fd.data_ptr(), nout, nout
    lib.call("hawq_conv2d", C.byref(a), stream())
    torch.cuda.synchronize()
    assert torch.equal(out, via_conv)


def test_fixedpoint_fn_matches_reference_kats():
    from hawq_amd.quant_utils import fixedpoint_fn
    kf = H.load("kat_functions.npz")
    for tag in ["rand8", "rand4", "rand16", "tie8", "tie16"]:
        bits, sym = kf[f"fp0_{tag}_bits"]
        mode = "symmetric" if sym else "asymmetric"
        t = lambda k: torch.from_numpy(kf[k]).cuda()
        y = fixedpoint_fn.apply(t(f"fp0_{tag}_z"), int(bits), mode, t(f"fp0_{tag}_sout"), 0, t(f"fp0_{tag}_sa"),
                                t(f"fp0_{tag}_sw"))
        assert np.array_equal(y.cpu().numpy(), kf[f"fp0_{tag}_y"]), tag
        for itag in ("pass", "conv"):
            k = f"fp1_{tag}_{itag}"
            ident = t(k + "_ident")
            y1 = fixedpoint_fn.apply(t(f"fp0_{tag}_z") + ident, int(bits), mode, t(f"fp0_{tag}_sout"), 1,
                                     t(f"fp0_{tag}_sa"), t(f"fp0_{tag}_sw"), ident, t(k + "_sida"), t(k + "_sidw"))
            assert np.array_equal(y1.cpu().numpy(), kf[k + "_y"]), k


def test_quant_functions_match_reference_kats():
    from hawq_amd.quant_utils import AsymmetricQuantFunction, SymmetricQuantFunction
    kf = H.load("kat_functions.npz")
    x = torch.from_numpy(kf["q_x"]).cuda()
    for tag in "abc":
        s = torch.from_numpy(kf[f"symq_{tag}_scale"]).cuda()
        assert np.array_equal(SymmetricQuantFunction.apply(x, 8, s).cpu().numpy(), kf[f"symq_{tag}_8"])
        assert np.array_equal(SymmetricQuantFunction.apply(x, 4, s).cpu().numpy(), kf[f"symq_{tag}_4"])
        assert np.array_equal(AsymmetricQuantFunction.apply(x, 4, s).cpu().numpy(), kf[f"asymq_{tag}_4"])


def test_modules_match_reference_kats():
    """QuantBnConv2d / QuantLinear / QuantAveragePool2d / QuantAct module forwards (fp32 tuple
    convention) against tensors recorded from the live reference modules."""
    from hawq_amd import quant_modules as qm
    km = H.load("kat_modules.npz")
    for tag in ["c3", "c1s2", "c3s2", "c7"]:
        g = lambda k: torch.from_numpy(km[f"conv_{tag}_{k}"])
        cin, cout, k, stride, pad, bits, hw = km[f"conv_{tag}_cfg"]
        conv = torch.nn.Conv2d(int(cin), int(cout), int(k), int(stride), int(pad), bias=False)
        bn = torch.nn.BatchNorm2d(int(cout))
        with torch.no_grad():
            conv.weight.copy_(g("w")); bn.weight.copy_(g("gamma")); bn.bias.copy_(g("beta"))
            bn.running_mean.copy_(g("mean")); bn.running_var.copy_(g("var"))
        m = qm.QuantBnConv2d(weight_bit=int(bits), bias_bit=32, per_channel=True, fix_BN=True)
        m.set_param(conv, bn)
        m.fix()
        m = m.cuda().eval()
        s_a = g("s_a").cuda()
        y, s_w = m(((g("q") * g("s_a")).cuda(), s_a))
        if np.array_equal(s_w.cpu().numpy(), km[f"conv_{tag}_s_w"]):  # else: sqrt quirk
            assert np.array_equal(m.weight_integer.cpu().numpy(), km[f"conv_{tag}_weight_integer"])
            assert np.array_equal(m.bias_integer.cpu().numpy(), km[f"conv_{tag}_bias_integer"])
            # reference fp32 output carries un-rounded x/S_a; the recovered integers must agree
            rec = lambda t: torch.round(t / s_a.view(1, -1, 1, 1).cpu() / g("s_w").view(1, -1, 1, 1))
            assert torch.equal(rec(y.cpu()), rec(g("y"))), tag
    lin = torch.nn.Linear(64, 10)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(km["lin_w"])); lin.bias.copy_(torch.from_numpy(km["lin_b"]))
    m = qm.QuantLinear(weight_bit=8, bias_bit=32, per_channel=True)
    m.set_param(lin)
    m = m.cuda()
    s_a = torch.from_numpy(km["lin_s_a"]).cuda()
    y = m(torch.from_numpy(km["lin_q"]).cuda() * s_a, s_a)
    assert np.array_equal(y.cpu().numpy(), km["lin_y"])
    assert np.array_equal(m.weight_integer.cpu().numpy(), km["lin_weight_integer"])
    p = qm.QuantAveragePool2d(7, 1)
    s = torch.from_numpy(km["pool_s"]).cuda()
    y, _ = p(torch.from_numpy(km["pool_q"]).cuda() * s, s)
    assert np.array_equal(y.cpu().numpy(), km["pool_y"])
    a = qm.QuantAct(activation_bit=8)
    a.x_min += float(km["act_in_rng"][0])
    a.x_max += float(km["act_in_rng"][1])
    a.fix()
    a = a.cuda()
    y, s = a(torch.from_numpy(km["act_in_x"]).cuda())
    assert np.array_equal(s.cpu().numpy(), km["act_in_s"]) and np.array_equal(y.cpu().numpy(), km["act_in_y"])


def test_cpu_tensors_raise():
    from hawq_amd import quant_modules as qm
    a = qm.QuantAct(activation_bit=8)
    with pytest.raises(RuntimeError):
        a(torch.randn(1, 3, 4, 4))


def _rec(y, s_a, s_w):
    """integers carried by a module's fp32 output: rint(y / S_a / S_w[c])"""
    return torch.round(y / s_a.view(1, -1, 1, 1) / s_w.view(1, -1, 1, 1))


def test_quantconv2d_grouped_depthwise_and_percentile_match_reference_kats():
    """QuantConv2d (quant_modules.py:605-736) - plain, grouped, depthwise (MobileNetV2's layer type), without bias,
    4-bit weights, percentile weight ranges (per channel and per tensor) - and QuantBnConv2d with a depthwise conv:
    module forwards against tensors recorded from the LIVE reference (tests/golden/make_kat_extra.py)."""
    from hawq_amd import quant_modules as qm
    kx = H.load("kat_extra.npz")
    tags = sorted(k[len("qconv_"):-len("_cfg")] for k in kx.files if k.startswith("qconv_") and k.endswith("_cfg"))
    assert {"c3", "c1nb", "g4", "dw", "dws2", "dw4", "pct", "pct_t"} <= set(tags)
    for tag in tags:
        g = lambda k: torch.from_numpy(kx[f"qconv_{tag}_{k}"])
        cin, cout, k, stride, pad, groups, bias, wbit, pc, hw = (int(v) for v in kx[f"qconv_{tag}_cfg"])
        pct = float(kx[f"qconv_{tag}_pct"][0])
        conv = torch.nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=bool(bias))
        with torch.no_grad():
            conv.weight.copy_(g("w"))
            if bias:
                conv.bias.copy_(g("b"))
        m = qm.QuantConv2d(weight_bit=wbit, bias_bit=32 if bias else None, per_channel=bool(pc), weight_percentile=pct)
        m.set_param(conv)
        m = m.cuda()
        s_a = g("s_a").cuda()
        y, s_w = m((g("q") * g("s_a")).cuda(), s_a)
        assert np.array_equal(s_w.cpu().numpy().reshape(-1), kx[f"qconv_{tag}_s_w"].reshape(-1)), tag
        assert np.array_equal(m.weight_integer.cpu().numpy(), kx[f"qconv_{tag}_weight_integer"]), tag
        if bias:
            assert np.array_equal(m.bias_integer.cpu().numpy().reshape(-1), kx[f"qconv_{tag}_bias_integer"].reshape(-1)), tag
        sw = s_w.cpu().reshape(-1).expand(cout) if s_w.numel() == 1 else s_w.cpu()
        # the reference's fp32 conv runs on the un-rounded x / S_a (quant_modules.py:727-736): compare the integers it carries
        assert torch.equal(_rec(y.cpu(), g("s_a"), sw), _rec(g("y"), g("s_a"), sw)), tag
    for tag in ("bndw", "bndws2", "bnpct"):
        g = lambda k: torch.from_numpy(kx[f"{tag}_{k}"])
        c, stride, pct = int(kx[f"{tag}_cfg"][0]), int(kx[f"{tag}_cfg"][1]), float(kx[f"{tag}_cfg"][2])
        conv = torch.nn.Conv2d(c, c, 3, stride, 1, groups=c if tag != "bnpct" else 1, bias=False)
        bn = torch.nn.BatchNorm2d(c)
        with torch.no_grad():
            conv.weight.copy_(g("w")); bn.weight.copy_(g("gamma")); bn.bias.copy_(g("beta"))
            bn.running_mean.copy_(g("mean")); bn.running_var.copy_(g("var"))
        m = qm.QuantBnConv2d(weight_bit=8, bias_bit=32, per_channel=True, fix_BN=True, weight_percentile=pct)
        m.set_param(conv, bn)
        m.fix()
        m = m.cuda().eval()
        s_a = g("s_a").cuda()
        y, s_w = m(((g("q") * g("s_a")).cuda(), s_a))
        if np.array_equal(s_w.cpu().numpy(), kx[f"{tag}_s_w"]):  # else: sqrt quirk
            assert np.array_equal(m.weight_integer.cpu().numpy(), kx[f"{tag}_weight_integer"]), tag
            assert np.array_equal(m.bias_integer.cpu().numpy(), kx[f"{tag}_bias_integer"]), tag
            assert torch.equal(_rec(y.cpu(), g("s_a"), g("s_w")), _rec(g("y"), g("s_a"), g("s_w"))), tag


@pytest.mark.parametrize("shape", [(2, 9, 7, 32, 1), (1, 14, 14, 96, 2), (3, 5, 6, 8, 2), (2, 56, 56, 144, 1), (1, 113, 57, 24, 2), (2, 3, 3, 4, 1)])
def test_depthwise3x3_matches_the_grouped_kernel_and_numpy(lib, shape):
    """hawq_depthwise3x3 (4 channels x 4 pixels per thread, tap-major weights; MobileNetV2's conv2) against the plain
    one-thread-per-output hawq_conv2d_grouped and, for the small cases, a numpy convolution: exact int32 accumulators, odd
    widths / heights, both strides, with and without bias."""
    n, h, w, c, stride = shape
    rng = np.random.default_rng(c * 131 + h)
    x = rng.integers(-128, 128, (n, h, w, c)).astype(np.int8)
    wt = rng.integers(-127, 128, (c, 3, 3)).astype(np.int8)
    b = rng.integers(-30000, 30000, c).astype(np.int32)
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    xd, wg, w9, bd = dev(x), dev(np.ascontiguousarray(wt.reshape(c, 3, 3, 1))), dev(np.ascontiguousarray(wt.reshape(c, 9).T)), dev(b)
    for bias in (bd, None):
        out_g = torch.zeros(n * ho * wo * c, dtype=torch.int32, device='cuda')
        out_d = torch.zeros_like(out_g)
        bp = bias.data_ptr() if bias is not None else None
        lib.call("hawq_conv2d_grouped", xd.data_ptr(), wg.data_ptr(), bp, n, h, w, c, c, 3, 3, stride, 1, c, out_g.data_ptr(), stream())
        lib.call("hawq_depthwise3x3", xd.data_ptr(), w9.data_ptr(), bp, n, h, w, c, stride, out_d.data_ptr(), stream())
        assert torch.equal(out_d, out_g), shape
    if n * h * w * c < 50000:
        xp = np.pad(x.astype(np.int64), ((0, 0), (1, 1), (1, 1), (0, 0)))
        ref = np.zeros((n, ho, wo, c), np.int64)
        for kh in range(3):
            for kw in range(3):
                ref += xp[:, kh:kh + (ho - 1) * stride + 1:stride, kw:kw + (wo - 1) * stride + 1:stride, :] * wt[:, kh, kw].astype(np.int64)
        assert np.array_equal(out_d.cpu().numpy().reshape(n, ho, wo, c), ref)
    assert lib.load().hawq_depthwise3x3(xd.data_ptr(), w9.data_ptr(), None, n, h, w, c + 1, stride, out_d.data_ptr(), None) != 0


@pytest.mark.parametrize("shape", [(2, 9, 7, 32, 1, 1), (1, 14, 14, 96, 2, 1), (3, 5, 6, 8, 2, 0), (1, 113, 57, 24, 2, 1), (2, 3, 3, 4, 1, 0)])
def test_depthwise3x3_requant_matches_accumulators_plus_host_dyadic(lib, orc, shape):
    """hawq_depthwise3x3_requant (MobileNetV2's conv2 fused with the QuantAct behind it: per-channel dyadic requant of the
    optionally rectified accumulator, clamp to the activation range) against hawq_depthwise3x3's accumulators pushed through
    the oracle's dyadic RNE; the optional accumulator output must equal the plain kernel's."""
    from hawq_amd.quant_utils import requant_table
    n, h, w, c, stride, relu = shape
    rng = np.random.default_rng(c * 17 + h)
    x = rng.integers(-128, 128, (n, h, w, c)).astype(np.int8)
    wt = rng.integers(-127, 128, (c, 3, 3)).astype(np.int8)
    b = rng.integers(-30000, 30000, c).astype(np.int32)
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    ratio = torch.from_numpy(rng.uniform(2e-4, 3e-3, c).astype(f32))
    m, e = requant_table(ratio, torch.ones(c), torch.ones(1), lift=False)
    if c >= 8:
        m[-4:] = 0   # a padding channel group (multiplier 0): the kernel skips its loads and MACs, the result is still rne(acc * 0) = 0
    xd, w9, bd, md, ed = dev(x), dev(np.ascontiguousarray(wt.reshape(c, 9).T)), dev(b), dev(m), dev(e)
    acc = torch.zeros(n * ho * wo * c, dtype=torch.int32, device='cuda')
    lib.call("hawq_depthwise3x3", xd.data_ptr(), w9.data_ptr(), bd.data_ptr(), n, h, w, c, stride, acc.data_ptr(), stream())
    lo, hi = (0, 127) if relu else (-128, 127)
    a = acc.cpu().numpy().astype(np.int64).reshape(n, ho, wo, c)
    ref = odyadic(orc, (np.maximum(a, 0) if relu else a).transpose(0, 3, 1, 2), m, e, (lo, hi)).transpose(0, 2, 3, 1)
    for keep_acc in (True, False):
        q = torch.full((n * ho * wo * c,), 77, dtype=torch.int8, device='cuda')
        acc2 = torch.zeros_like(acc)
        lib.call("hawq_depthwise3x3_requant", xd.data_ptr(), w9.data_ptr(), bd.data_ptr(), md.data_ptr(), ed.data_ptr(), n, h, w, c, 0, stride,
                 relu, lo, hi, q.data_ptr(), acc2.data_ptr() if keep_acc else None, stream())
        assert np.array_equal(q.cpu().numpy().reshape(n, ho, wo, c).astype(np.int64), ref), (shape, keep_acc)
        assert not keep_acc or torch.equal(acc2, acc)
    assert ref.min() < 0 or relu
    if c >= 8:   # padding channels declared (C_valid): same tensor with zeros there - their weights, bias and multipliers are zero by contract
        wz, bz = wt.copy(), b.copy()
        wz[c - 4:], bz[c - 4:] = 0, 0
        q2 = torch.full_like(q, 55)
        acc3 = torch.full_like(acc, 55)
        wzd, bzd = dev(np.ascontiguousarray(wz.reshape(c, 9).T)), dev(bz)
        lib.call("hawq_depthwise3x3_requant", xd.data_ptr(), wzd.data_ptr(), bzd.data_ptr(), md.data_ptr(),
                 ed.data_ptr(), n, h, w, c, c - 4, stride, relu, lo, hi, q2.data_ptr(), acc3.data_ptr(), stream())
        assert np.array_equal(q2.cpu().numpy().reshape(n, ho, wo, c).astype(np.int64), ref)
        az = a.copy()
        az[..., c - 4:] = 0
        assert np.array_equal(acc3.cpu().numpy().reshape(n, ho, wo, c), az)
    assert lib.load().hawq_depthwise3x3_requant(xd.data_ptr(), w9.data_ptr(), None, md.data_ptr(), ed.data_ptr(), n, h, w, c, 0, stride, 0, -129, 127,
                                                q.data_ptr(), None, None) != 0
    assert lib.load().hawq_depthwise3x3_requant(xd.data_ptr(), w9.data_ptr(), None, md.data_ptr(), ed.data_ptr(), n, h, w, c, c + 4, stride, 0, -128, 127,
                                                q.data_ptr(), None, None) != 0


@pytest.mark.parametrize("mode", ["linear", "linear_add", "relu_clamp_noid", "linear_add_pad"])
def test_residual_epilogue_without_relu_and_with_the_16_bit_clamp(lib, orc, mode):
    """The general RESIDUAL epilogue's MobileNetV2 switches (hawq_conv_args.res_no_relu / res_clamp16 / res_in == NULL):
    a linear bottleneck ends without an activation, so the 32-bit carrier holds signed values - requant(acc) [+ identity],
    clamped to the signed 16-bit range only where the reference's QuantAct runs fixedpoint_fn case 0
    (quant_utils.py:390-413), then requantised for the next unit with a signed clamp."""
    from hawq_amd.quant_utils import requant_table
    rng = np.random.default_rng(len(mode))
    n, h, w, cin, cout = 2, 9, 7, 64, 128
    x, wt, b = make_conv(rng, n, h, w, cin, cout, 1, 8, 8)
    pad_from = 72 if mode == "linear_add_pad" else cout   # n_valid: channels from here on are padding and skip the arithmetic
    wt[pad_from:], b[pad_from:] = 0, 0
    a, keep = conv_args(lib, x, wt, b, 1, 0, 8, 8)
    a.n_valid = pad_from if mode == "linear_add_pad" else 0
    mode = "linear_add" if mode == "linear_add_pad" else mode
    acc = orc.conv2d(x, wt, b, 1, 0)
    ratio = torch.from_numpy(rng.uniform(0.5, 3.0, cout).astype(f32))   # wide enough to leave the 16-bit range
    m2, e2 = requant_table(ratio, torch.ones(cout), torch.ones(1), lift=False)
    m2[pad_from:] = 0
    v = odyadic(orc, acc, m2, e2)
    if mode == "linear_add":
        res = rng.integers(-30000, 30000, (n, cout, h, w)).astype(np.int64)
        res[:, pad_from:] = 0
        m1, e1 = requant_table(torch.tensor([0.81]), torch.ones(1), torch.ones(1), lift=False)
        keep['res'] = dev(nhwc(res).astype(np.int32))
        a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 32, int(m1[0]), int(e1[0])
        v = v + odyadic(orc, res, m1, e1)
        a.res_no_relu, a.res_clamp16 = 1, 0
    elif mode == "linear":
        a.res_no_relu, a.res_clamp16 = 1, 1
        v = np.clip(v, -32768, 32767)
    else:
        a.res_no_relu, a.res_clamp16 = 0, 1
        v = np.clip(np.maximum(v, 0), -32768, 32767)
    assert np.abs(v).max() >= 32767 or mode == "linear_add"
    mq, eq = requant_table(torch.tensor([0.0031]), torch.ones(1), torch.ones(1), lift=False)
    lo, hi = (0, 127) if mode == "relu_clamp_noid" else (-128, 127)
    ref_q = odyadic(orc, v, mq, eq, (lo, hi))
    md, ed = dev(m2), dev(e2)
    flags = torch.zeros(1, dtype=torch.int32, device='cuda')
    out_res = torch.full((v.size,), 7, dtype=torch.int32, device='cuda')
    out_q = torch.full((v.size,), 7, dtype=torch.int8, device='cuda')
    a.epilogue, a.m, a.e, a.flags = lib.EPI_RESIDUAL, md.data_ptr(), ed.data_ptr(), flags.data_ptr()
    a.res_out, a.res_out_bits = out_res.data_ptr(), 32
    a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out_q.data_ptr(), 8, lo, hi, int(mq[0]), int(eq[0])
    lib.call("hawq_conv2d", C.byref(a), stream())
    got = out_res.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2)
    assert np.array_equal(got, v)
    assert np.array_equal(out_q.cpu().numpy().astype(np.int64).reshape(n, h, w, cout).transpose(0, 3, 1, 2), ref_q)
    # the fast (uint16) kernels have no signed carrier: the switches are refused there
    a.fast_tables, a.res_out_bits = 1, 16
    assert lib.load().hawq_conv2d(C.byref(a), None) != 0


@pytest.mark.parametrize("hw", [(8, 8), (9, 7), (33, 46), (224, 224)])
def test_quantize_im2col_is_the_input_quantiser_plus_patch_gather(lib, orc, hw):
    """hawq_quantize_im2col3x3s2 (MobileNetV2's input QuantAct fused with the im2col of its 3x3 / stride 2 / pad 1 init conv):
    every 64-byte row = the oracle-quantised 27 patch values in (kh, kw, c) order + zeros; a 1x1 conv on those rows with the
    weights in the same order equals the oracle's 3x3 stride-2 convolution of the quantised image."""
    h, w = hw
    n = 2
    rng = np.random.default_rng(h * 31 + w)
    x = rng.normal(0, 1.1, (n, 3, h, w)).astype(f32)
    x.reshape(-1)[:6] = [0.5, 1.5, 2.5, -0.5, 1e9, -1e9]
    scale = f32(0.0173)
    q = orc.quantize_f32(x, scale, 8)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    out = torch.full((n * ho * wo * 64,), 55, dtype=torch.int8, device='cuda')
    xd = dev(x)
    lib.call("hawq_quantize_im2col3x3s2", xd.data_ptr(), out.data_ptr(), n, 3, h, w, float(f32(1) / scale), -128, 127, stream())
    got = out.cpu().numpy().reshape(n, ho, wo, 64).astype(np.int64)
    qp = np.pad(q, ((0, 0), (0, 0), (1, 1), (1, 1)))
    ref = np.zeros((n, ho, wo, 64), np.int64)
    for kh in range(3):
        for kw in range(3):
            for c in range(3):
                ref[..., (kh * 3 + kw) * 3 + c] = qp[:, c, kh:kh + 2 * (ho - 1) + 1:2, kw:kw + 2 * (wo - 1) + 1:2]
    assert np.array_equal(got, ref)
    if h <= 64:
        wt = rng.integers(-127, 128, (64, 3, 3, 3)).astype(np.int64)
        b = rng.integers(-1000, 1000, 64).astype(np.int64)
        acc = orc.conv2d(q, wt, b, 2, 1)
        w27 = np.zeros((64, 64, 1, 1), np.int64)
        w27[:, :27, 0, 0] = wt.transpose(0, 2, 3, 1).reshape(64, 27)
        a, keep = conv_args(lib, ref.transpose(0, 3, 1, 2), w27, b, 1, 0, 8, 8)
        a.in_ = out.data_ptr()
        o = torch.zeros(n * ho * wo * 64, dtype=torch.int32, device='cuda')
        a.epilogue, a.out_acc = lib.EPI_RAW, o.data_ptr()
        lib.call("hawq_conv2d", C.byref(a), stream())
        assert np.array_equal(o.cpu().numpy().reshape(n, ho, wo, 64).transpose(0, 3, 1, 2), acc)
    assert lib.load().hawq_quantize_im2col3x3s2(xd.data_ptr(), out.data_ptr(), n, 4, h, w, 1.0, -128, 127, None) != 0


def test_range_statistics_kernels_match_reference_kats(lib):
    """hawq_minmax_f32 / hawq_kthvalue_f32 behind get_percentile_min_max and the un-frozen QuantAct (min/max and
    percentile ranges, initialisation + momentum / running-extremum updates) against the live reference's numbers."""
    from hawq_amd import quant_modules as qm
    from hawq_amd.quant_utils import device_min_max, get_percentile_min_max
    kx = H.load("kat_extra.npz")
    for i in range(5):
        x = torch.from_numpy(kx[f"pct{i}_x"])
        lowp, upp = (float(v) for v in kx[f"pct{i}_cfg"])
        lo, hi = get_percentile_min_max(x.cuda(), lowp, upp, output_tensor=True)
        assert np.array_equal(np.array([float(lo), float(hi)], np.float32), kx[f"pct{i}_out"]), i
        lo_h, hi_h = get_percentile_min_max(x, lowp, upp, output_tensor=True)   # host path (weight preparation)
        assert float(lo_h) == float(lo) and float(hi_h) == float(hi)
    g = torch.Generator().manual_seed(3)
    for n in (1, 5, 4096, 1000003):
        x = torch.randn(n, generator=g)
        lo, hi = device_min_max(x.cuda())
        assert float(lo) == float(x.min()) and float(hi) == float(x.max())
        for k in {1, (n + 1) // 2, n}:
            out = torch.zeros(1, device='cuda')
            scratch = torch.zeros(264, dtype=torch.int32, device='cuda')
            xd = x.cuda()
            lib.call("hawq_kthvalue_f32", xd.data_ptr(), n, k, 0, out.data_ptr(), scratch.data_ptr(), stream())
            assert float(out) == float(torch.kthvalue(x, k).values), (n, k)
            lib.call("hawq_kthvalue_f32", xd.data_ptr(), n, k, 1, out.data_ptr(), scratch.data_ptr(), stream())
            assert float(out) == float(-torch.kthvalue(-x, k).values), (n, k)
    for tag in ("mm", "mmx", "ps", "pa"):
        bits, pct, mom = kx[f"act_{tag}_cfg"]
        a = qm.QuantAct(activation_bit=int(bits), act_range_momentum=float(mom) if mom != -1 else -1,
                        quant_mode="asymmetric" if tag == "pa" else "symmetric", act_percentile=float(pct)).cuda()
        for it in range(3):
            y, s = a(torch.from_numpy(kx[f"act_{tag}_x"][it]).cuda())
            got = np.array([float(a.x_min), float(a.x_max), float(s)], np.float32)
            assert np.array_equal(got, kx[f"act_{tag}_rng"][it]), (tag, it, got, kx[f"act_{tag}_rng"][it])
            assert np.array_equal(y.cpu().numpy(), kx[f"act_{tag}_y"][it]), (tag, it)


def test_device_resize_center_crop_reproduces_real_pillow_output(tmp_path):
    """hawq_amd.image.resize_center_crop (hawq_resample_u8) against REAL Pillow: the crops tests/golden/make_pillow.py recorded from
    `Image.resize(.., BILINEAR)` + CenterCrop(224) for nine geometries and a decoded JPEG, byte for byte.  With Pillow installed: the
    committed JPEG through decode_image -> device pipeline, and an ImageFolder tree through folder_loader (classes / order / batches)."""
    from hawq_amd.image import decode_image, folder_loader, resize_center_crop
    from tests.test_host_logic import _synth_image
    fx = H.load("pillow_resize.npz")
    for i, (h, w) in enumerate(fx["geoms"]):
        img = _synth_image(int(h), int(w), int(fx["seeds"][i]))
        got = resize_center_crop(torch.from_numpy(img).cuda()).cpu().numpy()
        assert np.array_equal(got, fx[f"crop_{i}"]), (h, w, int(np.abs(got.astype(int) - fx[f"crop_{i}"].astype(int)).max()))
    got = resize_center_crop(torch.from_numpy(fx["jpeg_decoded"]).cuda()).cpu().numpy()
    assert np.array_equal(got, fx["jpeg_crop"])
    pytest.importorskip("PIL")
    from PIL import Image
    jpg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_500x375.jpg")
    dec = decode_image(jpg)
    ref = np.asarray(Image.open(jpg).convert("RGB").resize((341, 256), Image.BILINEAR))[16:240, 58:282]
    assert np.array_equal(resize_center_crop(dec.cuda()).cpu().numpy(), ref)
    rng = np.random.default_rng(2)
    want = {}
    for c in ("n01", "n02"):
        (tmp_path / c).mkdir()
        for k in range(3):
            h, w = (int(v) for v in rng.integers(230, 420, 2))
            Image.fromarray(_synth_image(h, w, k)).save(tmp_path / c / f"im{k}.jpg", quality=92)
    batches = list(folder_loader(str(tmp_path), batch_size=4))
    assert [tuple(b[0].shape) for b in batches] == [(4, 224, 224, 3), (2, 224, 224, 3)]
    assert torch.cat([b[1] for b in batches]).tolist() == [0, 0, 0, 1, 1, 1]
    first = np.asarray(Image.open(tmp_path / "n01" / "im0.jpg").convert("RGB"))
    from oracle import pil_resample
    assert np.array_equal(batches[0][0][0].cpu().numpy(), pil_resample.resize_center_crop(first))


def test_resize_center_crop_matches_the_restated_pillow_algorithm():
    """hawq_amd.image.resize_center_crop (hawq_resample_u8, crop window only) vs the whole-image numpy restatement of Pillow's
    antialiased bilinear resize + CenterCrop (oracle/pil_resample.py): landscape, portrait, up-scaling, identity axis."""
    from hawq_amd.image import preprocess_batch, resize_center_crop
    from oracle import pil_resample
    rng = np.random.default_rng(0)
    imgs = []
    for h, w in ((375, 500), (500, 333), (256, 256), (200, 180), (256, 400), (1200, 900), (224, 224)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        img[::7, ::5] = 255
        img[3::11, 2::13] = 0
        ref = pil_resample.resize_center_crop(img)
        got = resize_center_crop(torch.from_numpy(img).cuda()).cpu().numpy()
        assert got.shape == (224, 224, 3) and np.array_equal(got, ref), (h, w, int(np.abs(got.astype(int) - ref.astype(int)).max()))
        imgs.append(torch.from_numpy(img))
    batch = preprocess_batch(imgs[:3])
    assert batch.shape == (3, 224, 224, 3) and batch.dtype == torch.uint8 and batch.is_cuda
