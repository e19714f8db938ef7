"""MobileNetV2's round-4 launches through the C ABI against the CPU oracle (oracle/oracle.py, oracle/oracle_mbv2.py):
  * hawq_conv_args.in_pitch / out_pitch (ABI 4): narrow tensors stored at their own width,
  * the direct RESIDUAL epilogue with the fast contract's arithmetic (signed 32-bit carriers),
  * hawq_linear_bottleneck: one launch per unit (both organisations, slice groups, stride 1 / 2, identity or not, ragged tiles),
  * hawq_stem3x3s2: the init block as one launch (fp32 and uint8 images).
Every expectation is computed with the oracle's exact integer conv / depthwise conv / round-half-even dyadic."""
import ctypes as C

import numpy as np
import pytest
import torch

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


def odyadic(orc, acc, m, ek, clamp=None):
    ek = np.asarray(ek, np.int64)
    return orc.dyadic(acc, np.asarray(m, np.int64), ((ek & 0xff) - (ek >> 8)).astype(np.int32), clamp)


def stored(x_nchw, pitch, rng, dtype=np.int8):
    """[N,C,H,W] values -> NHWC rows of `pitch` channels (+ 64 elements of slack); channels >= C hold garbage (they meet zero weights)"""
    n, c, h, w = x_nchw.shape
    lo, hi = (-128, 128) if dtype == np.int8 else (-1000, 1000)
    out = rng.integers(lo, hi, n * h * w * pitch + 64).astype(dtype)
    out[:n * h * w * pitch].reshape(n, h, w, pitch)[..., :c] = x_nchw.transpose(0, 2, 3, 1)
    return out


def unstored(t, n, h, w, pitch, c):
    return t.cpu().numpy().astype(np.int64)[:n * h * w * pitch].reshape(n, h, w, pitch)[..., :c].transpose(0, 3, 1, 2)


def fast_table(s_w, s_out, bias, w_abs_sum, cp, in_max=128):
    """per-channel requant of a conv: exact (m, e) for the oracle and the packed fast-contract constants for the device"""
    from hawq_amd.packing import pack_ctab
    from hawq_amd.quant_utils import requant_table, tables_are_fast, tables_fit_fast
    vb = np.array([int(v).bit_length() for v in (w_abs_sum * in_max + np.abs(bias))], np.int64)
    # (numerator and denominator both scaled by 0.7: a non-trivial output scale gives m a full 31-bit mantissa, as in a real network)
    m, ek = requant_table(torch.ones(1), torch.from_numpy((s_w * 0.7).astype(f32)), torch.tensor([s_out * 0.7], dtype=torch.float32), vbits=vb)
    assert tables_fit_fast(m, ek, vb)
    pad = lambda v, fill=0: np.concatenate([np.asarray(v, np.int64), np.full(cp - len(v), fill, np.int64)])
    ctab = dev(pack_ctab(pad(bias), pad(m), pad(ek, 33)))
    return m, ek, ctab, (1 if tables_are_fast(m, ek, vb) else 5) | (0 if (np.asarray(ek) >> 8).any() else 8)


def fast_scalar(ratio, vmax):
    from hawq_amd.quant_utils import requant_table, tables_are_fast, tables_fit_fast
    vb = int(vmax).bit_length()
    m, ek = requant_table(torch.tensor([ratio * 0.7], dtype=torch.float32), torch.ones(1), torch.tensor([0.7]), vbits=vb)
    assert tables_fit_fast(m, ek, vb)
    return m, ek, not tables_are_fast(m, ek, vb)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("mode", ["closing_clamp16", "closing_identity", "requant"])
@pytest.mark.parametrize("widths", [(24, 32, 16, 16), (16, 16, 24, 32), (96, 96, 40, 48)])
def test_conv2d_on_narrow_tensors(lib, orc, widths, mode, fast):
    """ABI 4: a 1x1 conv whose K and N are padded to 64 reads rows of in_pitch bytes and writes rows of out_pitch channels.  The bytes a
    64-byte chunk reads beyond its row are the next pixel's (garbage here) and meet zero weights; channels >= out_pitch are not written
    (the buffer ends right behind the last row).  REQUANT on both its epilogues, and the signed RESIDUAL forms of MobileNetV2 - exact
    and with the fast contract's arithmetic on the direct epilogue (hawq_conv_args.fast_tables with a 32-bit carrier)."""
    from hawq_amd.packing import pack_conv_weight
    cin, ipitch, cout, opitch = widths
    rng = np.random.default_rng(cin * 7 + cout + len(mode) + fast)
    n, h, w = 2, 11, 9
    cin_p, cout_p = (cin + 63) // 64 * 64, (cout + 63) // 64 * 64
    x = rng.integers(0, 128, (n, cin, h, w)).astype(np.int64)
    wt = rng.integers(-127, 128, (cout, cin, 1, 1)).astype(np.int64)
    b = rng.integers(-3000, 3000, cout).astype(np.int64)
    acc = orc.conv2d(x, wt, b, 1, 0)
    s_w = rng.uniform(2e-4, 2e-3, cout)
    m, ek, ctab, fbits = fast_table(s_w, 0.7 if mode == "requant" else 0.002, b, np.abs(wt).reshape(cout, -1).sum(1), cout_p)
    pad = lambda v, fill=0: dev(np.concatenate([np.asarray(v, np.int64), np.full(cout_p - len(v), fill, np.int64)]).astype(np.int32))
    keep = dict(x=dev(stored(x, ipitch, rng)), w=dev(pack_conv_weight(wt, 8, cin_p, cout_p)), b=pad(b), m=pad(m), e=pad(ek, 33), ctab=ctab)
    a = lib.ConvArgs()
    a.in_, a.wgt, a.bias, a.m, a.e = (keep[k].data_ptr() for k in ("x", "w", "b", "m", "e"))
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad, a.in_bits, a.w_bits = n, h, w, cin_p, cout_p, 1, 1, 1, 0, 8, 8
    a.in_pitch, a.out_pitch = (ipitch if ipitch != cin_p else 0), (opitch if opitch != cout_p else 0)
    flags = torch.zeros(1, dtype=torch.int32, device='cuda')
    a.flags = flags.data_ptr()
    npx = n * h * w
    out_q = torch.full((npx * opitch,), 7, dtype=torch.int8, device='cuda')
    a.out_q, a.out_bits = out_q.data_ptr(), 8
    if fast:
        a.fast_tables, a.ctab = fbits & 7, ctab.data_ptr()
    else:
        a.n_valid = cout
    if mode == "requant":
        a.epilogue, a.relu, a.q_lo, a.q_hi = lib.EPI_REQUANT, 1, 0, 127
        ref_q = odyadic(orc, np.maximum(acc, 0), m, ek, (0, 127))
        out16 = None
    else:
        v = odyadic(orc, acc, m, ek)
        a.epilogue, a.res_no_relu = lib.EPI_RESIDUAL, 1
        if mode == "closing_identity":
            res = rng.integers(-30000, 30000, (n, cout, h, w)).astype(np.int64)
            m1, e1, tie1 = fast_scalar(0.81, 32768)
            keep['res'] = dev(stored(res, opitch, rng, np.int32))
            if opitch > cout:   # padding channels of the carrier are zeros by contract
                keep['res'][:npx * opitch].view(npx, opitch)[:, cout:] = 0
            a.res_in, a.res_in_bits, a.m_id_scalar, a.e_id_scalar = keep['res'].data_ptr(), 32, int(m1[0]), int(e1[0])
            v = v + odyadic(orc, res, m1, e1)
        else:
            a.res_clamp16 = 1
            v = np.clip(v, -32768, 32767)
            tie1 = False
        mq, eq, tieq = fast_scalar(0.0031, int(np.abs(v).max()) + 1)
        if fast and (tie1 or tieq):
            a.fast_tables |= 4
        ref_q = odyadic(orc, v, mq, eq, (-128, 127))
        out16 = torch.full((npx * opitch,), 7, dtype=torch.int32, device='cuda')
        a.res_out, a.res_out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out16.data_ptr(), 32, -128, 127, int(mq[0]), int(eq[0])
    tiles = range(1, lib.load().hawq_conv2d_num_tiles() - lib.load().hawq_conv2d_num_band_tiles() + 1)
    ran = 0
    for tile in [0] + list(tiles):
        a.tile = tile
        out_q.fill_(7)
        if lib.load().hawq_conv2d(C.byref(a), stream()) != 0:
            assert tile != 0, lib.load().hawq_last_error().decode()
            continue
        ran += 1
        assert np.array_equal(unstored(out_q, n, h, w, opitch, cout), ref_q), tile
        if opitch > cout:
            assert not unstored(out_q, n, h, w, opitch, opitch)[:, cout:].any(), tile
        if out16 is not None:
            assert np.array_equal(unstored(out16, n, h, w, opitch, cout), v), tile
    assert ran >= 2
    # a pitch the layer cannot have is refused
    a.tile, a.out_pitch = 0, 24
    assert lib.load().hawq_conv2d(C.byref(a), None) != 0
    a.out_pitch, a.in_pitch = 0, cin_p + 16
    assert lib.load().hawq_conv2d(C.byref(a), None) != 0


def bottleneck_reference(orc, x, w1, b1, t1, hi1, w2, b2, t2, hi2, w3, b3, t3, stride, res, tid, clamp16, tq, q_rng):
    from oracle import oracle_mbv2
    h1 = odyadic(orc, np.maximum(orc.conv2d(x, w1, b1, 1, 0), 0), t1[0], t1[1], (0, hi1))
    h2 = odyadic(orc, np.maximum(oracle_mbv2.depthwise3x3(h1, w2, b2, stride), 0), t2[0], t2[1], (0, hi2))
    v = odyadic(orc, orc.conv2d(h2, w3, b3, 1, 0), t3[0], t3[1])
    if res is not None:
        v = v + odyadic(orc, res, tid[0], tid[1])
    if clamp16:
        v = np.clip(v, -32768, 32767)
    return v, odyadic(orc, v, tq[0], tq[1], q_rng)


# cin, in_pitch, hidden, cout, out_pitch, stride, identity, (n, h, w), small hidden-side weights (-> requant ratios >= 1/4: pre-shifts)
UNITS = [(32, 32, 32, 16, 16, 1, False, (2, 20, 37), False),       # unit 1 of the width-1 network: one slice, ragged tiles on both axes
         (16, 16, 96, 24, 32, 2, False, (2, 21, 35), False),       # stride 2 from odd sizes
         (24, 32, 144, 24, 32, 1, True, (1, 17, 16), False),       # identity; the last slice is half padding (144 = 4.5 x 32)
         (24, 32, 144, 32, 32, 2, False, (2, 13, 18), True),       # per-channel pre-shifts on conv1 and the depthwise conv
         (64, 64, 384, 64, 64, 1, True, (3, 14, 14), False),       # 12 slices, 64-wide input and output
         (32, 32, 192, 64, 64, 2, False, (2, 28, 28), False),
         (96, 96, 576, 96, 96, 1, True, (2, 14, 14), False),       # three K steps, three output blocks (planar organisation only)
         (64, 64, 384, 96, 96, 1, False, (1, 7, 9), True),
         (160, 160, 960, 160, 160, 1, True, (2, 7, 7), False),      # the 7 x 7 units of the width-1 network: 8 x 8 tiles, 5 K steps, 5 output blocks
         (160, 160, 960, 320, 320, 1, False, (1, 7, 7), False),     # ten output blocks
         (96, 96, 576, 160, 160, 2, False, (2, 14, 14), False),     # stride 2 onto a 7 x 7 map
         (160, 160, 128, 160, 160, 1, True, (1, 5, 8), True)]       # as many slices as slice groups, a ragged map, pre-shifts


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("unit", UNITS)
def test_linear_bottleneck_against_the_oracle(lib, orc, unit, tile):
    """hawq_linear_bottleneck: conv1 1x1 + ReLU + QuantAct, depthwise 3x3 + ReLU + QuantAct, conv3 1x1, quant_act_int32 (identity or
    the 16-bit clamp), next block-input QuantAct - against the oracle's exact integer layers chained on the host.  tile 0: the default
    (planar hidden tensor, slice groups by workgroup count: 4 at these sizes), 1: [pixel][channel] organisation, 2 / 3 / 4: planar with
    1 / 2 / 4 slice groups."""
    from hawq_amd.packing import pack_conv_weight
    cin, ipitch, hid, cout, opitch, stride, identity, (n, h, w), small = unit
    if tile == 1 and (ipitch > 64 or opitch > 64):
        pytest.skip("the first organisation takes at most 64-channel inputs and outputs")
    if tile > 1 and (ipitch > 96 or opitch > 96):
        pytest.skip("the wide units have one organisation (8 x 8 tiles, two slice groups)")
    rng = np.random.default_rng(cin + hid + cout + stride)
    cin_p, hid_p, cout_p = (cin + 63) // 64 * 64, (hid + 63) // 64 * 64, (cout + 63) // 64 * 64
    x = rng.integers(0, 128, (n, cin, h, w)).astype(np.int64)
    wmax, bmax = (2, 40) if small else (127, 3000)
    w1 = rng.integers(-wmax, wmax + 1, (hid, cin, 1, 1)).astype(np.int64)
    b1 = rng.integers(-bmax, bmax, hid).astype(np.int64)
    w2 = rng.integers(-wmax, wmax + 1, (hid, 1, 3, 3)).astype(np.int64)
    b2 = rng.integers(-bmax, bmax, hid).astype(np.int64)
    w3 = rng.integers(-127, 128, (cout, hid, 1, 1)).astype(np.int64)
    b3 = rng.integers(-3000, 3000, cout).astype(np.int64)
    # requant ratios that use the 0 .. 127 range of the hidden activations (and saturate some of them)
    m1, e1, ct1, f1 = fast_table(rng.uniform(0.6, 1.6, hid) * 127 / (np.abs(w1).reshape(hid, -1).sum(1) * 40 + 1), 1.0, b1, np.abs(w1).reshape(hid, -1).sum(1), hid_p)
    m2, e2, ct2, f2 = fast_table(rng.uniform(0.6, 1.6, hid) * 127 / (np.abs(w2).reshape(hid, -1).sum(1) * 40 + 1), 1.0, b2, np.abs(w2).reshape(hid, -1).sum(1), hid_p)
    m3, e3, ct3, f3 = fast_table(rng.uniform(0.5, 3.0, cout) * (45000 if identity else 150000) * (2 if hid >= 900 else 1) / (np.abs(w3).reshape(cout, -1).sum(1) * 30 + 1), 1.0, b3,
                                 np.abs(w3).reshape(cout, -1).sum(1), cout_p)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = rng.integers(-30000, 30000, (n, cout, h, w)).astype(np.int64) if identity else None
    tid = fast_scalar(0.81, 32768) if identity else (np.zeros(1), np.full(1, 33), False)
    hi1, hi2 = 127, 100
    v, _ = bottleneck_reference(orc, x, w1, b1, (m1, e1), hi1, w2, b2, (m2, e2), hi2, w3, b3, (m3, e3), stride, res, tid, not identity,
                                (np.ones(1), np.full(1, 33)), (-128, 127))
    tq = fast_scalar(0.0031, int(np.abs(v).max()) + 1)
    v, ref_q = bottleneck_reference(orc, x, w1, b1, (m1, e1), hi1, w2, b2, (m2, e2), hi2, w3, b3, (m3, e3), stride, res, tid, not identity,
                                    tq, (-128, 127))
    w9 = np.zeros((9, hid_p), np.int8)
    w9[:, :hid] = w2.reshape(hid, 9).T
    keep = dict(x=dev(stored(x, ipitch, rng)), w1=dev(pack_conv_weight(w1, 8, cin_p, hid_p)), w9=dev(w9), w3=dev(pack_conv_weight(w3, 8, hid_p, cout_p)))
    if identity:
        keep['res'] = dev(stored(res, opitch, rng, np.int32))
        if opitch > cout:
            keep['res'][:n * h * w * opitch].view(-1, opitch)[:, cout:] = 0
    b = lib.BottleneckArgs()
    e, q = b.expand, b.project
    e.in_, e.wgt, e.ctab = keep['x'].data_ptr(), keep['w1'].data_ptr(), ct1.data_ptr()
    e.N, e.H, e.W, e.Cin, e.Cout, e.KH, e.KW, e.stride, e.pad, e.in_bits, e.w_bits = n, h, w, cin_p, hid_p, 1, 1, 1, 0, 8, 8
    e.in_pitch = ipitch if ipitch != cin_p else 0
    e.epilogue, e.relu, e.q_lo, e.q_hi, e.out_bits, e.fast_tables = lib.EPI_REQUANT, 1, 0, hi1, 8, f1
    b.dw_wgt9c, b.dw_ctab, b.dw_stride, b.dw_q_lo, b.dw_q_hi, b.dw_fast_tables, b.c_mid, b.tile = keep['w9'].data_ptr(), ct2.data_ptr(), stride, 0, hi2, f2, hid, tile
    q.wgt, q.ctab = keep['w3'].data_ptr(), ct3.data_ptr()
    q.N, q.H, q.W, q.Cin, q.Cout, q.KH, q.KW, q.stride, q.pad, q.in_bits, q.w_bits = n, ho, wo, hid_p, cout_p, 1, 1, 1, 0, 8, 8
    q.out_pitch = opitch if opitch != cout_p else 0
    # (stride-2 units: the exact-tie instantiation although no table needs it - it is exact round-half-even for any table)
    q.epilogue, q.res_no_relu, q.res_clamp16, q.fast_tables = lib.EPI_RESIDUAL, 1, int(not identity), (f3 & 7) | (4 if (tid[2] or tq[2] or stride == 2) else 0)
    out16 = torch.full((n * ho * wo * opitch,), 7, dtype=torch.int32, device='cuda')
    out_q = torch.full((n * ho * wo * opitch,), 7, dtype=torch.int8, device='cuda')
    q.res_out, q.res_out_bits, q.out_q, q.out_bits, q.q_lo, q.q_hi, q.mq, q.eq = out16.data_ptr(), 32, out_q.data_ptr(), 8, -128, 127, int(tq[0][0]), int(tq[1][0])
    if identity:
        q.res_in, q.res_in_bits, q.m_id_scalar, q.e_id_scalar = keep['res'].data_ptr(), 32, int(tid[0][0]), int(tid[1][0])
    assert lib.load().hawq_linear_bottleneck_ok(C.byref(b)) == 1, lib.load().hawq_last_error().decode()
    lib.call("hawq_linear_bottleneck", C.byref(b), stream())
    assert np.array_equal(unstored(out16, n, ho, wo, opitch, cout), v)
    assert np.array_equal(unstored(out_q, n, ho, wo, opitch, cout), ref_q)
    assert np.abs(v).max() >= 32767   # the 16-bit clamp acted / the un-clamped sum left the 16-bit range
    assert bool(((np.asarray(e1) | np.asarray(e2)) >> 8).any()) == small   # hidden-side pre-shifts exactly where intended (the other instantiation)
    # refusals: what the launch does not take is an error, never a silent different result
    b.dw_stride = 3
    assert lib.load().hawq_linear_bottleneck_ok(C.byref(b)) == 0 and lib.load().hawq_linear_bottleneck(C.byref(b), None) != 0
    b.dw_stride = stride
    q.fast_tables = 0
    assert lib.load().hawq_linear_bottleneck(C.byref(b), None) != 0


@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("hw", [(224, 224), (33, 46), (9, 7)])
def test_stem3x3s2_against_the_oracle(lib, orc, hw, u8):
    """hawq_stem3x3s2: input QuantAct + 3x3 / stride 2 / pad 1 conv on 3 channels + ReLU + quant_act_int32 (16-bit clamp) + the first
    unit's QuantAct in one launch, from fp32 NCHW images and from uint8 NHWC images through the look-up table."""
    from hawq_amd.packing import pack_conv_weight
    from hawq_amd.quant_utils import input_quant_lut
    H, W = hw
    rng = np.random.default_rng(H * 3 + W + u8)
    n, cout, opitch = 3, 32, 32
    inv_s = f32(1.0) / f32(0.0207)
    if u8:
        xu = rng.integers(0, 256, (n, H, W, 3)).astype(np.uint8)
        lut = input_quant_lut(float(inv_s), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
        xq = np.stack([lut[c].numpy().astype(np.int64)[xu[..., c]] for c in range(3)], 1)
    else:
        xf = (rng.standard_normal((n, 3, H, W)) * 1.3).astype(f32)
        xq = np.clip(np.rint((inv_s * xf).astype(f32)), -128, 127).astype(np.int64)
    wt = rng.integers(-127, 128, (cout, 3, 3, 3)).astype(np.int64)
    bias = rng.integers(-3000, 3000, cout).astype(np.int64)
    acc = orc.conv2d(xq, wt, bias, 2, 1)
    m, ek, ctab, fb = fast_table(rng.uniform(0.5, 3.0, cout) * 20000 / (np.abs(wt).reshape(cout, -1).sum(1) * 40 + 1), 1.0, bias, np.abs(wt).reshape(cout, -1).sum(1), 64)
    v = np.clip(np.maximum(odyadic(orc, acc, m, ek), 0), -32768, 32767)
    tq = fast_scalar(0.0031, 32768)
    ref_q = odyadic(orc, v, tq[0], tq[1], (0, 127))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wrow = np.ascontiguousarray(wt.transpose(0, 2, 3, 1)).reshape(cout, 27, 1, 1)   # (kh, kw, c) order: the im2col path's K = 64 rows
    keep = dict(w=dev(pack_conv_weight(wrow, 8, 64, 64)))
    a = lib.ConvArgs()
    a.wgt, a.ctab = keep['w'].data_ptr(), ctab.data_ptr()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad, a.in_bits, a.w_bits = n, Ho, Wo, 64, 64, 1, 1, 1, 0, 8, 8
    a.epilogue, a.res_no_relu, a.res_clamp16, a.fast_tables, a.out_pitch = lib.EPI_RESIDUAL, 0, 1, (fb & 7) | (4 if tq[2] else 0), opitch
    out16 = torch.full((n * Ho * Wo * opitch,), 7, dtype=torch.int32, device='cuda')
    out_q = torch.full((n * Ho * Wo * opitch,), 7, dtype=torch.int8, device='cuda')
    a.res_out, a.res_out_bits, a.out_q, a.out_bits, a.q_lo, a.q_hi, a.mq, a.eq = out16.data_ptr(), 32, out_q.data_ptr(), 8, 0, 127, int(tq[0][0]), int(tq[1][0])
    if u8:
        keep['x'], keep['lut'] = dev(xu), lut.reshape(-1).cuda()
        args = (None, keep['x'].data_ptr(), keep['lut'].data_ptr(), H, W, 0.0, 0, 0)
    else:
        keep['x'] = dev(xf)
        args = (keep['x'].data_ptr(), None, None, H, W, float(inv_s), -128, 127)
    assert lib.load().hawq_stem3x3s2_ok(args[0], args[1], args[2], H, W, C.byref(a)) == 1
    lib.call("hawq_stem3x3s2", *args, C.byref(a), stream())
    assert np.array_equal(unstored(out16, n, Ho, Wo, opitch, cout), v)
    assert np.array_equal(unstored(out_q, n, Ho, Wo, opitch, cout), ref_q)
    assert v.max() == 32767 and (v == 0).any()
    # both image forms at once, or a grid that is not the conv's, are refused
    assert lib.load().hawq_stem3x3s2(keep['x'].data_ptr(), keep['x'].data_ptr(), None, H, W, float(inv_s), -128, 127, C.byref(a), None) != 0
    a.H += 1
    assert lib.load().hawq_stem3x3s2(*args, C.byref(a), None) != 0


@pytest.mark.parametrize("shape", [(2, 9, 7, 32, 32, 1), (1, 14, 14, 96, 96, 2), (3, 7, 7, 160, 144, 1), (1, 113, 57, 32, 24, 2)])
@pytest.mark.parametrize("tie", [False, True])
def test_depthwise3x3_requant_fast_against_the_oracle(lib, orc, shape, tie):
    """hawq_depthwise3x3_requant_fast: depthwise 3x3 + ReLU + QuantAct with the host-proved short requant (fused constants, bias folded
    in) - against the oracle's depthwise conv + round-half-even dyadic; tie: the exact-tie instantiation (exact for any table)."""
    from oracle import oracle_mbv2
    n, h, w, pitch, c, stride = shape
    rng = np.random.default_rng(h * w + c + tie)
    x = rng.integers(0, 128, (n, c, h, w)).astype(np.int64)
    wt = rng.integers(-127, 128, (c, 1, 3, 3)).astype(np.int64)
    b = rng.integers(-3000, 3000, c).astype(np.int64)
    m, ek, ctab, f = fast_table(rng.uniform(0.6, 1.6, c) * 127 / (np.abs(wt).reshape(c, -1).sum(1) * 40 + 1), 1.0, b, np.abs(wt).reshape(c, -1).sum(1), pitch)
    ref = odyadic(orc, np.maximum(oracle_mbv2.depthwise3x3(x, wt, b, stride), 0), m, ek, (0, 100))
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    w9 = np.zeros((9, pitch), np.int8)
    w9[:, :c] = wt.reshape(c, 9).T
    keep = dict(x=dev(stored(x, pitch, rng)), w=dev(w9))
    if pitch > c:   # padding channels of a stored activation tensor are zeros
        keep['x'][:n * h * w * pitch].view(-1, pitch)[:, c:] = 0
    out = torch.full((n * ho * wo * pitch,), 7, dtype=torch.int8, device='cuda')
    lib.call("hawq_depthwise3x3_requant_fast", keep['x'].data_ptr(), keep['w'].data_ptr(), ctab.data_ptr(), (f & 7) | (4 if tie else 0), n, h, w, pitch, c, stride,
             0, 100, out.data_ptr(), stream())
    assert np.array_equal(unstored(out, n, ho, wo, pitch, c), ref)
    assert not unstored(out, n, ho, wo, pitch, pitch)[:, c:].any()
    assert (ref == 100).any() and (ref == 0).any()
    assert lib.load().hawq_depthwise3x3_requant_fast(keep['x'].data_ptr(), keep['w'].data_ptr(), ctab.data_ptr(), 0, n, h, w, pitch, c, stride, 0, 100, out.data_ptr(), None) != 0
    assert lib.load().hawq_depthwise3x3_requant_fast(keep['x'].data_ptr(), keep['w'].data_ptr(), ctab.data_ptr(), 1, n, h, w, pitch, c, stride, -5, 100, out.data_ptr(), None) != 0
