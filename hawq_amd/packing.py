"""Host-side packing of integer parameters into the layouts the gfx950 kernels read.

Layouts (include/hawq_mi355.h): conv weights [Cout][KH][KW][Cin] with Cin contiguous (the GEMM K
axis), int8 or "hawq4" nibble-packed; stem weights [64][7][8][4] int8.  The reference's own
deployment packing (tvm_benchmark/hawq_utils_resnet50.py:21-42, 111-153: OIHW->HWOI, eight
nibbles per int32 big-endian) targets TVM's WMMA kernels and is deliberately not reused: hawq4
is chosen so that two AND/shift ops turn one dword into two int8x4 MFMA operands in order.
"""
from __future__ import annotations

import numpy as np


def pack_hawq4(v: np.ndarray) -> np.ndarray:
    """[..., C] integers in [-8, 15] (C % 8 == 0) -> [..., C//2] uint8, hawq4 nibble order:
    byte k of each 8-channel group = (c_k & 15) | (c_{k+4} & 15) << 4."""
    assert v.shape[-1] % 8 == 0
    g = (np.asarray(v).astype(np.int64) & 0xF).astype(np.uint8).reshape(v.shape[:-1] + (v.shape[-1] // 8, 8))
    b = g[..., 0:4] | (g[..., 4:8] << 4)
    return np.ascontiguousarray(b.reshape(v.shape[:-1] + (v.shape[-1] // 2,)))


def unpack_hawq4(b: np.ndarray, signed: bool = False) -> np.ndarray:
    """Inverse of pack_hawq4: [..., C//2] uint8 -> [..., C] int32."""
    g = np.asarray(b, np.uint8).reshape(b.shape[:-1] + (b.shape[-1] // 4, 4))
    lo, hi = (g & 0xF).astype(np.int32), (g >> 4).astype(np.int32)
    v = np.concatenate([lo, hi], axis=-1).reshape(b.shape[:-1] + (b.shape[-1] * 2,))
    if signed:
        v = np.where(v >= 8, v - 16, v)
    return v


def pack_conv_weight(w_int: np.ndarray, w_bits: int, cin_pad: int | None = None, cout_pad: int | None = None):
    """OIHW integer weights -> [Cout_p][KH][KW][Cin_p] int8 bytes (w_bits 8) or hawq4 (w_bits 4);
    channel padding is zero-filled.  Returns a flat uint8 array."""
    w = np.rint(np.asarray(w_int, np.float64)).astype(np.int64)
    co, ci, kh, kw = w.shape
    cin_pad = cin_pad or ci
    cout_pad = cout_pad or co
    out = np.zeros((cout_pad, kh, kw, cin_pad), np.int64)
    out[:co, :, :, :ci] = w.transpose(0, 2, 3, 1)
    if w_bits == 8:
        assert out.min() >= -128 and out.max() <= 127
        return np.ascontiguousarray(out.astype(np.int8)).view(np.uint8).reshape(-1)
    assert w_bits == 4 and out.min() >= -8 and out.max() <= 7, "4-bit weights must lie in [-8, 7]"
    return pack_hawq4(out).reshape(-1)


def pack_w3x3_band(w_packed: np.ndarray, cout: int, cin: int) -> np.ndarray:
    """[Cout][3][3][Cin] int8 bytes (pack_conv_weight's layout) -> the stream the round-5 3x3 kernels consume
    (include/hawq_mi355.h: hawq_conv_args.wgt_band): [Cout/64][Cin/64][kh][kw][64 rows][64 B], the four 16-byte slots of
    row r stored at slot ^ ((r >> 2) & 3) - every weight-ring LDS-DMA instruction then copies one contiguous KiB and the tile
    arrives in LDS already in the bank-conflict-free order the MFMA fragment reads expect.  Same integers, only re-ordered
    (the C twin is hawq_pack_w3x3_band)."""
    assert cout % 64 == 0 and cin % 64 == 0
    w = np.asarray(w_packed, np.uint8).reshape(cout // 64, 64, 9, cin // 64, 4, 16)      # ct, r, tap, cc, slot, byte
    r = np.arange(64)
    src_slot = np.arange(4)[None, :] ^ ((r[:, None] >> 2) & 3)                           # stored slot s holds logical slot s ^ sw(r)
    w = w[:, r[:, None], :, :, src_slot, :]                                             # -> r, s, ct, tap, cc, byte
    w = w.transpose(2, 4, 3, 0, 1, 5)                                                    # ct, cc, tap, r, s, byte
    return np.ascontiguousarray(w).reshape(-1)


def pack_w1x1_k128(w_packed: np.ndarray, cout: int, cin: int) -> np.ndarray:
    """[Cout][Cin] int8 bytes (pack_conv_weight's layout of a 1x1 conv) -> the stream the round-5 1x1 kernels consume
    (include/hawq_mi355.h: hawq_conv_args.wgt_k128): [Cout/64][Cin/128][64 rows][128 B], the eight 16-byte slots of row r stored at
    slot ^ ((r >> 1) & 7).  Same integers, only re-ordered (the C twin is hawq_pack_w1x1_k128)."""
    assert cout % 64 == 0 and cin % 128 == 0
    w = np.asarray(w_packed, np.uint8).reshape(cout // 64, 64, cin // 128, 8, 16)       # g, r, ch, slot, byte
    r = np.arange(64)
    src_slot = np.arange(8)[None, :] ^ ((r[:, None] >> 1) & 7)                          # stored slot s holds logical slot s ^ sw(r)
    w = w[:, r[:, None], :, src_slot, :]                                                # -> r, s, g, ch, byte
    w = w.transpose(2, 3, 0, 1, 4)                                                      # g, ch, r, s, byte
    return np.ascontiguousarray(w).reshape(-1)


def pack_stem_weight(w_int: np.ndarray) -> np.ndarray:
    """[64][3][7][7] integer stem weights -> [64][7][8][4] int8 (kw and c zero padded)."""
    w = np.rint(np.asarray(w_int, np.float64)).astype(np.int64)
    co, ci, kh, kw = w.shape
    assert (co, kh, kw) == (64, 7, 7) and ci <= 4
    out = np.zeros((64, 7, 8, 4), np.int8)
    out[:, :, :7, :ci] = w.transpose(0, 2, 3, 1)
    return np.ascontiguousarray(out).view(np.uint8).reshape(-1)


def pack_ctab(bias, m, ek) -> np.ndarray:
    """Fused per-channel requant constants of the fast conv epilogues: int32 [C][4] =
    {m, (e - 32) | k << 8, lo32(Cc), hi32(Cc)} with Cc = (bias << k)*m + 2^(e-1), so that
    q = hi32((acc << k)*m + Cc) >> (e-32)  ==  round_half_up((((acc + bias) << k) * m) / 2^e)
    (one shift, one v_mad_i64_i32, one v_ashrrev_i32).  Only valid for tables that satisfy the fast
    contract (e >= 33, no tie possible)."""
    bias = np.asarray(bias, np.int64).reshape(-1)
    m = np.asarray(m, np.int64).reshape(-1)
    ek = np.asarray(ek, np.int64).reshape(-1)
    e, k = ek & 0xff, ek >> 8
    assert (e >= 33).all() and (e <= 62).all() and (k >= 0).all(), "pack_ctab needs fast-contract tables"
    cc = (bias << k) * m + (np.int64(1) << (e - 1))
    out = np.empty((bias.size, 4), np.int32)
    out[:, 0] = m.astype(np.int32)
    out[:, 1] = ((e - 32) | (k << 8)).astype(np.int32)
    out[:, 2] = (cc & 0xffffffff).astype(np.uint32).view(np.int32)
    out[:, 3] = (cc >> 32).astype(np.int32)
    return out
