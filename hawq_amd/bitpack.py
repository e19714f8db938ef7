"""Bit-packed archives of HAWQ integer checkpoints.

README.md:61 of the reference: "Checkpoints in model zoo are saved in floating point precision.  To shrink the memory
size, BitPack (github.com/Zhen-Dong/BitPack) can be applied on weight_integer tensors, or directly on
quantized_checkpoint.pth.tar".  BitPack is an external tool that is not part of the reference tree (nor installable
here), so its on-disk format cannot be pinned: this module restates the IDEA - every ``weight_integer`` tensor of a
``quantized_checkpoint.pth.tar`` (quant_train.py:665-670; fp32 tensors holding b-bit integers) stored as a dense stream
of b-bit two's-complement fields - in a self-describing format of its own (``FORMAT`` below), and round-trips it.

    pack_quantized_checkpoint(ckpt, "packed.pth.tar")        # 4x (W8) ... 8x (W4) smaller than the fp32 file
    load_packed_checkpoint(model, "packed.pth.tar")          # -> hawq_amd.api.load_quantized_checkpoint

Host-side I/O only (numpy); nothing here runs on the inference path.
"""
from __future__ import annotations

import numpy as np
import torch

FORMAT = "hawq_amd.bitpack/1: fields of `bits` bits, two's complement, element i at stream bits [i*bits, (i+1)*bits), " \
         "stream bit k = bit (k % 8) of byte k // 8 (LSB first)"


def needed_bits(t) -> int:
    """Smallest two's-complement width (>= 2) that holds every value of an integer-valued tensor."""
    a = np.asarray(t, np.float64)
    if a.size == 0:
        return 2
    if not np.array_equal(a, np.rint(a)):
        raise ValueError("not an integer-valued tensor")
    lo, hi = int(a.min()), int(a.max())
    b = 2
    while lo < -(1 << (b - 1)) or hi > (1 << (b - 1)) - 1:
        b += 1
    return b


def pack_tensor(t, bits: int) -> np.ndarray:
    """Integer-valued tensor -> uint8 stream (FORMAT)."""
    a = np.rint(np.asarray(t, np.float64)).astype(np.int64).reshape(-1)
    if a.size and (a.min() < -(1 << (bits - 1)) or a.max() > (1 << (bits - 1)) - 1):
        raise ValueError(f"values do not fit {bits} bits")
    u = (a & ((1 << bits) - 1)).astype(np.uint64)
    fields = ((u[:, None] >> np.arange(bits, dtype=np.uint64)[None, :]) & 1).astype(np.uint8).reshape(-1)
    return np.packbits(fields, bitorder="little")


def unpack_tensor(stream, bits: int, shape) -> np.ndarray:
    """Inverse of pack_tensor -> int32 array of `shape`."""
    n = int(np.prod(shape))
    fields = np.unpackbits(np.asarray(stream, np.uint8), count=n * bits, bitorder="little").reshape(n, bits).astype(np.int64)
    u = (fields << np.arange(bits, dtype=np.int64)[None, :]).sum(1)
    u = np.where(u >= (1 << (bits - 1)), u - (1 << bits), u)
    return u.astype(np.int32).reshape(shape)


def pack_quantized_checkpoint(ckpt, path=None, bits=None) -> dict:
    """``ckpt``: a quantized_checkpoint.pth.tar (path or loaded dict of the reference's five groups).  Every
    ``weight_integer`` tensor becomes {"packed", "bits", "shape"}; ``bits`` maps state_dict keys to widths (default: the
    smallest width that holds the tensor's values, i.e. the layer's weight_bit or less).  The other groups (scales, int32
    biases, activation scales) stay as they are.  Returns the archive dict and, with ``path``, saves it."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    out = {"format": FORMAT}
    for group, entries in ckpt.items():
        if group != "weight_integer":
            out[group] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in entries.items()}
            continue
        packed = {}
        for k, v in entries.items():
            a = v.detach().cpu().numpy()
            b = int(bits[k]) if bits and k in bits else needed_bits(a)
            packed[k] = {"packed": torch.from_numpy(pack_tensor(a, b)), "bits": b, "shape": tuple(a.shape)}
        out[group] = packed
    if path is not None:
        torch.save(out, path)
    return out


def unpack_quantized_checkpoint(archive) -> dict:
    """Archive (path or dict) -> the reference's quantized-checkpoint dict (fp32 tensors holding integers)."""
    if not isinstance(archive, dict):
        archive = torch.load(archive, map_location="cpu")
    if archive.get("format") != FORMAT:
        raise KeyError("not a hawq_amd.bitpack archive (or an unknown format version)")
    out = {}
    for group, entries in archive.items():
        if group == "format":
            continue
        if group != "weight_integer":
            out[group] = dict(entries)
            continue
        out[group] = {k: torch.from_numpy(unpack_tensor(e["packed"].numpy(), e["bits"], e["shape"]).astype(np.float32))
                      for k, e in entries.items()}
    return out


def load_packed_checkpoint(model, archive, strict: bool = True):
    """Restore a frozen network from a packed archive alone (see hawq_amd.api.load_quantized_checkpoint)."""
    from .api import load_quantized_checkpoint
    return load_quantized_checkpoint(model, unpack_quantized_checkpoint(archive), strict=strict)
