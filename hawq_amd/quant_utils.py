"""Quantisation math of the frozen forward - host side of hawq_amd.

Mirrors the public names of the reference's utils/quantization_utils/quant_utils.py so that
code written against it keeps working, but splits the work MI355X-first:

* everything that depends only on parameters (scales, folded/quantised weights, dyadic
  requantisation tables) is computed ONCE on the host in binary32/binary64, not per forward;
* everything that touches activations runs in HIP kernels (hawq_amd/csrc) - the
  ``*.apply`` entry points below dispatch there for device tensors and raise otherwise.

Reference lines: symmetric_/asymmetric_linear_quantization_params quant_utils.py:128-185,
linear_quantize :73-97, batch_frexp :188-213, SymmetricQuantFunction :231-258,
AsymmetricQuantFunction :275-308, fixedpoint_fn :344-456 (forward paths only; the STE
backward passes are QAT and out of scope).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


# --------------------------------------------------------------------------- range statistics
_SCRATCH = {}


def _stat_scratch(device):
    """Scratch of the statistics kernels, [2 floats out | 260 uint32 state + histogram], one per (device, current stream):
    two QuantAct statistics calls on different streams of one device must not share the histogram / state words."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _SCRATCH:
        _SCRATCH[key] = (torch.zeros(2, dtype=torch.float32, device=device), torch.zeros(264, dtype=torch.int32, device=device))
    return _SCRATCH[key]


def device_min_max(x: torch.Tensor):
    """(x.min(), x.max()) of a CUDA fp32 tensor as 0-dim device tensors (hawq_minmax_f32; quant_modules.py:233-236)."""
    if not x.is_cuda:
        raise RuntimeError("hawq_amd: activation statistics run on the MI355X only (no CPU path)")
    x = x.contiguous().float()
    if x.data_ptr() % 16:   # an offset view: the kernel reads 16 bytes per lane
        x = x.clone()
    out, scratch = _stat_scratch(x.device)
    _lib.call("hawq_minmax_f32", x.data_ptr(), x.numel(), out.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    r = out.clone()
    return r[0], r[1]


def get_percentile_min_max(input, lower_percentile, upper_percentile, output_tensor=False):
    """The reference's percentile range (quant_utils.py:38-70), same signature: `input` a flat tensor,
    upper bound = kthvalue(input, round(n * upper%)), lower bound = -kthvalue(-input, round(n * (1 - lower%)))
    (0 when lower_percentile == 0).  CUDA tensors go through hawq_kthvalue_f32 (exact radix select, no sort, no host
    round trip); host tensors (weight preparation) use torch.kthvalue."""
    input_length = input.shape[0]
    lower_index = round(input_length * (1 - lower_percentile * 0.01))
    upper_index = round(input_length * upper_percentile * 0.01)
    if input.is_cuda:
        x = input.contiguous().float()
        out, scratch = _stat_scratch(x.device)
        sp = torch.cuda.current_stream(x.device).cuda_stream
        _lib.call("hawq_kthvalue_f32", x.data_ptr(), x.numel(), upper_index, 0, out.data_ptr(), scratch.data_ptr(), sp)
        upper_bound = out[0].clone()
        if lower_percentile == 0:
            lower_bound = upper_bound * 0
        else:
            _lib.call("hawq_kthvalue_f32", x.data_ptr(), x.numel(), lower_index, 1, out.data_ptr(), scratch.data_ptr(), sp)
            lower_bound = out[0].clone()
    else:
        upper_bound = torch.kthvalue(input, k=upper_index).values
        lower_bound = upper_bound * 0 if lower_percentile == 0 else -torch.kthvalue(-input, k=lower_index).values
    if not output_tensor:
        lower_bound, upper_bound = lower_bound.item(), upper_bound.item()
    return lower_bound, upper_bound


# --------------------------------------------------------------------------- scale formulas
def symmetric_linear_quantization_params(num_bits, saturation_min, saturation_max, per_channel=False):
    """S = clamp(max(|min|,|max|), 1e-8) / (2^(b-1)-1)   (quant_utils.py:128-152)."""
    with torch.no_grad():
        n = 2 ** (num_bits - 1) - 1
        if per_channel:
            scale, _ = torch.max(torch.stack([saturation_min.abs(), saturation_max.abs()], dim=1), dim=1)
        else:
            scale = torch.maximum(saturation_min.abs(), saturation_max.abs())
        return torch.clamp(scale, min=1e-8) / n


def asymmetric_linear_quantization_params(num_bits, saturation_min, saturation_max, integral_zero_point=True):
    """S = clamp(max-min, 1e-8) / (2^b-1); the zero point is computed for API parity but the
    integer path never applies it (quant_utils.py:155-185, 296-304)."""
    with torch.no_grad():
        n = 2 ** num_bits - 1
        scale = torch.clamp(saturation_max - saturation_min, min=1e-8) / float(n)
        zero_point = -saturation_min / scale
        if integral_zero_point:
            zero_point = zero_point.round() if isinstance(zero_point, torch.Tensor) else float(round(zero_point))
        return scale, zero_point


# --------------------------------------------------------------------- dyadic requant tables
def batch_frexp(inputs):
    """(mantissa*2^31 rounded half-up, 31 - exponent) of each scale (quant_utils.py:188-213).

    Host computation, vectorised: mant*2^31 is exact in binary64 with <= 22 fraction bits, so
    floor(v + 0.5) equals the reference's Decimal ROUND_HALF_UP.  Returns int64 / float64
    tensors shaped like ``inputs`` (on ``inputs.device``) as the reference does."""
    shape = inputs.size()
    mant, ex = np.frexp(inputs.detach().reshape(-1).cpu().numpy().astype(np.float64))
    m = np.floor(mant * 2147483648.0 + 0.5).astype(np.int64)
    e = 31.0 - ex
    return (torch.from_numpy(m).to(inputs.device).view(shape),
            torch.from_numpy(np.asarray(e, np.float64)).to(inputs.device).view(shape))


def requant_table(pre_act_scaling_factor, pre_weight_scaling_factor, z_scaling_factor, vbits=24, lift=True):
    """Device-ready (m, ek) int32 numpy tables of fixedpoint_fn (quant_utils.py:394-404):
    r = dbl(fl(S_a*S_w)) / dbl(fl(S_out)),  (m, e) = batch_frexp(r),  ek = e' | k << 8  with
    q = round_half_even(((v << k) * m) / 2^e').  Normalised for the kernels' contract
    0 <= m < 2^31, 1 <= e' <= 62 without changing any rounded result:
      m == 2^31 (mantissa rounded up to 1.0)  ->  (2^30, e-1)   same rational m/2^e
      e  > 62                                  ->  (0, 33)       |v*m/2^e| < 1/2 -> 0 either way
      lift: e < 33  ->  k = 33 - e, e' = 33   same rational (v*2^k*m)/2^(e+k); lets the conv
            epilogues use their high-word fast path.  Needs |v| < 2^vbits with vbits + k <= 31
            (``vbits`` scalar or per-channel array: an upper bound on the bit length of |v|).
    """
    a = pre_act_scaling_factor.detach().reshape(-1).cpu().double()
    w = pre_weight_scaling_factor.detach().reshape(-1).cpu().double()
    out = z_scaling_factor.detach().reshape(-1).cpu().float().double()
    r = (a * w).float().double() / out
    m, e = batch_frexp(r)
    m = m.numpy().copy()
    e = e.numpy().astype(np.int64)
    top = m == (1 << 31)
    m[top] = 1 << 30
    e[top] -= 1
    tiny = e > 62
    m[tiny], e[tiny] = 0, 33
    k = np.zeros_like(e)
    if lift:
        k = np.maximum(33 - e, 0)
        e = e + k
    if (e < 1).any():
        raise ValueError("requantisation ratio >= 2^30 is not supported by the integer kernels")
    if (np.broadcast_to(np.asarray(vbits, np.int64), e.shape) + k > 31).any():
        raise ValueError("requantisation ratio too large for the accumulator width "
                         f"(needs pre-shift {int(k.max())} on values of up to {np.max(vbits)} bits)")
    return m.astype(np.int32), (e | (k << 8)).astype(np.int32)


def tables_fit_fast(m, ek, vbits, allow_shift=True) -> bool:
    """The structural half of the fast contract: e in [33, 62] and |v << k| < 2^31 for |v| < 2^vbits.  Tables that
    fit but are not provably tie-free (``tables_are_fast``) run the fast kernels in their exact-tie mode
    (``fast_tables`` bit 2)."""
    ek = np.asarray(ek, np.int64).reshape(-1)
    e, k = ek & 0xff, ek >> 8
    vb = np.broadcast_to(np.asarray(vbits, np.int64), ek.shape)
    if ((e < 33) | (e > 62)).any():
        return False
    return not (((k != 0).any() and not allow_shift) or (vb + k > 31).any())


def tables_are_fast(m, ek, vbits, allow_shift=True) -> bool:
    """True if a (m, ek) table satisfies the conv kernels' fast contract for inputs |v| < 2^vbits:
    e in [33, 62], vbits + k <= 31, and no exact rounding tie is possible.  A tie needs
    2^(e-1) | (v << k) * m, i.e. tz(v) >= e - 1 - k - tz(m); that exceeds every non-zero
    |v| < 2^vbits iff tz(m) <= e - 1 - k - vbits  (m == 0 gives 0 on both paths)."""
    m = np.asarray(m, np.int64).reshape(-1)
    ek = np.asarray(ek, np.int64).reshape(-1)
    e, k = ek & 0xff, ek >> 8
    vb = np.broadcast_to(np.asarray(vbits, np.int64), m.shape)
    if ((e < 33) | (e > 62)).any():
        return False
    if ((k != 0).any() and not allow_shift) or (vb + k > 31).any():
        return False
    tz = np.array([((int(x) & -int(x)).bit_length() - 1) if x else 0 for x in m], np.int64)
    return bool(((m == 0) | (tz <= e - 1 - k - vb)).all())


def input_quant_lut(inv_scale: float, mean, std, lo: int = -128, hi: int = 127) -> torch.Tensor:
    """int8 [3][256]: lut[c][u] = clamp(rint(fl(1/S) * Normalize_c(ToTensor(u)))) with the reference pipeline's own
    float32 operations on the host - torchvision ToTensor ``u.float().div(255)``, Normalize ``sub(mean).div(std)``
    (quant_train.py:432-440), then the input QuantAct (quant_utils.py:73-97).  A table look-up of a uint8 pixel is
    therefore bit-identical to quantising the normalised fp32 tensor."""
    u = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    mean32, std32 = torch.as_tensor(mean, dtype=torch.float32), torch.as_tensor(std, dtype=torch.float32)
    v = (u.view(1, 256) - mean32.view(3, 1)) / std32.view(3, 1)
    inv = torch.tensor(float(inv_scale), dtype=torch.float32)
    return torch.round(inv * v).clamp(lo, hi).to(torch.int8).contiguous()


# --------------------------------------------------------------------- quantise-from-float
def linear_quantize(input, scale, zero_point, inplace=False):
    """round(1/scale * x + zp) with the reference's broadcasting rules (quant_utils.py:73-97).
    Host tensors only (parameter preparation); activations go through the HIP kernels."""
    if input.dim() == 4:
        scale = scale.view(-1, 1, 1, 1)
        zero_point = zero_point.view(-1, 1, 1, 1)
    elif input.dim() == 2:
        scale = scale.view(-1, 1)
        zero_point = zero_point.view(-1, 1)
    else:
        scale = scale.view(-1)
        zero_point = zero_point.view(-1)
    if inplace:
        return input.mul_(1. / scale).add_(zero_point).round_()
    return torch.round(1. / scale * input + zero_point)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _device_quantize(x, scale, lo, hi):
    if scale.numel() != 1:
        raise ValueError("device-side quantisation supports per-tensor scales only")
    s = scale.detach().reshape(-1)[:1].float().cpu()
    inv = float((1. / s).item())  # binary32 reciprocal, as `1. / scale`
    x = x.contiguous().float()
    y = torch.empty_like(x)
    _lib.call("hawq_fakequant_f32", x.data_ptr(), y.data_ptr(), x.numel(), inv, 1.0, int(lo), int(hi), _stream())
    return y


class SymmetricQuantFunction:
    """clamp(round(x/scale), -2^(k-1), 2^(k-1)-1)  (quant_utils.py:231-258).  ``apply(x, k, scale)``."""

    @staticmethod
    def apply(x, k, specified_scale=None):
        if specified_scale is None:
            raise ValueError("The SymmetricQuantFunction requires a pre-calculated scaling factor")
        n = 2 ** (k - 1) - 1
        if x.is_cuda:
            return _device_quantize(x, specified_scale, -n - 1, n)
        with torch.no_grad():
            q = linear_quantize(x, specified_scale, torch.zeros(1), inplace=False)
            return torch.clamp(q, -n - 1, n)


class AsymmetricQuantFunction:
    """clamp(round(x/scale + zp), 0, 2^k-1)  (quant_utils.py:275-308); zp defaults to 0."""

    @staticmethod
    def apply(x, k, specified_scale=None, specified_zero_point=None):
        if specified_scale is None:
            raise ValueError("The AsymmetricQuantFunction requires a pre-calculated scaling factor")
        n = 2 ** k - 1
        if x.is_cuda and specified_zero_point is None:
            return _device_quantize(x, specified_scale, 0, n)
        with torch.no_grad():
            zp = specified_zero_point if specified_zero_point is not None else torch.zeros(1)
            return torch.clamp(linear_quantize(x, specified_scale, zp, inplace=False), 0, n)


def _dev_i32(arr, device):
    return torch.from_numpy(np.ascontiguousarray(arr, np.int32)).to(device)


class fixedpoint_fn:
    """Integer requantisation of a fp32 (integer*scale) tensor (quant_utils.py:344-456).

    ``apply`` keeps the reference signature and returns the INTEGER-valued fp32 tensor (the
    caller multiplies by the output scale, quant_modules.py:302-303).  Work happens in
    hawq_fixedpoint_f32; (m, e) tables are formed on the host without a device round trip per
    channel."""

    @staticmethod
    def apply(z, bitwidth, quant_mode, z_scaling_factor, case, pre_act_scaling_factor=None,
              pre_weight_scaling_factor=None, identity=None, identity_scaling_factor=None,
              identity_weight_scaling_factor=None):
        if not z.is_cuda:
            raise RuntimeError("hawq_amd.fixedpoint_fn runs on the MI355X only (no CPU path)")
        n = 2 ** (bitwidth - 1) - 1 if quant_mode == 'symmetric' else 2 ** bitwidth - 1
        lo, hi = (-n - 1, n) if quant_mode == 'symmetric' else (0, n)
        z = z.contiguous().float()
        N, Cc = z.shape[0], z.shape[1]
        HW = z.numel() // (N * Cc)
        dev = z.device
        s_a = float(pre_act_scaling_factor.detach().reshape(-1)[0].item())
        s_w = pre_weight_scaling_factor.detach().reshape(-1).float()
        m, e = requant_table(pre_act_scaling_factor, pre_weight_scaling_factor, z_scaling_factor, lift=False)
        md, ed, swd = _dev_i32(m, dev), _dev_i32(e, dev), s_w.to(dev).contiguous()
        y = torch.empty_like(z)
        if case == 0:
            _lib.call("hawq_fixedpoint_f32", z.data_ptr(), y.data_ptr(), N, Cc, HW, s_a, swd.data_ptr(),
                      md.data_ptr(), ed.data_ptr(), int(m.size), None, 0.0, None, None, None, 1, 1.0, 1, lo, hi,
                      _stream())
        else:
            ident = identity.contiguous().float()
            s_ida = float(identity_scaling_factor.detach().reshape(-1)[0].item())
            s_idw = identity_weight_scaling_factor.detach().reshape(-1).float()
            m1, e1 = requant_table(identity_scaling_factor, identity_weight_scaling_factor, z_scaling_factor,
                                   lift=False)
            m1d, e1d, sidwd = _dev_i32(m1, dev), _dev_i32(e1, dev), s_idw.to(dev).contiguous()
            _lib.call("hawq_fixedpoint_f32", z.data_ptr(), y.data_ptr(), N, Cc, HW, s_a, swd.data_ptr(),
                      md.data_ptr(), ed.data_ptr(), int(m.size), ident.data_ptr(), s_ida, sidwd.data_ptr(),
                      m1d.data_ptr(), e1d.data_ptr(), int(m1.size), 1.0, 0, 0, 0, _stream())
        return y


# --------------------------------------------------------------------- parameter preparation
def ieee_sqrt(x: torch.Tensor) -> torch.Tensor:
    """Correctly-rounded binary32 sqrt.  torch's CPU sqrt goes through MKL VML and is not
    always correctly rounded, which would make prepared weights host-dependent; sqrt in
    binary64 followed by one rounding is exact for binary32 inputs ("sqrt quirk")."""
    return torch.sqrt(x.double()).float()


def fold_bn(conv_weight, bn_weight, bn_bias, running_mean, running_var, eps, conv_bias=None):
    """BN folding of QuantBnConv2d's frozen branch (quant_modules.py:441-449) on host tensors."""
    with torch.no_grad():
        running_std = ieee_sqrt(running_var.detach().float().cpu() + eps)
        scale_factor = bn_weight.detach().float().cpu() / running_std
        scaled_weight = conv_weight.detach().float().cpu() * scale_factor.reshape([-1, 1, 1, 1])
        base = conv_bias.detach().float().cpu() if conv_bias is not None else torch.zeros_like(running_std)
        scaled_bias = (base - running_mean.detach().float().cpu()) * scale_factor + bn_bias.detach().float().cpu()
        return scaled_weight, scaled_bias


def quantize_weight_per_channel(w, weight_bit, per_channel=True, weight_percentile=0):
    """(weight_integer fp32-valued, scale[Cout]) as quant_modules.py:452-480 / 97-115; with ``weight_percentile`` the
    range is the reference's percentile range (quant_modules.py:458-474: per-channel torch.kthvalue with ceil'ed
    indices, or get_percentile_min_max on the whole tensor).  Host tensors: parameter preparation."""
    import math
    with torch.no_grad():
        w = w.detach().float().cpu()
        flat = w.contiguous().view(w.shape[0], -1)
        if per_channel:
            if weight_percentile == 0:
                w_min, w_max = flat.min(dim=1).values, flat.max(dim=1).values
            else:
                n = flat.shape[1]
                lower_index = math.ceil(n * (100 - weight_percentile) * 0.01)
                upper_index = math.ceil(n * weight_percentile * 0.01)
                w_min = torch.kthvalue(flat, k=lower_index, dim=1).values
                w_max = torch.kthvalue(flat, k=upper_index, dim=1).values
        elif weight_percentile == 0:
            w_min, w_max = flat.min().expand(1), flat.max().expand(1)
        else:
            w_min, w_max = get_percentile_min_max(w.reshape(-1), 100 - weight_percentile, weight_percentile, output_tensor=True)
            w_min, w_max = w_min.expand(1), w_max.expand(1)
        scale = symmetric_linear_quantization_params(weight_bit, w_min, w_max, per_channel)
        if not per_channel:
            scale = scale.reshape(1)
        return SymmetricQuantFunction.apply(w, weight_bit, scale), scale


def quantize_bias(bias, weight_scale, pre_act_scaling_factor, bias_bit=32):
    """bias_integer at scale fl(S_w[c]*S_a) (quant_modules.py:482-484 / 117-118)."""
    with torch.no_grad():
        bias_scale = weight_scale.view(1, -1).cpu() * pre_act_scaling_factor.detach().view(1, -1).float().cpu()
        return SymmetricQuantFunction.apply(bias.detach().float().cpu(), bias_bit, bias_scale), bias_scale
