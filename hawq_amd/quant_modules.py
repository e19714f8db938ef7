"""Quantised operator modules with HAWQ's Python API, executing on MI355X HIP kernels.

Same class names, constructor arguments, ``set_param`` / ``fix`` / ``unfix``, buffers (checkpoint
keys) and ``(tensor, scale)`` tuple convention as the reference's
utils/quantization_utils/quant_modules.py, so q_resnet graphs, ``load_state_dict`` and the
``setattr`` bit-config loop of quant_train.py:264-299 work unchanged.  Differences by design:

* ``forward`` of a frozen/eval module dispatches through the C ABI (include/hawq_mi355.h) into
  hand-written gfx950 kernels; there is no CPU or eager-PyTorch arithmetic path - CPU tensors
  raise.  (Whole-network inference should use hawq_amd.engine.IntegerEngine, which fuses these
  modules' arithmetic into ~3 launches per residual unit; the per-module path here exists for
  API compatibility, range calibration and per-module parity tests.)
* parameter-only work the reference redoes every forward (BN folding, weight/bias
  quantisation, quant_modules.py:441-484) is done once on the host and cached.
* training branches (unfolded BN statistics :417-438, percentile ranges :237-245/:458-474) are
  QAT-only and not implemented: they raise NotImplementedError.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import Module, Parameter

from . import _lib
from . import packing
from .quant_utils import (AsymmetricQuantFunction, SymmetricQuantFunction, asymmetric_linear_quantization_params,
                          device_min_max, get_percentile_min_max,
                          fixedpoint_fn, fold_bn, quantize_bias, quantize_weight_per_channel, requant_table,
                          symmetric_linear_quantization_params)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_device(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"hawq_amd.{who}: tensors must live on the MI355X (no CPU fallback exists)")


def _pad64(c):
    return (c + 63) // 64 * 64


class QuantAct(Module):
    """Activation quantiser (reference: quant_modules.py:133-305)."""

    def __init__(self, activation_bit=4, act_range_momentum=0.95, full_precision_flag=False, running_stat=True,
                 quant_mode="symmetric", fix_flag=False, act_percentile=0, fixed_point_quantization=False):
        super().__init__()
        self.activation_bit = activation_bit
        self.act_range_momentum = act_range_momentum
        self.full_precision_flag = full_precision_flag
        self.running_stat = running_stat
        self.quant_mode = quant_mode
        self.fix_flag = fix_flag
        self.act_percentile = act_percentile
        self.fixed_point_quantization = fixed_point_quantization
        self.register_buffer('x_min', torch.zeros(1))
        self.register_buffer('x_max', torch.zeros(1))
        self.register_buffer('act_scaling_factor', torch.zeros(1))
        self.register_buffer('pre_weight_scaling_factor', torch.ones(1))
        self.register_buffer('identity_weight_scaling_factor', torch.ones(1))

    def __repr__(self):
        return "{0}(activation_bit={1}, full_precision_flag={2}, quant_mode={3}, Act_min: {4:.2f}, Act_max: {5:.2f})" \
            .format(self.__class__.__name__, self.activation_bit, self.full_precision_flag, self.quant_mode,
                    self.x_min.item(), self.x_max.item())

    def fix(self):
        self.running_stat = False
        self.fix_flag = True

    def unfix(self):
        self.running_stat = True
        self.fix_flag = False

    def compute_scale(self):
        """act_scaling_factor from the frozen range (quant_modules.py:262-270).  Evaluated on the
        host in IEEE binary32: torch's GPU kernels turn `tensor / python_scalar` into a multiply by
        the rounded reciprocal, which would make the scale (and everything downstream) differ
        from the CPU reference by an ulp."""
        dev = self.x_min.device
        x_min, x_max = self.x_min.detach().float().cpu(), self.x_max.detach().float().cpu()
        if self.quant_mode == 'symmetric':
            self.act_scaling_factor = symmetric_linear_quantization_params(self.activation_bit, x_min, x_max,
                                                                           False).to(dev)
        elif self.quant_mode == 'asymmetric':
            scale, zp = asymmetric_linear_quantization_params(self.activation_bit, x_min, x_max, True)
            self.act_scaling_factor, self.act_zero_point = scale.to(dev), zp.to(dev)
        else:
            raise ValueError("unknown quant mode: {}".format(self.quant_mode))
        return self.act_scaling_factor

    def forward(self, x, pre_act_scaling_factor=None, pre_weight_scaling_factor=None, identity=None,
                identity_scaling_factor=None, identity_weight_scaling_factor=None):
        if type(x) is tuple:
            if len(x) == 3:
                raise NotImplementedError("multi-branch (Inception) QuantAct is outside the ResNet hot path")
            pre_act_scaling_factor = x[1]
            x = x[0]
        if self.quant_mode == "symmetric":
            self.act_function = SymmetricQuantFunction.apply
        elif self.quant_mode == "asymmetric":
            self.act_function = AsymmetricQuantFunction.apply
        else:
            raise ValueError("unknown quant mode: {}".format(self.quant_mode))
        _require_device(x, "QuantAct")

        if self.running_stat:  # range calibration / QAT range tracking (quant_modules.py:233-258) on the device
            if self.act_percentile == 0:
                x_min, x_max = device_min_max(x.data)
            elif self.quant_mode == 'symmetric':
                x_min, x_max = get_percentile_min_max(x.detach().view(-1), 100 - self.act_percentile, self.act_percentile,
                                                      output_tensor=True)
            else:  # 'asymmetric' (post-ReLU, unsigned, no zero point): the lower bound stays 0 (quant_modules.py:241-245)
                x_min, x_max = get_percentile_min_max(x.detach().view(-1), 0, self.act_percentile, output_tensor=True)
            if self.x_min == self.x_max:
                self.x_min += x_min
                self.x_max += x_max
            elif self.act_range_momentum == -1:
                self.x_min = torch.minimum(self.x_min, x_min)
                self.x_max = torch.maximum(self.x_max, x_max)
            else:
                self.x_min = self.x_min * self.act_range_momentum + x_min * (1 - self.act_range_momentum)
                self.x_max = self.x_max * self.act_range_momentum + x_max * (1 - self.act_range_momentum)

        if self.full_precision_flag:
            return x
        if not (getattr(self, "use_integer_buffers", False) and not self.running_stat):
            self.compute_scale()   # else: act_scaling_factor as loaded from a quantized checkpoint (the file carries no x_min / x_max)
        if (pre_act_scaling_factor is None) or (self.fixed_point_quantization is True):
            quant_act_int = self.act_function(x, self.activation_bit, self.act_scaling_factor)
        elif type(pre_act_scaling_factor) is list:
            raise NotImplementedError("multi-branch (Inception) QuantAct is outside the ResNet hot path")
        elif identity is None:
            if pre_weight_scaling_factor is None:
                pre_weight_scaling_factor = self.pre_weight_scaling_factor
            quant_act_int = fixedpoint_fn.apply(x, self.activation_bit, self.quant_mode, self.act_scaling_factor, 0,
                                                pre_act_scaling_factor, pre_weight_scaling_factor)
        else:
            if identity_weight_scaling_factor is None:
                identity_weight_scaling_factor = self.identity_weight_scaling_factor
            quant_act_int = fixedpoint_fn.apply(x, self.activation_bit, self.quant_mode, self.act_scaling_factor, 1,
                                                pre_act_scaling_factor, pre_weight_scaling_factor, identity,
                                                identity_scaling_factor, identity_weight_scaling_factor)
        correct_output_scale = self.act_scaling_factor.view(-1)
        return (quant_act_int * correct_output_scale, self.act_scaling_factor)


class _IntConvMixin:
    """Shared device plumbing: fp32 NCHW (int*scale) -> integer conv kernel -> fp32 NCHW."""

    def _run_int_conv(self, x, pre_act_scaling_factor, weight_integer, bias_integer, bias_scale, in_bits, stride,
                      padding, groups=1):
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight_integer.shape
        dev = x.device
        if groups != 1:
            return self._run_grouped_conv(x, pre_act_scaling_factor, weight_integer, bias_integer, bias_scale, stride, padding,
                                          groups)
        key = (weight_integer.data_ptr(), weight_integer._version, bias_integer.data_ptr(), bias_integer._version,
               str(dev), in_bits)
        if getattr(self, "_dev_key", None) != key:
            w_bits = 4 if (self.weight_bit <= 4 and in_bits == 4) else 8
            cin_p, cout_p = _pad64(Cin), _pad64(Cout)
            wp = packing.pack_conv_weight(weight_integer.detach().cpu().numpy(), w_bits, cin_p, cout_p)
            b = np.zeros(cout_p, np.int32)
            b[:Cout] = bias_integer.detach().cpu().numpy().astype(np.int64).clip(-2 ** 31, 2 ** 31 - 1)
            fs = np.zeros(cout_p, np.float32)
            fs[:Cout] = bias_scale.detach().reshape(-1).cpu().numpy()
            self._dev_w = torch.from_numpy(wp).to(dev)
            self._dev_b = torch.from_numpy(b).to(dev)
            self._dev_fs = torch.from_numpy(fs).to(dev)
            self._dev_wbits, self._dev_cin_p, self._dev_cout_p = w_bits, cin_p, cout_p
            self._dev_key = key
        cin_p, cout_p = self._dev_cin_p, self._dev_cout_p
        x = x.contiguous().float()
        s_a = float(pre_act_scaling_factor.detach().reshape(-1)[0].item())
        xq = torch.empty(N * H * W * cin_p * in_bits // 8, dtype=torch.uint8, device=dev)
        _lib.call("hawq_f32_nchw_to_q_nhwc", x.data_ptr(), xq.data_ptr(), N, Cin, H, W, cin_p, in_bits, s_a, _stream())
        Ho = (H + 2 * padding - KH) // stride + 1
        Wo = (W + 2 * padding - KW) // stride + 1
        acc = torch.empty(N * Ho * Wo * cout_p, dtype=torch.int32, device=dev)
        a = _lib.ConvArgs()
        a.in_, a.wgt, a.bias = xq.data_ptr(), self._dev_w.data_ptr(), self._dev_b.data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, cin_p, cout_p, KH, KW, stride, padding
        a.in_bits, a.w_bits = in_bits, self._dev_wbits
        a.epilogue, a.out_acc = _lib.EPI_RAW, acc.data_ptr()
        _lib.call("hawq_conv2d", a, _stream())
        y = torch.empty(N, Cout, Ho, Wo, dtype=torch.float32, device=dev)
        _lib.call("hawq_acc_nhwc_to_f32_nchw", acc.data_ptr(), y.data_ptr(), N, Cout, Ho, Wo, cout_p,
                  self._dev_fs.data_ptr(), _stream())
        self.last_accumulators = (acc, (N, Ho, Wo, cout_p))  # int32 NHWC, for parity checks
        return y


    def _run_grouped_conv(self, x, pre_act_scaling_factor, weight_integer, bias_integer, bias_scale, stride, padding, groups):
        """Grouped / depthwise layers (MobileNetV2): int8 NHWC activations, hawq_conv2d_grouped, exact int32 accumulators."""
        N, Cin, H, W = x.shape
        Cout, Cg, KH, KW = weight_integer.shape
        dev = x.device
        key = ("g", weight_integer.data_ptr(), weight_integer._version, bias_integer.data_ptr(), bias_integer._version, str(dev))
        # depthwise 3x3 (MobileNetV2's conv2): the vectorised kernel with tap-major weights [3][3][C]
        depthwise = groups == Cin == Cout and KH == KW == 3 and padding == 1 and stride in (1, 2) and Cin % 4 == 0
        if getattr(self, "_dev_key", None) != key:
            w = weight_integer.detach().cpu().numpy().astype(np.int8)
            w = w.reshape(Cout, 9).T if depthwise else w.transpose(0, 2, 3, 1)   # [3][3][C]  /  [Cout][KH][KW][Cin / groups]
            self._dev_w = torch.from_numpy(np.ascontiguousarray(w)).to(dev)
            b = bias_integer.detach().cpu().numpy().astype(np.int64).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
            self._dev_b = torch.from_numpy(b).to(dev)
            self._dev_fs = bias_scale.detach().reshape(-1).float().to(dev).contiguous()
            self._dev_key = key
        x = x.contiguous().float()
        s_a = float(pre_act_scaling_factor.detach().reshape(-1)[0].item())
        xq = torch.empty(N * H * W * Cin, dtype=torch.uint8, device=dev)
        _lib.call("hawq_f32_nchw_to_q_nhwc", x.data_ptr(), xq.data_ptr(), N, Cin, H, W, Cin, 8, s_a, _stream())
        Ho = (H + 2 * padding - KH) // stride + 1
        Wo = (W + 2 * padding - KW) // stride + 1
        acc = torch.empty(N * Ho * Wo * Cout, dtype=torch.int32, device=dev)
        if depthwise:
            _lib.call("hawq_depthwise3x3", xq.data_ptr(), self._dev_w.data_ptr(), self._dev_b.data_ptr(), N, H, W, Cin, stride, acc.data_ptr(), _stream())
        else:
            _lib.call("hawq_conv2d_grouped", xq.data_ptr(), self._dev_w.data_ptr(), self._dev_b.data_ptr(), N, H, W, Cin, Cout, KH, KW,
                      stride, padding, groups, acc.data_ptr(), _stream())
        y = torch.empty(N, Cout, Ho, Wo, dtype=torch.float32, device=dev)
        _lib.call("hawq_acc_nhwc_to_f32_nchw", acc.data_ptr(), y.data_ptr(), N, Cout, Ho, Wo, Cout, self._dev_fs.data_ptr(), _stream())
        return y

class QuantBnConv2d(_IntConvMixin, Module):
    """Conv with folded BatchNorm (reference: quant_modules.py:308-494); frozen/folded branch."""

    def __init__(self, weight_bit=4, bias_bit=None, full_precision_flag=False, quant_mode="symmetric",
                 per_channel=False, fix_flag=False, weight_percentile=0, fix_BN=False, fix_BN_threshold=None):
        super().__init__()
        self.weight_bit = weight_bit
        self.full_precision_flag = full_precision_flag
        self.per_channel = per_channel
        self.fix_flag = fix_flag
        self.weight_percentile = weight_percentile
        self.bias_bit = bias_bit
        self.quantize_bias = False if bias_bit is None else True
        self.quant_mode = quant_mode
        self.fix_BN = fix_BN
        self.training_BN_mode = fix_BN
        self.fix_BN_threshold = fix_BN_threshold
        self.counter = 1

    def set_param(self, conv, bn):
        self.out_channels = conv.out_channels
        self.register_buffer('convbn_scaling_factor', torch.zeros(self.out_channels))
        self.register_buffer('weight_integer', torch.zeros_like(conv.weight.data))
        self.register_buffer('bias_integer', torch.zeros_like(bn.bias))
        self.conv = conv
        self.bn = bn
        self.bn.momentum = 0.99

    def __repr__(self):
        conv_s = super().__repr__()
        return "({0}, weight_bit={1}, bias_bit={2}, groups={3}, wt-channel-wise={4}, wt-percentile={5}, " \
               "quant_mode={6})".format(conv_s, self.weight_bit, self.bias_bit, self.conv.groups, self.per_channel,
                                        self.weight_percentile, self.quant_mode)

    def fix(self):
        self.fix_flag = True
        self.fix_BN = True

    def unfix(self):
        self.fix_flag = False
        self.fix_BN = self.training_BN_mode

    def prepare(self, pre_act_scaling_factor):
        """Host-side, cached: BN fold + per-channel weight/bias quantisation
        (quant_modules.py:441-484).  Fills the same buffers the reference fills."""
        c, b = self.conv, self.bn
        s_a = pre_act_scaling_factor.detach().reshape(-1).float().cpu()
        key = (c.weight._version, c.weight.data_ptr(), b.weight._version, b.bias._version,
               b.running_mean._version, b.running_var._version, self.weight_bit, self.bias_bit, self.per_channel,
               self.weight_percentile, float(s_a[0]))
        if getattr(self, "_prep_key", None) == key:
            return self._prep_bias_scale
        if getattr(self, "use_integer_buffers", False):
            # integer weights / biases / scales as loaded (quantized_checkpoint.pth.tar, quant_train.py:665-670): nothing is
            # re-derived from the float parameters
            bias_scale = self.convbn_scaling_factor.detach().float().cpu().view(1, -1) * s_a.view(1, -1)
            self._prep_key, self._prep_bias_scale = key, bias_scale.to(c.weight.device)
            return self._prep_bias_scale
        if self.quant_mode != 'symmetric':
            raise Exception('For weight, we only support symmetric quantization.')
        dev = c.weight.device
        w_f, b_f = fold_bn(c.weight, b.weight, b.bias, b.running_mean, b.running_var, b.eps, c.bias)
        w_int, s_w = quantize_weight_per_channel(w_f, self.weight_bit, self.per_channel, self.weight_percentile)
        self.convbn_scaling_factor = s_w.to(dev)
        self.weight_integer = w_int.to(dev)
        if self.quantize_bias:
            b_int, bias_scale = quantize_bias(b_f, s_w, s_a, self.bias_bit)
            self.bias_integer = b_int.to(dev)
        else:
            raise NotImplementedError("un-quantised bias is not an integer-only path")
        self.convbn_scaled_bias = b_f.to(dev)
        self._prep_key = key
        self._prep_bias_scale = bias_scale.to(dev)
        return self._prep_bias_scale

    def forward(self, x, pre_act_scaling_factor=None):
        if type(x) is tuple:
            pre_act_scaling_factor = x[1]
            x = x[0]
        if self.quant_mode not in ("symmetric", "asymmetric"):
            raise ValueError("unknown quant mode: {}".format(self.quant_mode))
        if self.fix_flag is False:
            self.counter += 1
            if (self.fix_BN_threshold is None) or (self.counter < self.fix_BN_threshold):
                self.fix_BN = self.training_BN_mode
            else:
                self.fix_BN = True
        if self.fix_BN is False:
            raise NotImplementedError("unfolded-BN training forward (quant_modules.py:417-438) is QAT-only")
        if self.full_precision_flag:
            raise NotImplementedError("full_precision_flag bypasses the integer path (out of scope)")
        if self.conv.dilation[0] != 1:
            raise NotImplementedError("dilated convolutions are outside the path")
        _require_device(x, "QuantBnConv2d")
        bias_scale = self.prepare(pre_act_scaling_factor)
        in_bits = getattr(self, "input_bit", 8)
        y = self._run_int_conv(x, pre_act_scaling_factor, self.weight_integer, self.bias_integer, bias_scale, in_bits,
                               self.conv.stride[0], self.conv.padding[0], self.conv.groups)
        return (y, self.convbn_scaling_factor)


class QuantConv2d(_IntConvMixin, Module):
    """Conv without BN (reference: quant_modules.py:605-736; used by MobileNetV2 there)."""

    def __init__(self, weight_bit=4, bias_bit=None, full_precision_flag=False, quant_mode="symmetric",
                 per_channel=False, fix_flag=False, weight_percentile=0):
        super().__init__()
        self.full_precision_flag = full_precision_flag
        self.weight_bit = weight_bit
        self.quant_mode = quant_mode
        self.per_channel = per_channel
        self.fix_flag = fix_flag
        self.weight_percentile = weight_percentile
        self.bias_bit = bias_bit
        self.quantize_bias = (False if bias_bit is None else True)

    def __repr__(self):
        s = super().__repr__()
        return "(" + s + " weight_bit={}, full_precision_flag={}, quant_mode={})".format(
            self.weight_bit, self.full_precision_flag, self.quant_mode)

    def set_param(self, conv):
        self.in_channels = conv.in_channels
        self.out_channels = conv.out_channels
        self.kernel_size = conv.kernel_size
        self.stride = conv.stride
        self.padding = conv.padding
        self.dilation = conv.dilation
        self.groups = conv.groups
        self.conv = conv
        self.register_buffer('conv_scaling_factor', torch.zeros(self.out_channels))
        self.weight = Parameter(conv.weight.data.clone())
        self.register_buffer('weight_integer', torch.zeros_like(self.weight, dtype=torch.int8))
        try:
            self.bias = Parameter(conv.bias.data.clone())
        except AttributeError:
            self.bias = None

    def fix(self):
        self.fix_flag = True

    def unfix(self):
        self.fix_flag = False

    def forward(self, x, pre_act_scaling_factor=None):
        if type(x) is tuple:
            pre_act_scaling_factor = x[1]
            x = x[0]
        if self.quant_mode == "asymmetric" or self.quant_mode not in ("symmetric",):
            if self.quant_mode != "asymmetric":
                raise ValueError("unknown quant mode: {}".format(self.quant_mode))
            raise Exception('For weight, we only support symmetric quantization.')
        if self.dilation[0] != 1:
            raise NotImplementedError("dilated convolutions are outside the path")
        _require_device(x, "QuantConv2d")
        dev = self.weight.device
        if getattr(self, "use_integer_buffers", False):   # integer weights / scales as loaded, see QuantBnConv2d.prepare
            w_int, s_w = self.weight_integer.detach().float().cpu(), self.conv_scaling_factor.detach().float().cpu()
        else:
            w_int, s_w = quantize_weight_per_channel(self.weight, self.weight_bit, self.per_channel, self.weight_percentile)
            self.conv_scaling_factor = s_w.to(dev)
            self.weight_integer = w_int.to(dev)
        s_a = pre_act_scaling_factor.detach().reshape(-1).float().cpu()
        if self.quantize_bias and (self.bias is not None):
            b_int, bias_scale = quantize_bias(self.bias, s_w, s_a, self.bias_bit)
            self.bias_integer = b_int.to(dev)
        else:
            bias_scale = s_w.view(1, -1) * s_a.view(1, -1)
            self.bias_integer = None
            b_int = torch.zeros(self.out_channels)
        y = self._run_int_conv(x, pre_act_scaling_factor, self.weight_integer.float(), b_int.to(dev),
                               bias_scale.to(dev), getattr(self, "input_bit", 8), self.stride[0], self.padding[0], self.groups)
        return (y, self.conv_scaling_factor)


class QuantLinear(_IntConvMixin, Module):
    """Fully-connected layer (reference: quant_modules.py:12-130)."""

    def __init__(self, weight_bit=4, bias_bit=None, full_precision_flag=False, quant_mode='symmetric',
                 per_channel=False, fix_flag=False, weight_percentile=0):
        super().__init__()
        self.full_precision_flag = full_precision_flag
        self.weight_bit = weight_bit
        self.quant_mode = quant_mode
        self.per_channel = per_channel
        self.fix_flag = fix_flag
        self.weight_percentile = weight_percentile
        self.bias_bit = bias_bit
        self.quantize_bias = (False if bias_bit is None else True)
        self.counter = 0

    def __repr__(self):
        s = super().__repr__()
        return "(" + s + " weight_bit={}, full_precision_flag={}, quantize_fn={})".format(
            self.weight_bit, self.full_precision_flag, self.quant_mode)

    def set_param(self, linear):
        self.in_features = linear.in_features
        self.out_features = linear.out_features
        self.register_buffer('fc_scaling_factor', torch.zeros(self.out_features))
        self.weight = Parameter(linear.weight.data.clone())
        self.register_buffer('weight_integer', torch.zeros_like(self.weight))
        self.register_buffer('bias_integer', torch.zeros_like(linear.bias))
        try:
            self.bias = Parameter(linear.bias.data.clone())
        except AttributeError:
            self.bias = None

    def fix(self):
        self.fix_flag = True

    def unfix(self):
        self.fix_flag = False

    def prepare(self, prev_act_scaling_factor):
        """Host-side, cached weight/bias quantisation (quant_modules.py:94-118)."""
        if self.quant_mode == "asymmetric":
            raise Exception('For weight, we only support symmetric quantization.')
        if self.quant_mode != "symmetric":
            raise ValueError("unknown quant mode: {}".format(self.quant_mode))
        s_a = prev_act_scaling_factor.detach().reshape(-1).float().cpu()
        key = (self.weight._version, self.weight.data_ptr(), self.bias._version, self.weight_bit, self.bias_bit,
               self.per_channel, float(s_a[0]))
        if getattr(self, "_prep_key", None) == key:
            return self._prep_bias_scale
        dev = self.weight.device
        if getattr(self, "use_integer_buffers", False):   # integer weights / bias / scale as loaded, see QuantBnConv2d.prepare
            bias_scale = self.fc_scaling_factor.detach().float().cpu().view(1, -1) * s_a.view(1, -1)
            self._prep_key, self._prep_bias_scale = key, bias_scale.to(dev)
            return self._prep_bias_scale
        w_int, s_w = quantize_weight_per_channel(self.weight, self.weight_bit, self.per_channel)
        self.fc_scaling_factor = s_w.to(dev)
        self.weight_integer = w_int.to(dev)
        b_int, bias_scale = quantize_bias(self.bias, s_w, s_a, self.bias_bit)
        self.bias_integer = b_int.to(dev)
        self._prep_key = key
        self._prep_bias_scale = bias_scale.to(dev)
        return self._prep_bias_scale

    def forward(self, x, prev_act_scaling_factor=None):
        if type(x) is tuple:
            prev_act_scaling_factor = x[1]
            x = x[0]
        if self.full_precision_flag:
            raise NotImplementedError("full_precision_flag bypasses the integer path (out of scope)")
        _require_device(x, "QuantLinear")
        bias_scale = self.prepare(prev_act_scaling_factor)
        B, K = x.shape
        y = self._run_int_conv(x.reshape(B, K, 1, 1), prev_act_scaling_factor,
                               self.weight_integer.reshape(self.out_features, K, 1, 1), self.bias_integer, bias_scale,
                               8, 1, 0)
        return y.reshape(B, self.out_features)


class QuantAveragePool2d(Module):
    """Integer average pooling (reference: quant_modules.py:557-602)."""

    def __init__(self, kernel_size=7, stride=1, padding=0):
        super().__init__()
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.final_pool = nn.AvgPool2d(kernel_size=kernel_size, stride=stride, padding=padding)

    def set_param(self, pool):
        self.final_pool = pool

    def forward(self, x, x_scaling_factor=None):
        if type(x) is tuple:
            x_scaling_factor = x[1]
            x = x[0]
        if x_scaling_factor is None:
            return self.final_pool(x)
        _require_device(x, "QuantAveragePool2d")
        N, Cc, H, W = x.shape
        if not (H == W == self.kernel_size and self.padding == 0):
            raise NotImplementedError("only the global (kernel == feature map) average pool of the ResNets")
        x_scaling_factor = x_scaling_factor.view(-1)
        s = float(x_scaling_factor[0].item())
        x = x.contiguous().float()
        y = torch.empty(N, Cc, 1, 1, dtype=torch.float32, device=x.device)
        _lib.call("hawq_avgpool_f32", x.data_ptr(), y.data_ptr(), N * Cc, H * W, s, _stream())
        return (y, x_scaling_factor)


class QuantMaxPool2d(Module):
    """Pass-through max pooling on (tensor, scale) tuples (reference: quant_modules.py:497-529)."""

    def __init__(self, kernel_size=3, stride=2, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.pool = nn.MaxPool2d(kernel_size=kernel_size, stride=stride, padding=padding)

    def forward(self, x, x_scaling_factor=None):
        if type(x) is tuple:
            x_scaling_factor = x[1]
            x = x[0]
        return (self.pool(x), x_scaling_factor)


class QuantDropout(Module):
    """Pass-through dropout on (tensor, scale) tuples (reference: quant_modules.py:532-554)."""

    def __init__(self, p=0):
        super().__init__()
        self.dropout = nn.Dropout(p)

    def forward(self, x, x_scaling_factor=None):
        if type(x) is tuple:
            x_scaling_factor = x[1]
            x = x[0]
        return (self.dropout(x), x_scaling_factor)


_QUANT_TYPES = (QuantAct, QuantConv2d, QuantLinear, QuantBnConv2d)


def trust_integer_buffers(model, flag: bool):
    """Mark every quantized module of `model` as running on its LOADED integer buffers / scales
    (quantized_checkpoint.pth.tar, quant_train.py:665-670) - or, ``flag`` False, as deriving them from its float parameters
    again.  The module path (forward_modules, Q_MobileNetV2) reads this per module; the fused ResNet engine reads
    ``model.engine_defaults['from_buffers']``: hawq_amd.api keeps both in step."""
    for m in model.modules():
        if isinstance(m, (QuantBnConv2d, QuantConv2d, QuantLinear)):
            m.use_integer_buffers = bool(flag)
            m._prep_key = None
        elif isinstance(m, QuantAct):   # a frozen QuantAct then keeps the loaded act_scaling_factor instead of deriving it from x_min / x_max
            m.use_integer_buffers = bool(flag)


def freeze_model(model):
    """Freeze activation ranges / BN statistics (reference: quant_modules.py:739-758)."""
    if type(model) in _QUANT_TYPES:
        model.fix()
    elif type(model) == nn.Sequential:
        for _, m in model.named_children():
            freeze_model(m)
    else:
        for attr in dir(model):
            mod = getattr(model, attr)
            if isinstance(mod, nn.Module) and 'norm' not in attr:
                freeze_model(mod)


def unfreeze_model(model):
    """Inverse of freeze_model (reference: quant_modules.py:761-780)."""
    if type(model) in _QUANT_TYPES:
        model.unfix()
    elif type(model) == nn.Sequential:
        for _, m in model.named_children():
            unfreeze_model(m)
    else:
        for attr in dir(model):
            mod = getattr(model, attr)
            if isinstance(mod, nn.Module) and 'norm' not in attr:
                unfreeze_model(mod)
