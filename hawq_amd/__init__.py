"""hawq_amd - MI355X-native integer inference path for HAWQ's quantized ResNets.

Product code: HIP kernels behind a C ABI (csrc/, include/hawq_mi355.h), a Python mirror of the
reference's operator API (quant_modules, quant_utils, q_resnet), and the fused integer engine.
"""
from .bit_schedules import bit_config_dict, get_bit_config  # noqa: F401
from .skeleton import build_float_resnet, init_synthetic, synthetic_images  # noqa: F401

__all__ = ["bit_config_dict", "get_bit_config", "build_float_resnet", "init_synthetic", "synthetic_images"]
