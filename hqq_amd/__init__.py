"""hqq_amd — MI355X (gfx950) native implementation of HQQ's two hot paths.

  * Quantizer.quantize: the half-quadratic proximal solver + bit-packing   (hqq_amd/csrc/quantize.hip)
  * HQQLinear.forward:  fused unpack -> dequantize -> GEMV / MFMA GEMM      (hqq_amd/csrc/gemv.hip, gemm.hip)

behind the reference's HQQLinear / HQQBackend / prepare_for_inference plug-in surface
(hqq_amd.core.quantize, hqq_amd.utils.patching).  Compute goes through the C ABI in include/hqq_hip.h.
"""
__version__ = "0.1.0"
