"""In-situ quantization: host-side mirror of `apply_isq` for the GGML targets (mistralrs-quant/src/utils/isq.rs:24-83,247-287,323-361,
gguf/mod.rs:633-708).  The dense weight (bf16 / f16 / f32, as loaded from safetensors) is quantized ON THE GPU into standard GGML
blocks (Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q4_K Q5_K Q6_K, bit-identical to GGML's reference quantizers) and becomes a `QTensor` that the GGUF
kernels consume -- `IsqType::Q4K` ... -> `GgufMatMul`.  `get_quantization_behaviour` mirrors the reference's dtype fallback chain."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .gguf.qtensor import GgmlDType, QTensor

_SRC = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 30}


def quantize_q8_0(w: torch.Tensor) -> QTensor:
    """w: dense [N, K] on the GPU -> QTensor(Q8_0).  Raises ValueError like the reference's dtype fallback trigger when
    K is not a multiple of the block size (utils/isq.rs:249-287)."""
    if w.dim() != 2 or not w.is_cuda:
        raise ValueError("isq: expected a 2-D GPU weight")
    if w.dtype not in _SRC:
        raise ValueError(f"isq: unsupported source dtype {w.dtype}")
    n, k = w.shape
    if k % GgmlDType.Q8_0.block_size:
        raise ValueError(f"isq: last dimension {k} is not a multiple of the Q8_0 block size 32")
    w = w.contiguous()
    out = torch.empty(n * (k // 32) * 34, dtype=torch.uint8, device=w.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_isq_quantize_q8_0", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p], C.c_int)
    if fn(w.data_ptr(), _SRC[w.dtype], out.data_ptr(), n * k, torch.cuda.current_stream().cuda_stream) != 0:
        raise ValueError("isq: quantizer refused the tensor")
    return QTensor(GgmlDType.Q8_0, (n, k), out)


_ISQ_TARGETS = (GgmlDType.Q4_0, GgmlDType.Q4_1, GgmlDType.Q5_0, GgmlDType.Q5_1, GgmlDType.Q8_0, GgmlDType.Q4K, GgmlDType.Q5K, GgmlDType.Q6K)
# get_fallback (utils/isq.rs:247-261): the 32-wide `Q` formats are more lenient than the 256-wide `K` formats
_FALLBACK = {GgmlDType.Q2K: GgmlDType.Q4_0, GgmlDType.Q3K: GgmlDType.Q4_0, GgmlDType.Q4K: GgmlDType.Q4_1, GgmlDType.Q5K: GgmlDType.Q5_0,
             GgmlDType.Q6K: GgmlDType.Q5_1, GgmlDType.Q8K: GgmlDType.Q8_1}


def get_quantization_behaviour(shape, dtype: GgmlDType):
    """utils/isq.rs:263-287: the dtype this tensor is quantized with, or None (skip): the requested dtype if the last dimension is a
    multiple of its block size, else its fallback (recursively)."""
    if dtype == GgmlDType.F32:
        return None
    if len(shape) > 0 and shape[-1] % dtype.block_size == 0:
        return dtype
    fb = _FALLBACK.get(dtype)
    return None if fb is None else get_quantization_behaviour(shape, fb)


def quantize(w: torch.Tensor, dtype: GgmlDType) -> QTensor:
    """w: dense [N, K] on the GPU -> QTensor(dtype) with GGML-reference blocks, quantized on the device.  Raises ValueError when K is not a
    multiple of the block size (callers pick the dtype with get_quantization_behaviour first, as generate_isq! does)."""
    if w.dim() != 2 or not w.is_cuda:
        raise ValueError("isq: expected a 2-D GPU weight")
    if w.dtype not in _SRC:
        raise ValueError(f"isq: unsupported source dtype {w.dtype}")
    if dtype not in _ISQ_TARGETS:
        raise ValueError(f"isq: {dtype.name} is not a device ISQ target")
    n, k = w.shape
    if k % dtype.block_size:
        raise ValueError(f"isq: last dimension {k} is not a multiple of the {dtype.name} block size {dtype.block_size}")
    w = w.contiguous()
    out = torch.empty(n * (k // dtype.block_size) * dtype.type_size, dtype=torch.uint8, device=w.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_isq_quantize", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p], C.c_int)
    if fn(w.data_ptr(), _SRC[w.dtype], out.data_ptr(), n * k, dtype.id, torch.cuda.current_stream().cuda_stream) != 0:
        raise ValueError("isq: quantizer refused the tensor")
    return QTensor(dtype, (n, k), out)
