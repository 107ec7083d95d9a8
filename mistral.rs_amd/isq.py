"""In-situ quantization: host-side mirror of `apply_isq` for the Q8_0 target (mistralrs-quant/src/utils/isq.rs:24-83,323-361,
gguf/mod.rs:633-708).  The dense weight (bf16 / f16 / f32, as loaded from safetensors) is quantized ON THE GPU into standard GGML
Q8_0 blocks and becomes a `QTensor` that the GGUF kernels consume -- `IsqType::Q8_0` -> `GgufMatMul`."""
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
