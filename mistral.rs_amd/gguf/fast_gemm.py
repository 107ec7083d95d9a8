"""Host-side mirror of the prefill branch (`fast_mmq::plain`, mistralrs-quant/src/gguf/fast_mmq.rs:762): one fused
dequant -> bf16 MFMA GEMM launch per weight (mrs_gemm_q_f32 in libmrs_hip_ext.so).  x: f32 [..., K] -> f32 [..., N]."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .qtensor import GgmlDType, QTensor

_SUPPORTED = {GgmlDType.Q4K, GgmlDType.Q5K, GgmlDType.Q6K, GgmlDType.Q8_0}


def supports(dtype: GgmlDType) -> bool:
    return dtype in _SUPPORTED


def plain(w: QTensor, xs: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    if not supports(w.dtype):
        raise ValueError(f"fast_gemm: unsupported quant dtype {w.dtype!r}")
    if xs.dtype != torch.float32 or not xs.is_cuda:
        raise ValueError("fast_gemm: input must be an f32 GPU tensor")
    n, k = w.shape
    if xs.shape[-1] != k:
        raise ValueError(f"fast_gemm: shape mismatch: weight [{n}, {k}] vs input tail {xs.shape[-1]}")
    x2 = xs.reshape(-1, k).contiguous()
    m = x2.shape[0]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=xs.device)
        accumulate = False
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_gemm_q_f32", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int)
    rc = fn(w.data.data_ptr(), w.dtype.id, n, k, x2.data_ptr(), x2.stride(0), out.data_ptr(), out.stride(0), m, int(accumulate),
            torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise ValueError(f"fast_gemm: unsupported shape K={k} for {w.dtype.name}")
    return out.reshape(*xs.shape[:-1], n)
