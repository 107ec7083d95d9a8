"""In-situ quantization: host-side mirror of `apply_isq` for the GGML targets (mistralrs-quant/src/utils/isq.rs:24-83,247-287,323-361,
gguf/mod.rs:633-708).  The dense weight (bf16 / f16 / f32, as loaded from safetensors) is quantized ON THE GPU into standard GGML
blocks (Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K, bit-identical to GGML's reference quantizers) and becomes a `QTensor` that the GGUF
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


_ISQ_TARGETS = (GgmlDType.Q4_0, GgmlDType.Q4_1, GgmlDType.Q5_0, GgmlDType.Q5_1, GgmlDType.Q8_0, GgmlDType.Q2K, GgmlDType.Q3K, GgmlDType.Q4K, GgmlDType.Q5K,
                GgmlDType.Q6K)
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



def supports_imatrix(dtype: GgmlDType) -> bool:
    """`IsqType::supports_imatrix` (mistralrs-quant/src/lib.rs:1038-1043): the five K-quants consume an importance vector; every one of them has a weighted
    device quantizer (`imatrix_capable` is the same set)."""
    return dtype in (GgmlDType.Q2K, GgmlDType.Q3K, GgmlDType.Q4K, GgmlDType.Q5K, GgmlDType.Q6K)


def pack_factor(dtype: GgmlDType, src_bytes: int = 2) -> int:
    """`IsqType::pack_factor` for the GGML targets (lib.rs:946-957,996-1016 `block_pack_factor`): the integer factor by which the reference's loader divides
    the dense element count to size an ISQ'd tensor: floor(block * src_bytes / type_size), lowered while block / factor * src_bytes < type_size."""
    dense = dtype.block_size * src_bytes
    factor = max(dense // max(dtype.type_size, 1), 1)
    while factor > 1 and dtype.block_size // factor * src_bytes < dtype.type_size:
        factor -= 1
    return factor


def imatrix_capable(dtype: GgmlDType) -> bool:
    """gguf/mod.rs:221-224: the K-quants take an importance vector; all five have a weighted device quantizer (`mrs_isq_quantize_imatrix`)."""
    return supports_imatrix(dtype)


def quantize_imatrix(w: torch.Tensor, imatrix, dtype: GgmlDType) -> QTensor:
    """`QTensor::quantize_imatrix`: w dense [N, K] on the GPU, imatrix one f32 per input column (`ImatrixLayerStats.compute_imatrix`, or a vector
    of a `.cimatrix` file).  Raises ValueError for targets without a weighted quantizer (callers fall back to `quantize`, as
    `quantize_expert_stack` does, gguf/mod.rs:247-252)."""
    if w.dim() != 2 or not w.is_cuda:
        raise ValueError("isq: expected a 2-D GPU weight")
    if w.dtype not in _SRC:
        raise ValueError(f"isq: unsupported source dtype {w.dtype}")
    if not imatrix_capable(dtype):
        raise ValueError(f"isq: {dtype.name} has no importance-weighted device quantizer")
    n, k = w.shape
    if k % dtype.block_size:
        raise ValueError(f"isq: last dimension {k} is not a multiple of the {dtype.name} block size {dtype.block_size}")
    im = torch.as_tensor(imatrix, dtype=torch.float32).to(w.device).contiguous().reshape(-1)
    if im.numel() != k:
        raise ValueError(f"isq: imatrix has {im.numel()} entries, the weight {k} input columns")
    w = w.contiguous()
    out = torch.empty(n * (k // dtype.block_size) * dtype.type_size, dtype=torch.uint8, device=w.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_isq_quantize_imatrix", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    if fn(w.data_ptr(), _SRC[w.dtype], out.data_ptr(), n, k, dtype.id, im.data_ptr(), torch.cuda.current_stream().cuda_stream) != 0:
        raise ValueError("isq: the weighted quantizer refused the tensor")
    return QTensor(dtype, (n, k), out)


def quantize_expert_stack(stack: torch.Tensor, dtype: GgmlDType, imatrix=None) -> QTensor:
    """`GgufMatMul::quantize_expert_stack` (gguf/mod.rs:200-262): a stacked [E, out, in] expert tensor is quantized slab by slab -- row-block
    formats never cross expert boundaries, so the byte concatenation equals quantizing the whole stack.  `imatrix` is [in] (shared) or [E * in]
    (per expert, flattened); any other length is ignored with a warning; an all-zero vector (an expert that saw no traffic) and targets without
    a weighted quantizer fall back to the plain quantizer."""
    import warnings
    if stack.dim() != 3 or not stack.is_cuda:
        raise ValueError("isq: expected a stacked [E, out, in] GPU tensor")
    e_n, out_dim, in_dim = stack.shape
    target = get_quantization_behaviour((out_dim, in_dim), dtype)
    if target is None:
        raise ValueError("isq: the expert slabs are not quantizable (the reference keeps them dense: F32)")
    im = None
    if imatrix is not None:
        im = torch.as_tensor(imatrix, dtype=torch.float32).reshape(-1)
        if im.numel() not in (in_dim, e_n * in_dim):
            warnings.warn(f"Expert stack imatrix length {im.numel()} matches neither in_dim {in_dim} nor {e_n}x{in_dim}; quantizing without it.")
            im = None
    parts = []
    for e in range(e_n):
        v = None
        if im is not None:
            v = im if im.numel() == in_dim else im[e * in_dim:(e + 1) * in_dim]
        if v is not None and imatrix_capable(target) and bool((v != 0).any()):
            parts.append(quantize_imatrix(stack[e], v, target).data)
        else:
            parts.append(quantize(stack[e], target).data)
    return QTensor(target, (e_n, out_dim, in_dim), torch.cat(parts))
