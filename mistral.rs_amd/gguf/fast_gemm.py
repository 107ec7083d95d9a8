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


def to_slabs(xs: torch.Tensor) -> torch.Tensor:
    """f32 [M, K] -> bf16 slabs [K/64, M, 64] (mrs_convert_f32_bf16_slabs): the activation layout of `plain_bf16`."""
    if xs.dtype != torch.float32 or not xs.is_cuda or xs.dim() != 2 or xs.stride(1) != 1:
        raise ValueError("fast_gemm.to_slabs: f32 GPU matrix [M, K] with unit inner stride")
    m, k = xs.shape
    y = torch.empty(max(k // 64, 1), m, 64, dtype=torch.bfloat16, device=xs.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_convert_f32_bf16_slabs", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    if fn(xs.data_ptr(), xs.stride(0), m, k, y.data_ptr(), torch.cuda.current_stream().cuda_stream):
        raise ValueError("fast_gemm.to_slabs: K must be a multiple of 64 and the row stride a multiple of 4")
    return y


def plain_bf16(w: QTensor, xs: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False, split_k: bool = True,
               workspace: torch.Tensor | None = None) -> torch.Tensor:
    """Large-M path (mrs_gemm_q_bf16_multi): the activations are rounded to bf16 ONCE into k-slab-major layout (`to_slabs`; pass its
    result to reuse it across weights), 256-row tiles, split-K partials when the shape has fewer tiles than CUs.  Same values as
    `plain` up to f32 summation order.  Returns f32 [M, N]."""
    if not supports(w.dtype):
        raise ValueError(f"fast_gemm: unsupported quant dtype {w.dtype!r}")
    if not xs.is_cuda or xs.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("fast_gemm: input must be an f32 GPU matrix or the bf16 slabs of to_slabs()")
    n, k = w.shape
    if xs.dtype == torch.float32:
        if xs.shape[-1] != k:
            raise ValueError(f"fast_gemm: shape mismatch: weight [{n}, {k}] vs input tail {xs.shape[-1]}")
        slabs = to_slabs(xs.reshape(-1, k).contiguous())
    else:
        if xs.dim() != 3 or xs.shape[2] != 64 or xs.shape[0] * 64 != k or not xs.is_contiguous():
            raise ValueError(f"fast_gemm: shape mismatch: weight [{n}, {k}] vs slabs {tuple(xs.shape)}")
        slabs = xs
    m = slabs.shape[1]
    st = torch.cuda.current_stream().cuda_stream
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=xs.device)
        accumulate = False
    wsb = _lib.sym("ext", "mrs_gemm_q_bf16_workspace_bytes", [C.c_int], C.c_size_t)(m) if split_k else 0
    own = workspace is None or workspace.numel() * workspace.element_size() < wsb
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=xs.device) if own else workspace
    wp, np_, op, ld = (C.c_void_p * 1)(w.data.data_ptr()), (C.c_int * 1)(n), (C.c_void_p * 1)(out.data_ptr()), (C.c_int * 1)(out.stride(0))
    fn = _lib.sym("ext", "mrs_gemm_q_bf16_multi", [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int)
    rc = fn(1, wp, np_, op, ld, w.dtype.id, k, slabs.data_ptr(), m, int(accumulate), ws.data_ptr() if split_k else None, wsb, st)
    if rc != 0:
        raise ValueError(f"fast_gemm: unsupported shape K={k} for {w.dtype.name}")
    if own or slabs is not xs:
        torch.cuda.current_stream().synchronize()  # workspace / slabs stay alive until the launch ran
    return out


def rms_norm_slabs(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm of f32 [M, K] straight into the bf16 slab layout (mrs_rms_norm_bf16_slabs) == to_slabs(ops.rms_norm(x, w, eps))."""
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous():
        raise ValueError("fast_gemm.rms_norm_slabs: contiguous f32 matrix [M, K] and f32 weight")
    m, k = x.shape
    y = torch.empty(max(k // 64, 1), m, 64, dtype=torch.bfloat16, device=x.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_rms_norm_bf16_slabs", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p], C.c_int)
    if fn(x.data_ptr(), weight.data_ptr(), m, k, eps, y.data_ptr(), torch.cuda.current_stream().cuda_stream):
        raise ValueError("fast_gemm.rms_norm_slabs: K must be a multiple of 64")
    return y


def glu_slabs(g: torch.Tensor, u: torch.Tensor, activation: int = 0) -> torch.Tensor:
    """act(g) * u of f32 [M, N] straight into the bf16 slab layout (mrs_glu_bf16_slabs) == to_slabs(ops.fused_glu(g, u, act))."""
    if g.dtype != torch.float32 or g.shape != u.shape or g.dim() != 2 or g.stride(1) != 1 or u.stride() != g.stride():
        raise ValueError("fast_gemm.glu_slabs: two f32 matrices [M, N] of equal shape and strides")
    m, n = g.shape
    y = torch.empty(max(n // 64, 1), m, 64, dtype=torch.bfloat16, device=g.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_glu_bf16_slabs", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    if fn(g.data_ptr(), u.data_ptr(), g.stride(0), m, n, int(activation), y.data_ptr(), torch.cuda.current_stream().cuda_stream):
        raise ValueError("fast_gemm.glu_slabs: N must be a multiple of 64 and the row stride a multiple of 4")
    return y


def moe_grouped_bf16(w: QTensor, num_experts: int, slabs: torch.Tensor, bounds: torch.Tensor, sorted_ids: torch.Tensor, topk: int, gather: bool,
                     out: torch.Tensor, route_w: torch.Tensor | None = None) -> torch.Tensor:
    """Grouped MoE GEMM of a prompt on the matrix cores (mrs_moe_gemm_q_bf16): w = experts stacked along the rows [E * N, K]; slabs = bf16 slabs of the
    tokens (gather: row of sorted position pos is token sorted_ids[pos] // topk) or of the routes in sorted order; bounds / sorted_ids = launch_moe_dispatch
    tables (int32).  route_w None: out[pos] = W_e . x[row] (f32 [routes, N]); else out[token] += route_w[flat] * (W_e . x[pos]) with f32 atomics."""
    if not supports(w.dtype):
        raise ValueError(f"fast_gemm: unsupported quant dtype {w.dtype!r}")
    rows, k = w.shape
    if rows % num_experts or slabs.dtype != torch.bfloat16 or slabs.dim() != 3 or slabs.shape[0] * 64 != k or not slabs.is_contiguous():
        raise ValueError("fast_gemm.moe_grouped_bf16: stacked experts [E * N, K] and bf16 slabs [K / 64, rows, 64]")
    if bounds.dtype != torch.int32 or sorted_ids.dtype != torch.int32 or bounds.numel() != num_experts + 1 or out.dtype != torch.float32 or out.stride(-1) != 1:
        raise ValueError("fast_gemm.moe_grouped_bf16: int32 dispatch tables and an f32 output")
    n = rows // num_experts
    fn = _lib.sym("ext", "mrs_moe_gemm_q_bf16", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int)
    rc = fn(w.data.data_ptr(), w.dtype.id, n, k, num_experts, slabs.data_ptr(), slabs.shape[1], bounds.data_ptr(), sorted_ids.data_ptr(), topk, int(gather),
            route_w.data_ptr() if route_w is not None else None, out.data_ptr(), out.stride(0), sorted_ids.numel(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise ValueError(f"fast_gemm.moe_grouped_bf16: unsupported shape K={k} for {w.dtype.name}")
    return out


def fused_glu_bf16(w_gate: QTensor, w_up: QTensor, slabs: torch.Tensor, activation: int = 0) -> torch.Tensor:
    """act(W_g x) * (W_u x) of a prompt in ONE launch (mrs_gemm_q_bf16_glu; role of fast_mmq::fused_glu, gguf/fast_mmq.rs:762-821): bf16 slabs in, bf16 slabs
    [N / 64, M, 64] out -- the down GEMM's activation layout.  Same bits as plain_bf16(gate), plain_bf16(up), glu_slabs."""
    if not supports(w_gate.dtype) or w_gate.dtype != w_up.dtype or w_gate.shape != w_up.shape:
        raise ValueError("fast_gemm.fused_glu_bf16: gate and up must share a supported quant dtype and shape")
    n, k = w_gate.shape
    if slabs.dtype != torch.bfloat16 or slabs.dim() != 3 or slabs.shape[2] != 64 or slabs.shape[0] * 64 != k or not slabs.is_contiguous() or n % 64:
        raise ValueError(f"fast_gemm.fused_glu_bf16: weight [{n}, {k}] (N % 64 == 0) vs slabs {tuple(slabs.shape)}")
    m = slabs.shape[1]
    y = torch.empty(n // 64, m, 64, dtype=torch.bfloat16, device=slabs.device)
    fn = _lib.sym("ext", "mrs_gemm_q_bf16_glu", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    if fn(w_gate.data.data_ptr(), w_up.data.data_ptr(), w_gate.dtype.id, n, k, slabs.data_ptr(), m, int(activation), y.data_ptr(), torch.cuda.current_stream().cuda_stream):
        raise ValueError(f"fast_gemm.fused_glu_bf16: unsupported shape K={k} for {w_gate.dtype.name}")
    return y
