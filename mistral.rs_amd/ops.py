"""Host-side mirror of the glue ops on the hot path:
  apply_rotary_qk   mistralrs-quant/src/rotary/mod.rs:851 (-> rotary_embedding[_positions])
  fused_glu         mistralrs-quant/src/utils/ops.rs:2953
  rms_norm          mistralrs-core/src/layers.rs:403-414 (candle rms_norm)
  add_rms_norm / rms_norm_residual   mistralrs-core/src/cuda/ffi.rs:75-140
All through the C ABI on the caller's current stream; raises if the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_CODE = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
_TAG = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
_vp, _i, _l, _u, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def apply_rotary_qk(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, is_neox: bool,
                    positions: torch.Tensor | None = None) -> None:
    """In place.  q [tokens, heads, head_size], k [tokens, kv_heads, head_size] (token stride free);
    cos/sin [rows, rot_dim/2] in the tensor dtype; positions uint32/int32 [tokens] selects the rows
    (without it row = token index)."""
    if q.dtype not in _CODE or k.dtype != q.dtype or cos.dtype != q.dtype or sin.dtype != q.dtype:
        raise ValueError("rotary: q, k, cos and sin must share a dtype in {f16, bf16, f32}")
    tokens, heads, head_size = q.shape
    kv_heads = k.shape[1]
    pairs = cos.shape[-1]
    if k.shape[0] != tokens or k.shape[2] != head_size or 2 * pairs > head_size:
        raise ValueError(f"rotary: shape mismatch q {tuple(q.shape)} k {tuple(k.shape)} cos {tuple(cos.shape)}")
    if not (cos.is_contiguous() and sin.is_contiguous()):
        raise ValueError("rotary: cos/sin must be contiguous")
    if positions is None:
        fn = _lib.sym("quant", "rotary_embedding", [_vp] * 4 + [_i, _i, _l, _i, _i, _i, _l, _l, _u, _l])
        fn(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), int(is_neox), head_size, tokens, pairs, heads,
           kv_heads, q.stride(0), k.stride(0), _CODE[q.dtype], _stream())
    else:
        fn = _lib.sym("quant", "rotary_embedding_positions", [_vp] * 5 + [_i, _i, _l, _i, _i, _i, _i, _l, _l, _u, _l])
        fn(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), positions.data_ptr(), int(is_neox), head_size,
           tokens, pairs, cos.shape[0], heads, kv_heads, q.stride(0), k.stride(0), _CODE[q.dtype], _stream())


def fused_glu(a: torch.Tensor, b: torch.Tensor, activation: int = 0) -> torch.Tensor:
    """act(a) * b  (utils/ops.rs:2953): last dim contiguous, rows may be strided views of a wider tensor."""
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in _TAG:
        raise ValueError(f"fused_glu: shape/dtype mismatch {tuple(a.shape)} {a.dtype} vs {tuple(b.shape)} {b.dtype}")
    cols = a.shape[-1]
    a2, b2 = a.reshape(-1, cols) if a.is_contiguous() else a, b.reshape(-1, cols) if b.is_contiguous() else b
    if a2.dim() != 2 or a2.stride(1) != 1 or b2.stride(1) != 1:
        raise ValueError("fused_glu: inputs must be 2-D with a contiguous last dimension")
    out = torch.empty(a2.shape, dtype=a.dtype, device=a.device)
    fn = _lib.sym("quant", f"fused_glu_{_TAG[a.dtype]}", [_vp, _vp, _vp, _u, _u, _u, _u, _i, _vp])
    fn(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), a2.shape[0], cols, a2.stride(0), b2.stride(0), int(activation), _stream())
    return out.reshape(a.shape)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty_like(x)
    fn = _lib.sym("core", f"mrs_rms_norm_{_TAG[x.dtype]}", [_vp, _vp, _vp, _i, _i, _f, _l])
    fn(x.data_ptr(), weight.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps, _stream())
    return out


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """(residual_out, normed) = (x + residual, rms_norm(x + residual) * w)."""
    x, residual = x.contiguous(), residual.contiguous()
    res_out, norm_out = torch.empty_like(x), torch.empty_like(x)
    fn = _lib.sym("core", f"add_rms_norm_{_TAG[x.dtype]}", [_vp] * 5 + [_i, _i, _f, _l])
    fn(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), res_out.data_ptr(), norm_out.data_ptr(),
       x.numel() // x.shape[-1], x.shape[-1], eps, _stream())
    return res_out, norm_out


def rms_norm_residual(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                      scale: torch.Tensor | None = None) -> torch.Tensor:
    """(residual + rms_norm(x) * w) * scale."""
    x, residual = x.contiguous(), residual.contiguous()
    out = torch.empty_like(x)
    fn = _lib.sym("core", f"rms_norm_residual_{_TAG[x.dtype]}", [_vp] * 5 + [_i, _i, _f, _l])
    fn(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), scale.data_ptr() if scale is not None else None,
       out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps, _stream())
    return out


def moe_router_topk(logits: torch.Tensor, top_k: int, score_mode: int = 1, weight_mode: int = 0, renormalize: bool = True,
                    selection_bias: torch.Tensor | None = None, expert_scale: torch.Tensor | None = None, clamp: tuple | None = None,
                    norm_min: float = 0.0, output_scale: float = 1.0):
    """`moe_router_topk` (mistralrs-core/src/ops.rs:259-336 over cuda/ffi.rs:523-579): logits [rows, n_experts] (f32 / f16 / bf16) ->
    (ids uint32 [rows, top_k], weights f32 [rows, top_k]).  score_mode 0 raw / 1 softmax / 2 sigmoid; weight_mode 0 score / 1 softmax
    over the picked raw logits / 2 sigmoid(raw).  Mixtral: softmax, weights = scores, renormalised."""
    if logits.dim() != 2 or logits.dtype not in _TAG or not logits.is_contiguous():
        raise ValueError("moe_router_topk: contiguous logits [rows, n_experts] in f32 / f16 / bf16")
    rows, n_experts = logits.shape
    if n_experts not in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 576):
        raise ValueError(f"moe_router_topk: unsupported expert count {n_experts}")
    if not 1 <= top_k <= n_experts:
        raise ValueError(f"moe_router_topk: top_k {top_k} out of range for {n_experts} experts")
    for t in (selection_bias, expert_scale):
        if t is not None and (t.dtype != torch.float32 or t.numel() != n_experts or not t.is_contiguous()):
            raise ValueError("moe_router_topk: selection_bias / expert_scale must be contiguous f32 [n_experts]")
    ids = torch.empty(rows, top_k, dtype=torch.int32, device=logits.device)
    weights = torch.empty(rows, top_k, dtype=torch.float32, device=logits.device)
    import ctypes as C
    fn = _lib.sym("core", f"moe_router_topk_{_TAG[logits.dtype]}", [_vp] * 5 + [_i] * 5 + [C.c_bool, C.c_bool, _f, _f, _f, _f, _l])
    fn(logits.data_ptr(), weights.data_ptr(), ids.data_ptr(), selection_bias.data_ptr() if selection_bias is not None else None,
       expert_scale.data_ptr() if expert_scale is not None else None, rows, n_experts, top_k, score_mode, weight_mode, bool(renormalize),
       clamp is not None, float(clamp[0]) if clamp else 0.0, float(clamp[1]) if clamp else 0.0, float(norm_min), float(output_scale), _stream())
    return ids, weights
