"""Host-side mirror of `mistralrs-quant/src/gguf/fast_mmvq.rs` (plain :299, fused_glu :472,
fused_qkv :682): quantize the activation rows to Q8_1 in a workspace, then ONE GEMV launch.

Same argument meaning and error behaviour as the reference (its `candle_core::bail!`s become
ValueError).  Everything runs on the caller's current torch stream through the C ABI of
libmistralrsquant.so -- plain pointers + sizes, exactly what `gguf/ffi.rs` binds.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .qtensor import GgmlDType, QTensor

Q8_1_BLOCK_SIZE = 32
Q8_1_TYPE_SIZE = 36  # 2 halves + 32 int8  (fast_mmvq.rs:19-20)
MATRIX_ROW_PADDING = 512  # fast_mmvq.rs:21
MMVQ_MAX_BATCH = 8  # fast_mmvq.rs:52

_SUPPORTED = {GgmlDType.Q4_0, GgmlDType.Q4_1, GgmlDType.Q5_0, GgmlDType.Q5_1, GgmlDType.Q8_0,
              GgmlDType.Q2K, GgmlDType.Q3K, GgmlDType.Q4K, GgmlDType.Q5K, GgmlDType.Q6K}
_DT = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}
# GluActivationType codes (mistralrs-quant/src/utils/ops.rs:2601-2607)
GLU_SILU, GLU_GELU, GLU_RELU, GLU_GELU_ERF, GLU_SIGMOID = 0, 1, 2, 3, 4

_vp, _i = C.c_void_p, C.c_int


def supports(dtype: GgmlDType) -> bool:
    return dtype in _SUPPORTED


def pad(p: int, q: int) -> int:
    return (p + q - 1) // q * q


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_workspaces: dict = {}


def workspace_ensure(device: torch.device, nbytes: int) -> torch.Tensor:
    """Process-stable scratch per (device, stream) -- graph-capturable (fast_mmvq.rs:70-112)."""
    key = (device.index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _check(w: QTensor, xs: torch.Tensor, what: str):
    if not supports(w.dtype):
        raise ValueError(f"fast_mmvq: unsupported quant dtype {w.dtype!r}")
    if not w.data.is_cuda:
        raise ValueError("fast_mmvq: weight must live on the GPU")
    if xs.device != w.data.device:
        raise ValueError("fast_mmvq: input and weight are on different devices")
    if xs.dim() < 1:
        raise ValueError("fast_mmvq: input must have at least one dimension")
    nrows, ncols = w.shape
    k = xs.shape[-1]
    b_size = xs.numel() // k if k else 0
    if k != ncols:
        raise ValueError(f"fast_mmvq: shape mismatch: weight [{nrows}, {ncols}] vs input tail {k}")
    if b_size == 0 or b_size > MMVQ_MAX_BATCH:
        raise ValueError(f"fast_mmvq: batch size {b_size} out of supported range 1..={MMVQ_MAX_BATCH}")
    if xs.dtype not in _DT:
        raise ValueError(f"fast_mmvq: input dtype must be BF16, F16, or F32, got {xs.dtype}")
    return nrows, k, b_size


def quantize_q8_1(xs: torch.Tensor, k: int, b_size: int) -> tuple[torch.Tensor, int]:
    """launch_mmvq_gguf_quantize_q8_1_<dt>: returns (workspace, stride_col_y in blocks)."""
    k_padded = pad(k, MATRIX_ROW_PADDING)
    nblk = k_padded // Q8_1_BLOCK_SIZE
    ws = workspace_ensure(xs.device, b_size * nblk * Q8_1_TYPE_SIZE)
    fn = _lib.sym("quant", f"launch_mmvq_gguf_quantize_q8_1_{_DT[xs.dtype]}", [_vp, _vp, _i, _i, _i, _vp])
    fn(xs.data_ptr(), ws.data_ptr(), k, k_padded, b_size, _stream())
    return ws, nblk


def plain(w: QTensor, xs: torch.Tensor) -> torch.Tensor:
    nrows, k, b_size = _check(w, xs, "plain")
    xs = xs.contiguous()
    ws, stride_col_y = quantize_q8_1(xs, k, b_size)
    out = torch.empty(*xs.shape[:-1], nrows, dtype=xs.dtype, device=xs.device)
    fn = _lib.sym("quant", f"launch_mmvq_gguf_{w.dtype.tag}_{_DT[xs.dtype]}_plain",
                  [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp])
    fn(w.data.data_ptr(), ws.data_ptr(), out.data_ptr(), k, nrows, stride_col_y, nrows, b_size, _stream())
    return out


def fused_glu(gate: QTensor, up: QTensor, xs: torch.Tensor, activation: int = GLU_SILU) -> torch.Tensor:
    """act(gate . x) * (up . x) in one launch (fast_mmvq.rs:472)."""
    if gate.dtype != up.dtype:
        raise ValueError("fast_mmvq fused GLU: gate and up must share the quant dtype")
    if gate.shape != up.shape:
        raise ValueError(f"fast_mmvq fused GLU: shape mismatch: gate {gate.shape} vs up {up.shape}")
    nrows, k, b_size = _check(gate, xs, "fused_glu")
    xs = xs.contiguous()
    ws, stride_col_y = quantize_q8_1(xs, k, b_size)
    out = torch.empty(*xs.shape[:-1], nrows, dtype=xs.dtype, device=xs.device)
    fn = _lib.sym("quant", f"launch_mmvq_gguf_{gate.dtype.tag}_{_DT[xs.dtype]}_fused_glu",
                  [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp])
    fn(gate.data.data_ptr(), up.data.data_ptr(), ws.data_ptr(), out.data_ptr(), k, nrows, stride_col_y, nrows,
       b_size, int(activation), _stream())
    return out


def fused_qkv(q: QTensor, k_w: QTensor, v: QTensor, xs: torch.Tensor):
    """Q, K, V projections from one shared Q8_1 activation (fast_mmvq.rs:682)."""
    if not (q.dtype == k_w.dtype == v.dtype):
        raise ValueError("fast_mmvq fused QKV: q, k and v must share the quant dtype")
    if not (q.shape[1] == k_w.shape[1] == v.shape[1]):
        raise ValueError("fast_mmvq fused QKV: q, k and v must share the input dimension")
    nq, k, b_size = _check(q, xs, "fused_qkv")
    nk, nv = k_w.shape[0], v.shape[0]
    xs = xs.contiguous()
    ws, stride_col_y = quantize_q8_1(xs, k, b_size)
    outs = [torch.empty(*xs.shape[:-1], n, dtype=xs.dtype, device=xs.device) for n in (nq, nk, nv)]
    fn = _lib.sym("quant", f"launch_mmvq_gguf_{q.dtype.tag}_{_DT[xs.dtype]}_fused_qkv",
                  [_vp] * 7 + [_i] * 6 + [_vp])
    fn(q.data.data_ptr(), k_w.data.data_ptr(), v.data.data_ptr(), ws.data_ptr(),
       outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), k, nq, nk, nv, stride_col_y, b_size, _stream())
    return tuple(outs)
