"""Host-side mirror of `mistralrs-quant/src/gguf/fast_mmq.rs`: the prompt-sized quantized matmul route.

    shared_lhs :528   quantize the activations ONCE to block_q8_1_mmq, then one MMQ launch per weight
    plain :762, fused_qkv :768, fused_glu :781, fused_ffn :800, down_from_glu :636, grouped :822 (MoE)

Same argument meaning and error behaviour as the reference (its `candle_core::bail!`s become ValueError), same workspace sizing
(`batch * k_padded/128 * 144 + MMQ_X_MAX * 144` bytes, process-stable per (device, stream)), same launch sequence through the C ABI of
libmistralrsquant.so (`launch_mmq_quantize_q8_1_<layout>`, `launch_mmq_quantize_glu_q8_1_<layout>`, `launch_mmq_gguf_<t>[_moe]`; gguf/ffi.rs:1313-1452).
This is the COMPATIBLE route (csrc/mmq.hip); the fast MI355X prompt path is gguf/fast_gemm.py (block dequant fused into a bf16 MFMA GEMM).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .. import ops
from .qtensor import GgmlDType, QTensor

QK8_1 = 32
BLOCK_Q8_1_MMQ_SIZE = 4 * QK8_1 + 4 * 4  # fast_mmq.rs:24
MATRIX_ROW_PADDING = 512  # fast_mmq.rs:25
MMQ_X_MAX = 128  # fast_mmq.rs:26

_SUPPORTED = {GgmlDType.Q4_0, GgmlDType.Q4_1, GgmlDType.Q5_0, GgmlDType.Q5_1, GgmlDType.Q8_0,
              GgmlDType.Q2K, GgmlDType.Q3K, GgmlDType.Q4K, GgmlDType.Q5K, GgmlDType.Q6K}
# ds_layout_for (fast_mmq.rs; kernels/mmq_gguf/mmq_gguf.cuh:100-135)
_DS4 = {GgmlDType.Q4_0, GgmlDType.Q4_1, GgmlDType.Q5_1, GgmlDType.Q4K, GgmlDType.Q5K}
_TYPE_X = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 30}  # fast_mmq.rs:591-596

_vp, _i, _l = C.c_void_p, C.c_int, C.c_int64


def supports(dtype: GgmlDType) -> bool:
    return dtype in _SUPPORTED


def ds_layout_for(dtype: GgmlDType) -> str:
    return "DS4" if dtype in _DS4 else ("D2S6" if dtype == GgmlDType.Q2K else "D4")


def pad(p: int, q: int) -> int:
    return (p + q - 1) // q * q


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_workspaces: dict = {}


def workspace_ensure(device: torch.device, nbytes: int) -> torch.Tensor:
    key = (device.index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _scratch(device, rows: int, k: int):
    k_padded = pad(pad(k, MATRIX_ROW_PADDING), 4 * QK8_1)
    nbytes = rows * (k_padded // (4 * QK8_1)) * BLOCK_Q8_1_MMQ_SIZE + MMQ_X_MAX * BLOCK_Q8_1_MMQ_SIZE
    return workspace_ensure(device, nbytes), k_padded


def _device_info():
    # cc / nsm / smpbo / warp_size feed the reference's tile selection and stream-k; the MI355X launchers accept and ignore them
    return 0, 256, 160 << 10, 64


def _launch_mmq(w: QTensor, scratch: torch.Tensor, out: torch.Tensor, k: int, nrows: int, batch: int, type_dst: int):
    cc, nsm, smpbo, warp = _device_info()
    fn = _lib.sym("quant", f"launch_mmq_gguf_{w.dtype.tag}", [_vp] * 4 + [_l] * 5 + [_i, _i, _l, _i, _i, _vp])
    fn(None, w.data.data_ptr(), scratch.data_ptr(), out.data_ptr(), k, nrows, batch, k // w.dtype.block_size, nrows, cc, nsm, smpbo, warp, type_dst,
       _stream())


def shared_lhs(weights, xs: torch.Tensor):
    """One activation quantization pass, one MMQ launch per weight; outputs keep the input dtype (fast_mmq.rs:528-635)."""
    if not weights:
        raise ValueError("fast_mmq shared_lhs: at least one weight is required")
    first = weights[0]
    dtype = first.dtype
    if not supports(dtype):
        raise ValueError(f"fast_mmq shared_lhs: unsupported quant dtype {dtype!r}")
    if not first.data.is_cuda:
        raise ValueError("fast_mmq shared_lhs: weights must live on the GPU")
    ncols = first.shape[1]
    for w in weights[1:]:
        if w.dtype != dtype:
            raise ValueError("fast_mmq shared_lhs: weight dtype mismatch")
        if w.data.device != first.data.device:
            raise ValueError("fast_mmq shared_lhs: weights are on different devices")
        if w.shape[1] != ncols:
            raise ValueError(f"fast_mmq shared_lhs: weight ncols mismatch {ncols} vs {w.shape[1]}")
    if xs.device != first.data.device:
        raise ValueError("fast_mmq shared_lhs: input and weights are on different devices")
    if xs.dim() < 1:
        raise ValueError("fast_mmq shared_lhs: input must have at least one dimension")
    k = xs.shape[-1]
    batch = xs.numel() // k if k else 0
    if batch == 0:
        raise ValueError("fast_mmq shared_lhs: batch size must be greater than zero")
    if k != ncols:
        raise ValueError(f"fast_mmq shared_lhs: weight ncols {ncols} does not match input tail {k}")
    if k % dtype.block_size != 0:
        raise ValueError(f"fast_mmq shared_lhs: k={k} not divisible by qk={dtype.block_size}")
    if xs.dtype not in _TYPE_X:
        raise ValueError(f"fast_mmq shared_lhs: input dtype must be BF16, F16, or F32, got {xs.dtype}")
    xs = xs.contiguous()
    type_x = _TYPE_X[xs.dtype]
    scratch, k_padded = _scratch(xs.device, batch, k)
    q = _lib.sym("quant", f"launch_mmq_quantize_q8_1_{ds_layout_for(dtype)}", [_vp, _vp, _vp, _i] + [_l] * 8 + [_vp])
    q(xs.data_ptr(), None, scratch.data_ptr(), type_x, k, k, 0, 0, k_padded, batch, 1, 1, _stream())
    outs = []
    for w in weights:
        nrows = w.shape[0]
        out = torch.empty(*xs.shape[:-1], nrows, dtype=xs.dtype, device=xs.device)
        _launch_mmq(w, scratch, out, k, nrows, batch, type_x)
        outs.append(out)
    return outs


def down_from_glu(down: QTensor, gate: torch.Tensor, up: torch.Tensor, activation: int = 0) -> torch.Tensor:
    """down . (act(gate) * up): the GLU product is formed and quantized in one pass, never materialised (fast_mmq.rs:636-760)."""
    dtype = down.dtype
    if not supports(dtype):
        raise ValueError(f"fast_mmq down_from_glu: unsupported quant dtype {dtype!r}")
    if not down.data.is_cuda:
        raise ValueError("fast_mmq down_from_glu: weight must live on the GPU")
    if gate.shape != up.shape:
        raise ValueError(f"fast_mmq down_from_glu: gate/up shape mismatch {tuple(gate.shape)} vs {tuple(up.shape)}")
    if gate.dtype != up.dtype:
        raise ValueError(f"fast_mmq down_from_glu: gate/up dtype mismatch {gate.dtype} vs {up.dtype}")
    if gate.device != down.data.device or up.device != down.data.device:
        raise ValueError("fast_mmq down_from_glu: tensors are on different devices")
    if gate.dim() < 1:
        raise ValueError("fast_mmq down_from_glu: input must have at least one dimension")
    k = gate.shape[-1]
    batch = gate.numel() // k if k else 0
    if batch == 0:
        raise ValueError("fast_mmq down_from_glu: batch size must be greater than zero")
    nrows, ncols = down.shape
    if k != ncols:
        raise ValueError(f"fast_mmq down_from_glu: weight ncols {ncols} does not match input tail {k}")
    if k % dtype.block_size != 0:
        raise ValueError(f"fast_mmq down_from_glu: k={k} not divisible by qk={dtype.block_size}")
    if gate.dtype not in _TYPE_X:
        raise ValueError(f"fast_mmq down_from_glu: input dtype must be BF16, F16, or F32, got {gate.dtype}")
    gate, up = gate.contiguous(), up.contiguous()
    type_x = _TYPE_X[gate.dtype]
    scratch, k_padded = _scratch(gate.device, batch, k)
    q = _lib.sym("quant", f"launch_mmq_quantize_glu_q8_1_{ds_layout_for(dtype)}", [_vp, _vp, _vp, _vp, _i] + [_l] * 4 + [_i, _vp])
    q(gate.data_ptr(), up.data_ptr(), None, scratch.data_ptr(), type_x, k, k, k_padded, batch, int(activation), _stream())
    out = torch.empty(*gate.shape[:-1], nrows, dtype=gate.dtype, device=gate.device)
    _launch_mmq(down, scratch, out, k, nrows, batch, type_x)
    return out


def plain(w: QTensor, xs: torch.Tensor) -> torch.Tensor:
    return shared_lhs([w], xs)[0]


def fused_qkv(q_w: QTensor, k_w: QTensor, v_w: QTensor, xs: torch.Tensor):
    q, k, v = shared_lhs([q_w, k_w, v_w], xs)
    return q, k, v


def fused_glu(gate_w: QTensor, up_w: QTensor, xs: torch.Tensor, activation: int = 0) -> torch.Tensor:
    if gate_w.shape != up_w.shape:
        raise ValueError(f"fast_mmq fused_glu: gate/up shape mismatch {gate_w.shape} vs {up_w.shape}")
    gate, up = shared_lhs([gate_w, up_w], xs)
    return ops.fused_glu(gate, up, activation)


def fused_ffn(gate_w: QTensor, up_w: QTensor, down_w: QTensor, xs: torch.Tensor, activation: int = 0) -> torch.Tensor:
    if gate_w.shape != up_w.shape:
        raise ValueError(f"fast_mmq fused_ffn: gate/up shape mismatch {gate_w.shape} vs {up_w.shape}")
    gate, up = shared_lhs([gate_w, up_w], xs)
    return down_from_glu(down_w, gate, up, activation)


def grouped(weight: QTensor, xs: torch.Tensor, ids_src: torch.Tensor, ids_dst: torch.Tensor, expert_bounds: torch.Tensor, total_assignments: int,
            ncols_max: int, num_experts: int) -> torch.Tensor:
    """One MoE projection, llama.cpp-style grouped MMQ (fast_mmq.rs:822-1010): compact expert-sorted row j reads input row ids_src[j], belongs to
    the expert whose [expert_bounds[e], expert_bounds[e+1]) contains j, and lands in output row ids_dst[j].  weight [experts, nrows, k]; f32 out."""
    dtype = weight.dtype
    if not supports(dtype):
        raise ValueError(f"fast_mmq grouped: unsupported quant dtype {dtype!r}")
    if xs.dim() != 2:
        raise ValueError("fast_mmq grouped: input must be 2-D")
    k = xs.shape[1]
    if len(weight.shape) != 3:
        raise ValueError("fast_mmq grouped: weight must be [experts, rows, cols]")
    experts, nrows, ncols = weight.shape
    if experts != num_experts:
        raise ValueError(f"fast_mmq grouped: expected {num_experts} experts, got {experts}")
    if k != ncols:
        raise ValueError(f"fast_mmq grouped: shape mismatch: weight cols {ncols} vs input tail {k}")
    if k % dtype.block_size != 0:
        raise ValueError(f"fast_mmq grouped: k={k} not divisible by qk={dtype.block_size}")
    if xs.dtype not in _TYPE_X:
        raise ValueError(f"fast_mmq grouped: input dtype must be BF16, F16, or F32, got {xs.dtype}")
    xs = xs.contiguous()
    scratch, k_padded = _scratch(xs.device, total_assignments, k)
    q = _lib.sym("quant", f"launch_mmq_quantize_q8_1_{ds_layout_for(dtype)}", [_vp, _vp, _vp, _i] + [_l] * 8 + [_vp])
    q(xs.data_ptr(), ids_src.data_ptr(), scratch.data_ptr(), _TYPE_X[xs.dtype], k, k, 0, 0, k_padded, total_assignments, 1, 1, _stream())
    out = torch.empty(total_assignments, nrows, dtype=torch.float32, device=xs.device)
    cc, nsm, smpbo, warp = _device_info()
    fn = _lib.sym("quant", f"launch_mmq_gguf_{dtype.tag}_moe", [_vp] * 6 + [_l] * 7 + [_i, _i, _l, _i, _vp])
    fn(None, weight.data.data_ptr(), scratch.data_ptr(), ids_dst.data_ptr(), expert_bounds.data_ptr(), out.data_ptr(), k, nrows, total_assignments,
       k // dtype.block_size, nrows, num_experts, ncols_max, cc, nsm, smpbo, warp, _stream())
    return out
