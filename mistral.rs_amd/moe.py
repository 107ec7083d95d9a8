"""Mixture-of-experts decode block: host-side mirror of `SparseMoeBlock::forward` (mistralrs-core/src/models/mixtral.rs:280-304),
`moe_router_topk` (ops.rs:259-336, Mixtral settings: softmax over all experts -> top-k -> renormalise) and the quantized expert
path `MoEExperts::forward` -> `qmatmul_indexed_moe_forward` (moe/experts/mod.rs:306-342, gguf/cuda.rs:514-588): stacked packed
experts [E][N][K/blk]; per token, each selected expert runs fused gate/up (+SiLU*up) and a down GEMV whose result is added to the
residual stream scaled by the routing weight.  The expert id is read on the device, the host never sees the routing."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .gguf.qtensor import GgmlDType

_vp, _i = C.c_void_p, C.c_int


class StackedExperts:
    """E packed weights [N, K] of one type in one uint8 tensor (GGUF `ffn_{gate,up,down}_exps`, rank 3)."""

    def __init__(self, dtype: GgmlDType, num_experts: int, n: int, k: int, data: torch.Tensor):
        self.dtype, self.num_experts, self.n, self.k = dtype, num_experts, n, k
        self.expert_bytes = n * dtype.row_bytes(k)
        if data.dtype != torch.uint8 or data.numel() != num_experts * self.expert_bytes:
            raise ValueError(f"StackedExperts: expected {num_experts * self.expert_bytes} bytes, got {data.numel()}")
        self.data = data.contiguous().view(-1)


class SparseMoeBlock:
    def __init__(self, gate_w: torch.Tensor, gate_exps: StackedExperts, up_exps: StackedExperts, down_exps: StackedExperts, top_k: int = 2,
                 rms_eps: float = 1e-5):
        if gate_exps.dtype != up_exps.dtype or (gate_exps.n, gate_exps.k) != (up_exps.n, up_exps.k):
            raise ValueError("gate and up experts must share dtype and shape")
        self.gate_w = gate_w.to(torch.float32).contiguous()  # ffn_gate_inp [E, K]
        self.g, self.u, self.d, self.top_k, self.eps = gate_exps, up_exps, down_exps, top_k, rms_eps
        L = _lib.load("ext")
        _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
        self._router = _lib.sym("ext", "mrs_moe_router_topk", [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i)
        self._gate_up = _lib.sym("ext", "mrs_moe_decode_gate_up", [_vp, _vp, C.c_size_t, _vp, _i, _i, _i, _vp, _vp, C.c_float, _i, _vp, _i, _vp], _i)
        self._down = _lib.sym("ext", "mrs_moe_decode_down", [_vp, C.c_size_t, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp], _i)
        self._rms = _lib.sym("core", "mrs_rms_norm_f32", [_vp, _vp, _vp, _i, _i, C.c_float, C.c_int64])

    def forward(self, h: torch.Tensor, norm_w: torch.Tensor):
        """h: f32 [tokens, K] residual stream (updated in place: h += moe(rms_norm(h))); returns (h, ids, weights)."""
        tokens, K = h.shape
        E, ff = self.g.num_experts, self.g.n
        dev, st = h.device, torch.cuda.current_stream().cuda_stream
        xn = torch.empty_like(h)
        self._rms(h.data_ptr(), norm_w.data_ptr(), xn.data_ptr(), tokens, K, self.eps, st)
        ids = torch.empty(tokens, self.top_k, dtype=torch.int32, device=dev)
        wts = torch.empty(tokens, self.top_k, dtype=torch.float32, device=dev)
        if self._router(xn.data_ptr(), self.gate_w.data_ptr(), tokens, E, K, self.top_k, 1, ids.data_ptr(), wts.data_ptr(), None, st) != 0:
            raise ValueError("moe router: unsupported shape")
        stride = (ff + 511) // 512 * 512 // 32
        y = torch.zeros(self.top_k, stride * 36, dtype=torch.uint8, device=dev)
        for t in range(tokens):  # b = 1 per launch (decode); all activations of a token's experts are computed before h changes
            for s in range(self.top_k):
                rc = self._gate_up(self.g.data.data_ptr(), self.u.data.data_ptr(), self.g.expert_bytes, ids[t, s:].data_ptr(), self.g.dtype.id, ff, K,
                                   h[t].data_ptr(), norm_w.data_ptr(), self.eps, 0, y[s].data_ptr(), stride, st)
                if rc != 0:
                    raise ValueError("moe gate/up: unsupported dtype or shape")
            for s in range(self.top_k):
                rc = self._down(self.d.data.data_ptr(), self.d.expert_bytes, ids[t, s:].data_ptr(), wts[t, s:].data_ptr(), self.d.dtype.id, K, ff,
                                y[s].data_ptr(), stride, h[t].data_ptr(), st)
                if rc != 0:
                    raise ValueError("moe down: unsupported dtype or shape")
        return h, ids, wts
