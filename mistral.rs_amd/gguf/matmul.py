"""`GgufMatMul`: host-side mirror of the reference's GGUF / ISQ-GGML `QuantMethod` (mistralrs-quant/src/gguf/mod.rs:43-52 struct,
:298-323 `try_fast_forward`, :430-432 `dequantize_w`, :440-479 `forward_raw`, :485-516 `gather_forward_raw`, :633-708 `apply_isq`,
:755-793 UQFF serde; trait: lib.rs:1515-1688).

Dispatch, as the reference's `try_fast_forward`: flat batch 1..8 -> MMVQ (`fast_mmvq.plain`); batch > 8 -> the prompt route.  On MI355X the
prompt route is the fused block-dequant -> bf16 MFMA GEMM (`fast_gemm`) for f32 activations of the formats it covers, and the MMQ ABI
(`fast_mmq.plain`, the reference's own route) otherwise; `prompt_route="mmq"` forces the latter.  `forward` casts to `quantized_act_type`
(None for GGUF: activations stay in their dtype) and adds the bias after the matmul, as `QuantMethod::forward` (lib.rs:1524-1535).
No CPU path: the weight lives on the GPU and every op goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from . import fast_gemm, fast_mmq, fast_mmvq
from .qtensor import GgmlDType, QTensor

_OUT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 30}


def dequantize(w: QTensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Packed blocks -> dense [N, K] (`QTensor::dequantize`); w = scale*q - offset in f32, one rounding to `dtype`."""
    if dtype not in _OUT:
        raise ValueError(f"dequantize: output dtype must be F32, F16 or BF16, got {dtype}")
    if not w.data.is_cuda:
        raise ValueError("dequantize: weight must live on the GPU")
    *lead, k = w.shape
    rows = 1
    for d in lead:
        rows *= d
    out = torch.empty(*w.shape, dtype=dtype, device=w.data.device)
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    fn = _lib.sym("ext", "mrs_dequantize", [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p], C.c_int)
    if fn(w.data.data_ptr(), w.dtype.id, rows, k, out.data_ptr(), _OUT[dtype], torch.cuda.current_stream().cuda_stream) != 0:
        raise ValueError(f"dequantize: unsupported ggml dtype {w.dtype.name} (K={k})")
    return out


class GgufMatMul:
    """Weight-owning quantized linear op; immutable after construction (objects are shared like `Arc<dyn QuantMethod>`)."""

    def __init__(self, q_weight: QTensor, b: torch.Tensor | None = None, prompt_route: str = "auto"):
        if len(q_weight.shape) not in (2, 3):
            raise ValueError("GgufMatMul: weight must be [N, K] (or a stacked [E, N, K] expert tensor: ISQ / statistics only)")
        if prompt_route not in ("auto", "gemm", "mmq"):
            raise ValueError("GgufMatMul: prompt_route must be auto, gemm or mmq")
        self.w, self.b, self.prompt_route = q_weight, b, prompt_route
        from ..imatrix import ImatrixLayerStats
        self.stats = ImatrixLayerStats.empty()  # gguf/mod.rs:47: interior-mutable, shared by clones of the layer

    # ---- trait surface (lib.rs:1515-1688)
    def get_qtensor(self) -> QTensor:
        return self.w

    def has_bias(self) -> bool:
        return self.b is not None

    def quantized_act_type(self):
        return None  # GGUF kernels take bf16 / f16 / f32 activations as they are (gguf/mod.rs:518-520)

    def dtype_and_device(self):
        return torch.float32, self.w.data.device

    def dequantize_w(self, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        return dequantize(self.w, dtype)

    def add_delta_w(self, delta: torch.Tensor) -> "GgufMatMul":
        """Dequantize, add, keep dense is what the reference does (gguf/mod.rs:522-560); the dense result has no GGUF kernel here."""
        raise ValueError("GgufMatMul.add_delta_w: dense (unquantized) layers are outside this path; re-quantize with apply_isq")

    def try_fast_forward(self, a: torch.Tensor):
        """gguf/mod.rs:298-323: None when no fast route applies."""
        if not fast_mmvq.supports(self.w.dtype) or a.dtype not in _OUT or len(self.w.shape) != 2:
            return None
        k = a.shape[-1]
        flat = a.numel() // k if k else 0
        if 1 <= flat <= fast_mmvq.MMVQ_MAX_BATCH:
            return fast_mmvq.plain(self.w, a)
        if flat > fast_mmvq.MMVQ_MAX_BATCH:
            if self.prompt_route != "mmq" and a.dtype == torch.float32 and fast_gemm.supports(self.w.dtype):
                try:
                    return fast_gemm.plain(self.w, a)
                except ValueError:  # a K the GEMM tiles do not cover: the MMQ route takes every K that is a multiple of the block size
                    if self.prompt_route == "gemm":
                        raise
            if self.prompt_route == "gemm":
                raise ValueError(f"GgufMatMul: the MFMA GEMM route needs f32 activations and one of its formats (got {a.dtype}, {self.w.dtype.name})")
            return fast_mmq.plain(self.w, a)
        return None

    def forward_raw(self, a: torch.Tensor) -> torch.Tensor:
        self.stats.process(a)  # gguf/mod.rs:441: free when no collection is running
        out = self.try_fast_forward(a)
        if out is None:
            raise ValueError(f"GgufMatMul: no GPU route for {self.w.dtype.name} weights with {a.dtype} activations of shape {tuple(a.shape)}")
        return out

    def forward(self, a: torch.Tensor) -> torch.Tensor:
        out = self.forward_raw(a)
        return out if self.b is None else out + self.b.to(out.dtype)

    def gather_forward_raw(self, a: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        raise ValueError("GgufMatMul.gather_forward_raw: stacked experts go through mistralrs_amd.moe (indexed GEMV) / fast_mmq.grouped")

    def embedding_forward_raw(self, ids: torch.Tensor) -> torch.Tensor:
        """Row gather + dequantize of the token rows (gguf/mod.rs embedding_forward_raw; test :815-846)."""
        n, k = self.w.shape
        ids32 = ids.reshape(-1).to(torch.int32).contiguous()
        out = torch.empty(ids32.numel(), k, dtype=torch.float32, device=self.w.data.device)
        _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
        fn = _lib.sym("ext", "mrs_embedding", [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int)
        if fn(self.w.data.data_ptr(), self.w.dtype.id, ids32.data_ptr(), out.data_ptr(), k, ids32.numel(), torch.cuda.current_stream().cuda_stream) != 0:
            raise ValueError(f"gguf embedding_forward: unsupported dtype {self.w.dtype.name}")
        return out.reshape(*ids.shape, k)

    # ---- ISQ (gguf/mod.rs:633-708, utils/isq.rs:249-287,323-361)
    def plan_isq(self, dtype: GgmlDType | None):
        from .. import isq
        return None if dtype is None else isq.get_quantization_behaviour(self.w.shape, dtype)

    def apply_isq(self, dtype: GgmlDType | None, imatrix_weight=None) -> "GgufMatMul":
        """Re-quantize to `dtype` (with the reference's fallback chain); None keeps the layer.  Dequantize -> quantize on the device.  With an
        importance vector (gguf/mod.rs:633-700: `imatrix_weight`) the layer is always re-quantized, through the weighted quantizer when the target
        has one; a stacked [E, out, in] weight goes slab by slab (`quantize_expert_stack`)."""
        from .. import isq
        if dtype is None:
            return self
        if dtype == self.w.dtype and imatrix_weight is None:  # gguf/mod.rs:645-657: same type, no importance vector -> the layer is returned untouched,
            return self                                       # BEFORE the rank is looked at (no lossy dequantize -> re-quantize of expert stacks either)
        if len(self.w.shape) == 3:
            out = isq.quantize_expert_stack(self.dequantize_w(torch.float32), dtype, imatrix_weight)
            return GgufMatMul(out, self.b, self.prompt_route)
        target = self.plan_isq(dtype)
        if target is None or (target == self.w.dtype and imatrix_weight is None):
            return self
        dense = self.dequantize_w(torch.float32)
        if imatrix_weight is not None and isq.imatrix_capable(target):
            return GgufMatMul(isq.quantize_imatrix(dense, imatrix_weight, target), self.b, self.prompt_route)
        return GgufMatMul(isq.quantize(dense, target), self.b, self.prompt_route)

    # ---- imatrix collection (gguf/mod.rs:710-739)
    def begin_track_stats(self) -> None:
        dims = self.w.shape
        dev = self.w.data.device
        if len(dims) == 3:  # stacked [E, out, in] expert weights collect per expert through the routed path
            self.stats.enable_routed(dims[0], dims[2], dev)
        else:
            self.stats.enable(dims[-1], dev)

    def process_routed_stats(self, x: torch.Tensor, ids: torch.Tensor) -> None:
        self.stats.process_routed(x, ids)

    def stats_snapshot(self):
        return self.stats.snapshot()

    def end_track_stats(self) -> torch.Tensor:
        if not self.stats.is_enabled():
            raise ValueError("`gguf` is not tracking stats.")
        try:
            return self.stats.compute_imatrix()
        finally:
            self.stats.clear()

    # ---- UQFF (gguf/mod.rs:260-280,755-793)
    def serialize_uqff(self, prefix: str) -> dict:
        from .. import uqff
        bias = None if self.b is None else self.b.detach().float().cpu().numpy()
        return uqff.serialize_gguf_layer(prefix, self.w.dtype, self.w.shape, self.w.data.cpu().numpy(), bias)

    @classmethod
    def from_uqff(cls, reader, prefix: str, device, shard=None) -> "GgufMatMul":
        lay = reader.load_gguf_layer(prefix, shard)
        w = QTensor(lay.dtype, lay.shape, torch.from_numpy(lay.packed.copy()).to(device))
        b = None if lay.bias is None else torch.from_numpy(lay.bias.copy()).to(device)
        return cls(w, b)
