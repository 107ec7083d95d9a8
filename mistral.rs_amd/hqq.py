"""Host-side mirror of `HqqLayer` (mistralrs-quant/src/hqq/mod.rs:571-1171, quantize.rs:9-84, optimize.rs:29-95) over the drop-in
HQQ C ABI of libmistralrsquant.so (`dequantize_{8,4,2,1}bit_u8_kernel_*`, `dequantize_3bit_32_kernel_*`, `launch_pack_*bit_kernel`).

    layer = HqqLayer.quantize(w, HqqConfig(bits=4, group_size=64))     # axis 0, channel-wise, proximal optimiser (20 steps)
    w_hat = layer.dequantize()                                         # HIP unpack + (q - zero) * scale, shape of w
    y     = layer.forward(x)                                           # dequantize_w + dense linear (+ bias), as the reference

PyTorch owns the buffers and runs the quantizer's elementwise tensor algebra (load-time work, exactly the candle tensor ops the
reference runs) and the dense matmul (a plain library GEMM, as the reference's UnquantLinear); the packed format, the packing and
the unpack + dequantize are the HIP kernels.  No CPU fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib

_PACK = {8: 1, 4: 2, 3: 10, 2: 4, 1: 8}
_KERNEL = {8: "8bit_u8_kernel", 4: "4bit_u8_kernel", 2: "2bit_u8_kernel", 1: "1bit_u8_kernel", 3: "3bit_32_kernel"}
_TAG = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}
OPTIMIZER_HQQ_DEFAULT_STEPS = 20  # hqq/mod.rs


@dataclass(frozen=True)
class HqqConfig:
    bits: int = 4
    group_size: int = 64
    axis: int = 0                       # the GPU / CPU dequantizers of the reference require axis == 0 (mod.rs:839-844,885-890)
    optimization_steps: int | None = None
    round_zeros: bool = False
    channel_wise: bool = True

    def __post_init__(self):
        if self.bits not in _PACK:
            raise ValueError(f"HQQ bits must be one of 8, 4, 3, 2, 1, got {self.bits}")
        if self.group_size <= 0:
            raise ValueError("HQQ group_size must be positive")


def _need_gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise ValueError("hqq: tensors must live on the GPU (no CPU fallback in this package)")


def pack(bits: int, wq: torch.Tensor) -> torch.Tensor:
    """Unpacked values [rows, w] (any integer / float dtype holding 0 .. 2^bits - 1) -> packed [rows / P, w] (u8; i32 for 3 bit)."""
    _need_gpu(wq)
    if wq.dim() != 2:
        raise ValueError("hqq.pack: expected a 2-D tensor [rows, width]")
    rows, width = wq.shape
    p = _PACK[bits]
    st = torch.cuda.current_stream().cuda_stream
    if bits == 3:
        src = wq.to(torch.int32).contiguous()  # the reference feeds u32
        if rows % p:  # the reference zero-pads the rows to a multiple of 10 before packing (hqq/mod.rs:401-410); dequantize() trims
            src = torch.cat([src, torch.zeros(p - rows % p, width, dtype=torch.int32, device=wq.device)], 0).contiguous()
            rows = src.shape[0]
        out = torch.empty(rows // p, width, dtype=torch.int32, device=wq.device)
    else:
        src = wq.to(torch.uint8).contiguous()
        out = torch.empty(rows // p, width, dtype=torch.uint8, device=wq.device)
    if bits == 8:
        _lib.sym("quant", "launch_pack_8bit_kernel", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p])(src.data_ptr(), out.data_ptr(), src.numel(), st)
    else:
        _lib.sym("quant", f"launch_pack_{bits}bit_kernel", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p])(
            src.data_ptr(), out.data_ptr(), rows, width, st)
    return out


def dequantize_packed(bits: int, wq: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor) -> torch.Tensor:
    """Packed [h, w] + scale / zero [w] (f32 / f16 / bf16, same dtype) -> [P*h, w] of that dtype (the raw kernel of the C ABI)."""
    _need_gpu(wq, scale, zero)
    if scale.dtype != zero.dtype or scale.dtype not in _TAG:
        raise ValueError(f"Expected all dtypes to be the same, got ({scale.dtype}, {zero.dtype}).")
    if not (wq.is_contiguous() and scale.is_contiguous() and zero.is_contiguous()):
        raise ValueError("All tensors must be contiguous!")
    want = torch.int32 if bits == 3 else torch.uint8
    if wq.dtype != want or wq.dim() != 2:
        raise ValueError(f"hqq: packed weights must be a 2-D {want} tensor")
    h, w = wq.shape
    if scale.numel() != w or zero.numel() != w:
        raise ValueError(f"hqq: scale / zero must hold {w} values (one per group column)")
    out = torch.empty(_PACK[bits] * h, w, dtype=scale.dtype, device=wq.device)
    # the reference ABI has no stream argument (default stream): order it after the current stream's producers and before its consumers
    cur = torch.cuda.current_stream()
    if cur.cuda_stream != 0:
        cur.synchronize()
    fn = _lib.sym("quant", f"dequantize_{_KERNEL[bits]}_{_TAG[scale.dtype]}", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int])
    fn(wq.data_ptr(), scale.data_ptr(), zero.data_ptr(), out.data_ptr(), h, w)
    if cur.cuda_stream != 0:
        torch.cuda.default_stream().synchronize()
    return out


def _shrink_lp(x: torch.Tensor, beta: float, lp_norm: float) -> torch.Tensor:
    if lp_norm == 1.0:
        return x.sign() * (x.abs() - 1.0 / beta).relu()
    return x.sign() * (x.abs() - (1.0 / beta) * x.abs().pow(lp_norm - 1.0)).relu()


def _round_half_away(x: torch.Tensor) -> torch.Tensor:  # candle's round()
    return x.sign() * (x.abs() + 0.5).floor()


class HqqLayer:
    def __init__(self, w_q, zeros, scales, w_shape, cfg: HqqConfig, bias=None):
        self.w_q, self.zeros, self.scales, self.w_shape, self.cfg, self.bias = w_q, zeros, scales, tuple(w_shape), cfg, bias

    # ---- UQFF (hqq/mod.rs:1268-1336): the 8- and 4-bit widths have a UQFF type; artifacts load whole
    def uqff_type(self):
        return {8: "HQQ8", 4: "HQQ4"}.get(self.cfg.bits)

    def serialize_uqff(self, prefix: str) -> dict:
        from . import uqff
        c = self.cfg
        cpu = lambda t: None if t is None else t.detach().cpu().contiguous()
        return uqff.serialize_hqq_layer(prefix, cpu(self.w_q), cpu(self.scales), cpu(self.zeros), self.w_shape, c.bits, c.group_size, c.axis,
                                        c.optimization_steps, c.round_zeros, c.channel_wise, cpu(self.bias))

    @classmethod
    def from_uqff(cls, reader, prefix: str, device, shard=None) -> "HqqLayer":
        d = reader.load_hqq_layer(prefix, shard)
        cfg = HqqConfig(bits=d.bits, group_size=d.group_size, axis=d.axis, optimization_steps=d.optimization_steps, round_zeros=d.round_zeros,
                        channel_wise=d.channel_wise)
        to = lambda t: None if t is None else t.to(device)
        return cls(to(d.w_q), to(d.zeros), to(d.scales), d.w_shape, cfg, to(d.bias))

    @classmethod
    def quantize(cls, w: torch.Tensor, cfg: HqqConfig) -> "HqqLayer":
        """HqqLayer::quantize (quantize.rs:9-84) with optimize_weights_proximal_legacy (optimize.rs:44-95)."""
        _need_gpu(w)
        if cfg.axis != 0 or not cfg.channel_wise:
            raise ValueError("hqq: only axis == 0, channel_wise quantisation is dequantizable on the GPU path")
        if w.numel() % cfg.group_size:
            raise ValueError(f"`group_size` should be divisible by the tensor number of elements, which are {w.numel()}, got a group size of {cfg.group_size}.")
        wf = w.to(torch.float32).reshape(cfg.group_size, -1)
        mn, mx = wf.amin(0, keepdim=True), wf.amax(0, keepdim=True)
        max_v = float(round(2.0 ** cfg.bits - 1.0))
        scale = (max_v / (mx - mn)).clamp(0.0, 2e4)
        zero = -mn * scale
        if cfg.round_zeros:
            zero = _round_half_away(zero)
        beta, best = 10.0, 1e4
        for _ in range(cfg.optimization_steps if cfg.optimization_steps is not None else OPTIMIZER_HQQ_DEFAULT_STEPS):
            wq = _round_half_away(wf * scale + zero).clamp(0.0, max_v)
            wr = (wq - zero) / scale
            we = _shrink_lp(wf - wr, beta, 0.7)
            zero = (wq - (wf - we) * scale).mean(0, keepdim=True)
            beta *= 1.01
            err = float((wf - wr).abs().mean())
            if err < best:
                best = err
            else:
                break
        wq = _round_half_away(wf * scale + zero).clamp(0.0, max_v)
        return cls(pack(cfg.bits, wq), zero.contiguous(), (1.0 / scale).contiguous(), w.shape, cfg)

    def with_bias(self, bias: torch.Tensor) -> "HqqLayer":
        self.bias = bias
        return self

    def to_dtype(self, dtype: torch.dtype) -> "HqqLayer":
        """scales / zeros in the compute dtype (the dequantizer's output dtype follows them, mod.rs:893-1080)."""
        return HqqLayer(self.w_q, self.zeros.to(dtype), self.scales.to(dtype), self.w_shape, self.cfg, self.bias)

    def dequantize(self) -> torch.Tensor:
        if self.cfg.axis != 0:
            raise ValueError(f"CUDA HQQ dequantization requires axis == 0, got {self.cfg.axis}.")
        out = dequantize_packed(self.cfg.bits, self.w_q, self.scales.reshape(-1), self.zeros.reshape(-1))
        n = 1
        for d in self.w_shape:
            n *= d
        return out.reshape(-1)[:n].reshape(self.w_shape)  # 3 bit: ten values per i32 may over-cover the group rows

    def forward(self, xs: torch.Tensor) -> torch.Tensor:
        """HqqLayer::forward_raw (mod.rs:1092-1100,1163-1171): x @ W^T (+ bias).  Decode-sized inputs (<= 8 rows, 4 / 8 bit, group 64) take the fused
        dequant-GEMV of csrc/ext_hqq_gemv.hip (the packed bytes are read once, dequantize_w() is never materialised); everything else dequantizes and
        uses the dense matmul like the reference."""
        dt = self.scales.dtype
        if (self.cfg.bits in (4, 8) and self.cfg.group_size == 64 and self.cfg.axis == 0 and xs.dim() == 2 and 1 <= xs.shape[0] <= 8 and len(self.w_shape) == 2
                and self.w_shape[0] % 64 == 0 and self.w_shape[1] % 4 == 0 and dt in _TAG and self.zeros.dtype == dt):
            _need_gpu(xs)
            n, k = self.w_shape
            x = xs.to(dt).contiguous()
            out = torch.empty(x.shape[0], n, dtype=dt, device=x.device)
            bias = self.bias.to(dt).contiguous() if self.bias is not None else None
            fn = _lib.sym("ext", "mrs_hqq_gemv", [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int)
            rc = fn(self.cfg.bits, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dt], self.w_q.data_ptr(), self.scales.data_ptr(), self.zeros.data_ptr(),
                    bias.data_ptr() if bias is not None else None, x.data_ptr(), k, out.data_ptr(), n, n, k, x.shape[0], torch.cuda.current_stream().cuda_stream)
            if rc == 0:
                return out
        w = self.dequantize()
        y = xs.to(w.dtype) @ w.t()
        return y + self.bias.to(y.dtype) if self.bias is not None else y
