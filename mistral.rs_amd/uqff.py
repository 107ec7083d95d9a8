"""UQFF read / write for GGUF-quantized linear layers: host-side mirror of the on-disk step right after ISQ.

Reference: mistralrs-quant/src/uqff/mod.rs (version tensors :27-33,100-106; `shard_range` :109-160; `bias_shard` :162-190;
`slice_blocked_data` :224-330), uqff/tensor.rs (UqffTensor = one safetensors entry), gguf/mod.rs:54-93 (dtype codes = ggml type ids),
:260-280 (`GgufMatMul::from_uqff`), :755-793 (`serialize_uqff`).  A UQFF file is a safetensors file; a GGUF-quantized layer `prefix` is

    {prefix}.weight.format   u8  scalar   QuantizedSerdeType::Gguf = 0            (lib.rs:1178-1186)
    {prefix}.weight          u8  [nbytes] the packed GGUF blocks, row-major [N][K/blk] (unchanged bytes: what the kernels read)
    {prefix}.weight.dtype    u32 scalar   ggml type id (0 f32, 1 f16, 2 q4_0 ... 14 q6_k, 30 bf16)
    {prefix}.weight.shape    u32 [rank]   logical dims
    {prefix}.bias            optional, tensor dtype

preceded by `uqff.version.{major,minor,patch}` u32 scalars (1.2.0).  Loading applies a tensor-parallel `Shard` to the packed bytes
(whole rows, or columns at quant-block multiples) and to the bias (`Narrow` for a sharded output dim, `Skip` for a sharded input dim: the caller
adds it after the all-reduce).  Byte / shape logic on numpy arrays only; nothing here touches a device.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .distributed import Shard
from .gguf.qtensor import GgmlDType

UQFF_VERSION = (1, 2, 0)  # uqff/mod.rs:27-29
VERSION_KEYS = ("uqff.version.major", "uqff.version.minor", "uqff.version.patch")
WEIGHT_FORMAT_SUFFIX = "weight.format"
SERDE_GGUF, SERDE_UNQUANT, SERDE_HQQ, SERDE_FP8, SERDE_AFQ, SERDE_F8Q8, SERDE_MXFP4 = range(7)  # lib.rs:1178-1186


def version_tensors() -> dict:
    return {k: np.array(v, dtype=np.uint32) for k, v in zip(VERSION_KEYS, UQFF_VERSION)}


def shard_range(shard: Shard | None, dims) -> tuple[int, int, int] | None:
    """None for a full load, else (dim, start, len) (uqff/mod.rs:109-160, same error text)."""
    if shard is None:
        return None
    dims = list(dims)
    if shard.dim >= len(dims):
        raise ValueError(f"Cannot shard dimension {shard.dim} of rank-{len(dims)} tensor.")
    size = dims[shard.dim]
    if shard.offset is not None:
        end = shard.offset + shard.length
        if end > size:
            raise ValueError(f"Shard range {shard.offset}..{end} exceeds dimension {shard.dim} of size {size}.")
        return None if shard.offset == 0 and shard.length == size else (shard.dim, shard.offset, shard.length)
    if shard.world_size == 0:
        raise ValueError("Shard world size must be non-zero.")
    if shard.rank >= shard.world_size:
        raise ValueError(f"Shard rank {shard.rank} is outside world size {shard.world_size}.")
    if shard.world_size == 1:
        return None
    if size % shard.world_size:
        raise ValueError(f"Weight shard dim {shard.dim} of size {size} is not divisible by world size {shard.world_size}.")
    n = size // shard.world_size
    return shard.dim, shard.rank * n, n


def bias_shard(rng, weight_rank: int):
    """'full' | 'skip' | ('narrow', dim, start, len)   (uqff/mod.rs:162-190)."""
    if weight_rank < 2:
        return "full" if rng is None else "skip"
    if rng is None:
        return "full"
    dim, start, n = rng
    if dim == weight_rank - 1:
        return "skip"
    if dim < weight_rank - 1:
        return ("narrow", dim, start, n)
    return "skip"


def slice_blocked_data(data: np.ndarray, dims, block: int, block_bytes: int, dim: int, start: int, n: int) -> np.ndarray:
    """Slice raw block-quantized bytes along `dim`; the last dim is packed `block` elements per `block_bytes` bytes (uqff/mod.rs:224-330)."""
    dims = list(dims)
    if block == 0 or block_bytes == 0:
        raise ValueError("Packed block sizes must be non-zero.")
    if not dims:
        raise ValueError("Cannot shard scalar packed data.")
    if dim >= len(dims):
        raise ValueError(f"Cannot shard dimension {dim} of rank-{len(dims)} packed tensor.")
    last, size, end = dims[-1], dims[dim], start + n
    if end > size:
        raise ValueError(f"Packed shard range {start}..{end} exceeds dimension {dim} of size {size}.")
    if last % block:
        raise ValueError(f"Cannot shard block-quantized data: last dim {last} is not a multiple of block size {block}.")
    row_bytes = last // block * block_bytes
    rows = int(np.prod(dims[:-1], dtype=np.int64)) if len(dims) > 1 else 1
    expected = rows * row_bytes
    data = np.asarray(data, dtype=np.uint8).reshape(-1)
    if data.size < expected:
        raise ValueError(f"Packed tensor needs {expected} bytes for shape {dims}, but only {data.size} are available.")
    if dim == len(dims) - 1:
        if start % block or n % block:
            raise ValueError(f"Sharding the packed dim requires block alignment: start {start}, len {n}, block {block}.")
        off, sub = start // block * block_bytes, n // block * block_bytes
        return np.ascontiguousarray(data[:expected].reshape(rows, row_bytes)[:, off:off + sub]).reshape(-1)
    inner = int(np.prod(dims[dim + 1:-1], dtype=np.int64)) if dim + 1 < len(dims) - 1 else 1
    pre = int(np.prod(dims[:dim], dtype=np.int64)) if dim else 1
    view = data[:expected].reshape(pre, size, inner * row_bytes)
    return np.ascontiguousarray(view[:, start:end]).reshape(-1)


def serialize_gguf_layer(prefix: str, dtype: GgmlDType, shape, packed, bias=None) -> dict:
    """`GgufMatMul::serialize_uqff` (gguf/mod.rs:755-793): name -> numpy array, in the reference's order."""
    packed = np.ascontiguousarray(np.asarray(packed, dtype=np.uint8).reshape(-1))
    out = {
        f"{prefix}.{WEIGHT_FORMAT_SUFFIX}": np.array(SERDE_GGUF, dtype=np.uint8),
        f"{prefix}.weight": packed,
        f"{prefix}.weight.dtype": np.array(dtype.id, dtype=np.uint32),
        f"{prefix}.weight.shape": np.array(list(shape), dtype=np.uint32),
    }
    if bias is not None:
        out[f"{prefix}.bias"] = np.ascontiguousarray(bias)
    return out


def serialize_hqq_layer(prefix: str, w_q, scales, zeros, w_shape, bits: int, group_size: int, axis: int, optimization_steps, round_zeros: bool,
                        channel_wise: bool, bias=None) -> dict:
    """`HqqLayer::serialize_uqff` (mistralrs-quant/src/hqq/mod.rs:1275-1318): name -> numpy array, in the reference's order.  Only the 8- and 4-bit widths
    have a UQFF type (`uqff_type`, :1268-1274: HQQ8 / HQQ4); `optimization_steps` None is stored as 0."""
    if bits not in (8, 4):
        raise ValueError("Cannot serialize unsupported HQQ bit width as UQFF.")
    out = {
        f"{prefix}.{WEIGHT_FORMAT_SUFFIX}": np.array(SERDE_HQQ, dtype=np.uint8),
        f"{prefix}.weight": w_q,          # numpy or torch (CPU) tensors, dtype preserved (u8 quants; f32 / f16 / bf16 scales and zeros)
        f"{prefix}.weight.scales": scales,
        f"{prefix}.weight.zeros": zeros,
        f"{prefix}.weight.shape": np.array(list(w_shape), dtype=np.uint32),
        f"{prefix}.weight.bits": np.array(bits, dtype=np.uint8),
        f"{prefix}.weight.group_size": np.array(group_size, dtype=np.uint32),
        f"{prefix}.weight.axis": np.array(axis, dtype=np.uint8),
        f"{prefix}.weight.optimization_steps": np.array(optimization_steps or 0, dtype=np.uint32),
        f"{prefix}.weight.round_zeros": np.array(int(bool(round_zeros)), dtype=np.uint8),
        f"{prefix}.weight.channel_wise": np.array(int(bool(channel_wise)), dtype=np.uint8),
    }
    if bias is not None:
        out[f"{prefix}.bias"] = bias
    return out


@dataclass
class HqqLayerData:
    """What `HqqLayer::from_uqff` reads back (hqq/mod.rs:797-826)."""
    w_q: "torch.Tensor"      # CPU tensors in the stored dtypes
    scales: "torch.Tensor"
    zeros: "torch.Tensor"
    w_shape: tuple
    bits: int
    group_size: int
    axis: int
    optimization_steps: int | None
    round_zeros: bool
    channel_wise: bool
    bias: "torch.Tensor | None"


def write(path: str, layers: dict) -> None:
    """Version tensors + the given entries as one safetensors file.  Values are numpy arrays, or torch tensors where numpy has no dtype for them (bf16 HQQ
    scales / zeros): the file then goes through safetensors.torch, same bytes on disk."""
    tensors = dict(version_tensors())
    for k, v in layers.items():
        if k in tensors:
            raise ValueError(f"duplicate UQFF tensor `{k}`")
        tensors[k] = v
    if any(not isinstance(v, np.ndarray) for v in tensors.values()):
        import torch
        from safetensors.torch import save_file as save_torch
        save_torch({k: (v.detach().cpu().contiguous() if isinstance(v, torch.Tensor) else torch.from_numpy(v if v.ndim == 0 else np.ascontiguousarray(v))) for k, v in tensors.items()}, path)  # 0-d scalars stay 0-d (uqff/tensor.rs:52-58)
        return
    from safetensors.numpy import save_file
    save_file(tensors, path)


@dataclass
class GgufLayer:
    dtype: GgmlDType
    shape: tuple
    packed: np.ndarray  # uint8, the (sharded) GGUF blocks
    bias: np.ndarray | None
    bias_mode: str  # 'full' | 'skip' | 'narrow'


class UqffReader:
    """Reads the entries back; the major version must match (a newer minor only adds tensors)."""

    def __init__(self, path: str):
        from safetensors import safe_open
        self._path = path
        self._f = safe_open(path, framework="np")
        self._keys = set(self._f.keys())
        missing = [k for k in VERSION_KEYS if k not in self._keys]
        if missing:
            raise ValueError(f"Missing `{missing[0]}`")
        self.version = tuple(self.load_u32_scalar(k) for k in VERSION_KEYS)
        if self.version[0] != UQFF_VERSION[0]:
            raise ValueError(f"UQFF major version {self.version[0]} is not supported (this build reads {UQFF_VERSION[0]}.x).")

    def keys(self):
        return sorted(self._keys)

    def has(self, key: str) -> bool:
        return key in self._keys

    def _get(self, key: str) -> np.ndarray:
        if key not in self._keys:
            raise ValueError(f"Missing `{key}`")
        return self._f.get_tensor(key)

    def load_u8_scalar(self, key: str) -> int:
        t = self._get(key)
        if t.dtype != np.uint8 or t.shape != ():
            raise ValueError(f"UQFF tensor `{key}` is not a u8 scalar.")
        return int(t)

    def load_u32_scalar(self, key: str) -> int:
        t = self._get(key)
        if t.dtype != np.uint32 or t.shape != ():
            raise ValueError(f"UQFF tensor `{key}` is not a u32 scalar.")
        return int(t)

    def load_u32_vec(self, key: str) -> list:
        t = self._get(key)
        if t.dtype != np.uint32:
            raise ValueError(f"UQFF tensor `{key}` is not a u32 vector.")
        return [int(v) for v in t.reshape(-1)]

    def load_raw_u8(self, key: str) -> np.ndarray:
        t = self._get(key)
        if t.dtype != np.uint8:
            raise ValueError(f"UQFF tensor `{key}` is not raw u8 data.")
        return t.reshape(-1)

    def serde_type(self, prefix: str) -> int:
        return self.load_u8_scalar(f"{prefix}.{WEIGHT_FORMAT_SUFFIX}")

    def load_bias(self, prefix: str, rng, weight_rank: int):
        key = f"{prefix}.bias"
        if key not in self._keys:
            return None, "full"
        mode = bias_shard(rng, weight_rank)
        if mode == "skip":
            return None, "skip"
        b = self._get(key)
        if mode == "full":
            return b, "full"
        _, dim, start, n = mode
        # the bias has the weight's dims without the packed input dim; a rank-1 bias follows the last non-input dim
        bdim = min(dim, b.ndim - 1)
        return np.ascontiguousarray(np.take(b, np.arange(start, start + n), axis=bdim)), "narrow"

    def load_gguf_layer(self, prefix: str, shard: Shard | None = None) -> GgufLayer:
        """`GgufMatMul::from_uqff` (gguf/mod.rs:260-280)."""
        if self.serde_type(prefix) != SERDE_GGUF:
            raise ValueError(f"`{prefix}` is not a GGUF-quantized layer (format {self.serde_type(prefix)})")
        code = self.load_u32_scalar(f"{prefix}.weight.dtype")
        try:
            dt = GgmlDType.from_id(code)
        except Exception:
            raise ValueError(f"unknown dtype for quantized weight tensor {code}") from None
        dims = self.load_u32_vec(f"{prefix}.weight.shape")
        weight = self.load_raw_u8(f"{prefix}.weight")
        rng = shard_range(shard, dims)
        if rng is not None:
            dim, start, n = rng
            weight = slice_blocked_data(weight, dims, dt.block_size, dt.type_size, dim, start, n)
            dims[dim] = n
        bias, mode = self.load_bias(prefix, rng, len(dims))
        return GgufLayer(dt, tuple(dims), np.ascontiguousarray(weight), bias, mode)

    def load_hqq_layer(self, prefix: str, shard: Shard | None = None) -> HqqLayerData:
        """`HqqLayer::from_uqff` (hqq/mod.rs:797-826): HQQ artifacts load whole (no sharded loading), 0 optimization steps mean None."""
        if shard is not None and not (shard.offset is None and shard.world_size == 1):
            raise ValueError("HQQ UQFF artifacts do not support sharded loading.")
        if self.serde_type(prefix) != SERDE_HQQ:
            raise ValueError(f"`{prefix}` is not an HQQ layer (format {self.serde_type(prefix)})")
        bits = self.load_u8_scalar(f"{prefix}.weight.bits")
        if bits not in (8, 4, 3, 2, 1):
            raise ValueError(f"Unexpected value for HQQ bits: {bits}")
        group = self.load_u32_scalar(f"{prefix}.weight.group_size")
        if group == 0:
            raise ValueError("HQQ group_size must be non-zero")
        axis = self.load_u8_scalar(f"{prefix}.weight.axis")
        if axis not in (0, 1):
            raise ValueError(f"Unexpected value for HQQ axis: {axis}")
        steps = self.load_u32_scalar(f"{prefix}.weight.optimization_steps")
        key = f"{prefix}.bias"
        return HqqLayerData(self._get_torch(f"{prefix}.weight"), self._get_torch(f"{prefix}.weight.scales"), self._get_torch(f"{prefix}.weight.zeros"),
                            tuple(self.load_u32_vec(f"{prefix}.weight.shape")), bits, group, axis, steps or None,
                            self.load_u8_scalar(f"{prefix}.weight.round_zeros") != 0, self.load_u8_scalar(f"{prefix}.weight.channel_wise") != 0,
                            self._get_torch(key) if key in self._keys else None)

    def _get_torch(self, key: str):
        """A tensor in its stored dtype (bf16 has no numpy dtype): second handle on the same file, opened on first use."""
        if key not in self._keys:
            raise ValueError(f"Missing UQFF tensor `{key}`")
        if getattr(self, "_ft", None) is None:
            from safetensors import safe_open
            self._ft = safe_open(self._path, framework="pt")
        return self._ft.get_tensor(key)

    def isq_type(self, prefix: str) -> str:
        """`isq_type_from_uqff` of the layer's serde type: the GGML dtype name for GGUF layers, HQQ8 / HQQ4 for HQQ layers (hqq/mod.rs:1327-1336)."""
        t = self.serde_type(prefix)
        if t == SERDE_GGUF:
            return GgmlDType.from_id(self.load_u32_scalar(f"{prefix}.weight.dtype")).name
        if t == SERDE_HQQ:
            bits = self.load_u8_scalar(f"{prefix}.weight.bits")
            if bits in (8, 4):
                return f"HQQ{bits}"
            raise ValueError("Cannot convert HQQ bit width to an ISQ type.")
        raise ValueError(f"serde type {t} has no reader here")
