"""GGML dtype catalog + a device-resident packed tensor (the role of candle's `QTensor`).

Type ids, block sizes and type sizes: reference mistralrs-quant/src/gguf/archive.rs:73-160.
The packed bytes are the unmodified GGUF block bytes, row-major [N][K/blk].
"""
from __future__ import annotations

import enum
from dataclasses import dataclass

import torch


class GgmlDType(enum.Enum):
    #        id  block  bytes  tag
    F32 = (0, 1, 4, "f32")
    F16 = (1, 1, 2, "f16")
    Q4_0 = (2, 32, 18, "q4_0")
    Q4_1 = (3, 32, 20, "q4_1")
    Q5_0 = (6, 32, 22, "q5_0")
    Q5_1 = (7, 32, 24, "q5_1")
    Q8_0 = (8, 32, 34, "q8_0")
    Q8_1 = (9, 32, 36, "q8_1")
    Q2K = (10, 256, 84, "q2_k")
    Q3K = (11, 256, 110, "q3_k")
    Q4K = (12, 256, 144, "q4_k")
    Q5K = (13, 256, 176, "q5_k")
    Q6K = (14, 256, 210, "q6_k")
    Q8K = (15, 256, 292, "q8_k")
    BF16 = (30, 1, 2, "bf16")

    @property
    def id(self) -> int:
        return self.value[0]

    @property
    def block_size(self) -> int:
        return self.value[1]

    @property
    def type_size(self) -> int:
        return self.value[2]

    @property
    def tag(self) -> str:
        return self.value[3]

    @classmethod
    def from_id(cls, i: int) -> "GgmlDType":
        for d in cls:
            if d.id == i:
                return d
        raise ValueError(f"unsupported ggml dtype id {i}")

    def row_bytes(self, k: int) -> int:
        if k % self.block_size:
            raise ValueError(f"{self.name}: K={k} is not a multiple of the block size {self.block_size}")
        return k // self.block_size * self.type_size


@dataclass
class QTensor:
    """Packed quantized weight [N, K] living in HBM as raw GGUF blocks (uint8)."""
    dtype: GgmlDType
    shape: tuple  # (N, K) logical
    data: torch.Tensor  # uint8, N * row_bytes(K)

    def __post_init__(self):
        *lead, k = self.shape  # (N, K), or (experts, N, K) for stacked MoE experts (fast_mmq.grouped)
        n = 1
        for d in lead:
            n *= d
        want = n * self.dtype.row_bytes(k)
        if self.data.dtype != torch.uint8 or self.data.numel() != want:
            raise ValueError(f"QTensor: expected {want} packed bytes for {self.dtype.name} {self.shape}, "
                             f"got {self.data.numel()} ({self.data.dtype})")
        self.data = self.data.contiguous().view(-1)

    @property
    def device(self):
        return self.data.device

    def nbytes(self) -> int:
        return self.data.numel()

    @classmethod
    def from_numpy(cls, dtype: GgmlDType, shape, packed, device) -> "QTensor":
        t = torch.from_numpy(packed.reshape(-1)).to(device)
        return cls(dtype, tuple(shape), t)
