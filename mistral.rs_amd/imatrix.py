"""Importance-matrix collection for ISQ: host-side mirror of `mistralrs-quant/src/imatrix.rs`.

`ImatrixLayerStats` (imatrix.rs:13-173) accumulates, per input column of a layer, the sum of squares of the activations the layer sees during a
calibration run -- dense layers one vector `[in]`, stacked expert layers `[E, in]` scattered by the router's ids -- and `compute_imatrix` turns
them into the importance vector `mean square * ncalls` the weighted K-quant quantizers take.  `CollectedImatrixData` (imatrix.rs:175-232) is the
`.cimatrix` file of those vectors.  The accumulation runs on the device where the activations are (`mrs_imatrix_accumulate*`, csrc/ext_isq.hip);
the statistics never leave the GPU until `compute_imatrix`.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 30}


def _sym(name, argtypes):
    _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
    return _lib.sym("ext", name, argtypes, C.c_int)


class ImatrixLayerStats:
    """imatrix.rs:31-173.  `empty()` is the disabled state every layer starts in; `enable` / `enable_routed` start a collection."""

    def __init__(self):
        self._s = None

    @classmethod
    def empty(cls) -> "ImatrixLayerStats":
        return cls()

    def enable(self, in_dim: int, device) -> None:
        self._s = {"kind": "dense", "row_counts": 0, "ncalls": 0, "row_accum": torch.zeros(in_dim, dtype=torch.float32, device=device)}

    def enable_routed(self, num_experts: int, in_dim: int, device) -> None:
        self._s = {"kind": "routed", "ncalls": 0, "rows": 0, "counts": torch.zeros(num_experts, dtype=torch.float32, device=device),
                   "accum": torch.zeros(num_experts, in_dim, dtype=torch.float32, device=device)}

    def is_enabled(self) -> bool:
        return self._s is not None

    def snapshot(self):
        """(forward calls, token rows) so far; routed rows are token x top-k slots (imatrix.rs:64-71)."""
        if self._s is None:
            return None
        return (self._s["ncalls"], self._s["row_counts"] if self._s["kind"] == "dense" else self._s["rows"])

    def process(self, inp: torch.Tensor) -> None:
        """imatrix.rs:73-96: a plain forward through a routed layer contributes nothing; disabled stats are free."""
        s = self._s
        if s is None or s["kind"] != "dense":
            return
        if inp.dtype not in _DT:
            raise ValueError(f"imatrix: unsupported activation dtype {inp.dtype}")
        x = inp.reshape(-1, inp.shape[-1]).contiguous()
        if x.shape[1] != s["row_accum"].numel():
            raise ValueError(f"imatrix: activations have {x.shape[1]} columns, the layer {s['row_accum'].numel()}")
        s["ncalls"] += 1
        s["row_counts"] += x.shape[0]
        fn = _sym("mrs_imatrix_accumulate", [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p])
        if fn(x.data_ptr(), _DT[x.dtype], x.shape[0], x.shape[1], s["row_accum"].data_ptr(), torch.cuda.current_stream().cuda_stream) != 0:
            raise ValueError("imatrix: mrs_imatrix_accumulate refused the arguments")

    def process_routed(self, x: torch.Tensor, ids: torch.Tensor) -> None:
        """imatrix.rs:98-134: `ids` is (n, k); `x` is (n, in) -- the row scatters into all its experts -- or (n, k, in) -- slot (t, s) scatters into
        ids[t, s] only."""
        s = self._s
        if s is None or s["kind"] != "routed":
            return
        if ids.dim() != 2:
            raise ValueError("process_routed expects ids of rank 2")
        n, k = ids.shape
        if x.dim() not in (2, 3):
            raise ValueError(f"process_routed expects rank 2 or 3 input, got {x.dim()}")
        if x.dtype not in _DT:
            raise ValueError(f"imatrix: unsupported activation dtype {x.dtype}")
        in_dim = x.shape[-1]
        if in_dim != s["accum"].shape[1] or x.shape[0] != n or (x.dim() == 3 and x.shape[1] != k):
            raise ValueError("process_routed: activation / id shapes do not match the layer")
        xs = x.contiguous()
        ids32 = ids.reshape(-1).to(torch.int32).contiguous()
        fn = _sym("mrs_imatrix_accumulate_routed", [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_void_p])
        if fn(xs.data_ptr(), _DT[xs.dtype], ids32.data_ptr(), n, k, in_dim, int(x.dim() == 3), s["accum"].shape[0], s["accum"].data_ptr(),
              s["counts"].data_ptr(), torch.cuda.current_stream().cuda_stream) != 0:
            raise ValueError("imatrix: mrs_imatrix_accumulate_routed refused the arguments")
        s["rows"] += n * k
        s["ncalls"] += 1

    def compute_imatrix(self) -> torch.Tensor:
        """Dense: [in].  Routed: [E, in] with all-zero rows for zero-traffic experts (imatrix.rs:136-164)."""
        s = self._s
        if s is None:
            raise ValueError("Layer stats were deinitialized!")
        if s["kind"] == "dense":
            if s["row_counts"] == 0:
                raise ValueError("No activations were recorded for this layer.")
            return (s["row_accum"] / float(s["row_counts"])) * float(s["ncalls"])
        if float(s["counts"].sum()) == 0.0:
            raise ValueError("No activations were recorded for this layer.")
        safe = torch.clamp(s["counts"], min=1.0).unsqueeze(1)  # per-expert mean square; zero-traffic experts divide to zero, not NaN
        return (s["accum"] / safe) * float(s["ncalls"])

    def clear(self) -> None:
        self._s = None


class CollectedImatrixData(dict):
    """`.cimatrix`: u64 entry count, then per entry u64 key length, key bytes, u64 value count, f32 values, all little endian (imatrix.rs:175-232)."""

    def save_imatrix(self, fname) -> None:
        ext = os.path.splitext(str(fname))[1]
        if ext and ext != ".cimatrix":
            raise ValueError(f"Expected a .cimatrix file to save collected imatrix data to, got {ext[1:]!r}")
        buf = bytearray(struct.pack("<Q", len(self)))
        for key, data in self.items():
            kb = key.encode("utf-8")
            v = np.asarray(data, dtype="<f4").reshape(-1)
            buf += struct.pack("<Q", len(kb)) + kb + struct.pack("<Q", v.size) + v.tobytes()
        with open(fname, "wb") as f:
            f.write(bytes(buf))

    @classmethod
    def load_imatrix(cls, fname) -> "CollectedImatrixData":
        raw = open(fname, "rb").read()
        pos = 0

        def take(n):
            nonlocal pos
            if pos + n > len(raw):
                raise ValueError("cimatrix: unexpected end of file")
            out = raw[pos:pos + n]
            pos += n
            return out

        out = cls()
        (num,) = struct.unpack("<Q", take(8))
        for _ in range(num):
            (klen,) = struct.unpack("<Q", take(8))
            try:
                key = take(klen).decode("utf-8")
            except UnicodeDecodeError:
                raise ValueError("Invalid cimatrix key") from None
            (n,) = struct.unpack("<Q", take(8))
            out[key] = np.frombuffer(take(4 * n), dtype="<f4").astype(np.float32)
        return out
