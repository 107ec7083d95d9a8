"""GGUF container reader / writer: host-side mirror of `mistralrs-quant/src/gguf/archive.rs` (`GgufArchive::open` :351, dtype
catalog :73-160, 32-byte default alignment :25) and of the config synthesis + tensor bindings the GGUF loader performs
(`mistralrs-core/src/gguf/normal_config.rs:729-770,897-927`, `normal_bindings.rs:40-220`).

The reader memory-maps the file, validates the header (magic, version 2/3, little-endian), parses the metadata KV table and the
tensor table, and hands out zero-copy byte views; `load_llama` turns a llama/mistral-architecture file into a runner
(`mistralrs_amd.llama.Llama`) by uploading each tensor's raw blocks to HBM under its GGUF name.  The writer exists for tests and
for exporting synthetic checkpoints (the reference's tests build in-memory GGUF files the same way, archive.rs:1443-2114).
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass

import numpy as np

from .qtensor import GgmlDType

GGUF_MAGIC = b"GGUF"
DEFAULT_ALIGNMENT = 32
MAX_STRING_LENGTH = 1 << 30
MAX_ARRAY_ELEMENTS = 1 << 30
MAX_TENSOR_DIMS = 4

# metadata value types
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)
_SCALAR = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f", _BOOL: "<?", _U64: "<Q", _I64: "<q", _F64: "<d"}


class GgufError(ValueError):
    pass


@dataclass
class TensorInfo:
    name: str
    dtype: GgmlDType
    shape: tuple      # logical shape, outermost first (GGUF stores the dims innermost first)
    offset: int       # relative to the data section
    nbytes: int


class _Cursor:
    def __init__(self, buf):
        self.b, self.p = buf, 0

    def take(self, n: int) -> bytes:
        if self.p + n > len(self.b):
            raise GgufError("unexpected end of GGUF file (truncated header)")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def scalar(self, fmt: str):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))[0]

    def string(self) -> str:
        n = self.scalar("<Q")
        if n > MAX_STRING_LENGTH:
            raise GgufError(f"GGUF string length {n} exceeds the limit")
        return bytes(self.take(n)).decode("utf-8", errors="replace")

    def value(self, vt: int, depth: int = 0):
        if vt in _SCALAR:
            return self.scalar(_SCALAR[vt])
        if vt == _STR:
            return self.string()
        if vt == _ARR:
            if depth > 8:
                raise GgufError("GGUF metadata arrays nested too deeply")
            et, n = self.scalar("<I"), self.scalar("<Q")
            if n > MAX_ARRAY_ELEMENTS:
                raise GgufError(f"GGUF array length {n} exceeds the limit")
            if et in _SCALAR and et != _BOOL:
                return np.frombuffer(self.take(n * struct.calcsize(_SCALAR[et])), dtype=np.dtype(_SCALAR[et])).copy()
            return [self.value(et, depth + 1) for _ in range(n)]
        raise GgufError(f"unknown GGUF metadata value type {vt}")


def _nbytes(dt: GgmlDType, shape) -> int:
    n = 1
    for s in shape:
        n *= s
    if shape and shape[-1] % dt.block_size:
        raise GgufError(f"tensor row length {shape[-1]} is not a multiple of the {dt.name} block size {dt.block_size}")
    return n // dt.block_size * dt.type_size


class GgufArchive:
    """Parsed GGUF file: `.metadata` (dict), `.tensors` (name -> TensorInfo), `.tensor_bytes(name)` (zero-copy uint8 view)."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._f.close()
            raise GgufError("empty file is not a GGUF archive")
        c = _Cursor(memoryview(self._mm))
        if bytes(c.take(4)) != GGUF_MAGIC:
            raise GgufError("not a GGUF file (bad magic)")
        version = c.scalar("<I")
        if version in (0x01000000, 0x02000000, 0x03000000):
            raise GgufError("big-endian GGUF files are not supported")
        if version not in (2, 3):
            raise GgufError(f"unsupported GGUF version {version} (2 and 3 are supported)")
        self.version = version
        n_tensors, n_kv = c.scalar("<Q"), c.scalar("<Q")
        self.metadata: dict = {}
        for _ in range(n_kv):
            key = c.string()
            self.metadata[key] = c.value(c.scalar("<I"))
        self.alignment = int(self.metadata.get("general.alignment", DEFAULT_ALIGNMENT))
        if self.alignment <= 0 or self.alignment & (self.alignment - 1):
            raise GgufError(f"general.alignment {self.alignment} is not a power of two")
        self.tensors: dict[str, TensorInfo] = {}
        for _ in range(n_tensors):
            name = c.string()
            nd = c.scalar("<I")
            if nd > MAX_TENSOR_DIMS:
                raise GgufError(f"tensor {name}: {nd} dimensions exceed the limit of {MAX_TENSOR_DIMS}")
            dims = [c.scalar("<Q") for _ in range(nd)]
            tid, off = c.scalar("<I"), c.scalar("<Q")
            try:
                dt = GgmlDType.from_id(tid)
            except ValueError:
                raise GgufError(f"tensor {name}: unsupported ggml dtype {tid}")
            shape = tuple(reversed(dims))
            if off % self.alignment:
                raise GgufError(f"tensor {name}: offset {off} is not aligned to {self.alignment}")
            if name in self.tensors:
                raise GgufError(f"duplicate tensor name {name}")
            self.tensors[name] = TensorInfo(name, dt, shape, off, _nbytes(dt, shape))
        self.data_start = (c.p + self.alignment - 1) // self.alignment * self.alignment
        c.b.release()
        for t in self.tensors.values():
            if self.data_start + t.offset + t.nbytes > len(self._mm):
                raise GgufError(f"tensor {t.name}: data range exceeds the file size (truncated file)")

    def tensor_bytes(self, name: str) -> np.ndarray:
        t = self.tensors[name]
        return np.frombuffer(self._mm, dtype=np.uint8, count=t.nbytes, offset=self.data_start + t.offset)

    def close(self):
        try:
            self._mm.close()
        except BufferError:  # zero-copy tensor views are still alive: the map is released when they are collected
            pass
        finally:
            self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- config synthesis (normal_config.rs:729-770,897-927) ------------------------------------------------------------
    def llama_config(self, **overrides):
        from ..llama import LlamaConfig, RopeScaling
        md = self.metadata
        arch = md.get("general.architecture")
        if arch not in ("llama", "mistral"):
            raise GgufError(f"unsupported GGUF architecture {arch!r} (llama / mistral)")

        def req(k):
            if f"{arch}.{k}" not in md:
                raise GgufError(f"GGUF metadata is missing required key `{arch}.{k}`")
            return md[f"{arch}.{k}"]

        def req_uint(k, nonzero=False, uniform=False):
            """required_usize / required_uniform_usize / required_uniform_nonzero_usize (normal_config.rs:594-640,983-1002): `uniform` keys may be a
            per-layer array as long as every (non-zero, for `nonzero`) entry is the same value."""
            v = req(k)
            if uniform and isinstance(v, (list, tuple, np.ndarray)):
                vals = list(np.asarray(v).reshape(-1))
            else:
                vals = [v]
            if any(isinstance(x, (bool, np.bool_)) or not isinstance(x, (int, np.integer)) or int(x) < 0 for x in vals):
                raise GgufError(f"GGUF metadata `{arch}.{k}` must be a non-negative integer fitting this platform")
            vals = [int(x) for x in vals]
            if nonzero:
                vals = [x for x in vals if x != 0]
                if not vals:
                    raise GgufError(f"GGUF metadata `{arch}.{k}` has no non-zero value")
            if not vals:
                raise GgufError(f"GGUF metadata `{arch}.{k}` cannot be an empty array")
            if any(x != vals[0] for x in vals):
                raise GgufError(f"Native `{arch}` config cannot represent per-layer `{arch}.{k}` values {vals}; provide the original Hugging Face config "
                                "or use a compatible architecture path")
            return vals[0]
        # StandardFields::read_with_intermediate_size (gguf/normal_config.rs:897-925): every field below is REQUIRED there -- no silent defaults
        heads = req_uint("attention.head_count", uniform=True)
        d = req_uint("embedding_length", uniform=True)
        kv_heads = req_uint("attention.head_count_kv", nonzero=True, uniform=True)
        ctx_len = req_uint("context_length")
        eps = md.get(f"{arch}.attention.layer_norm_rms_epsilon", md.get(f"{arch}.attention.layer_norm_epsilon"))
        if eps is None:  # normal_config.rs:748-761
            raise GgufError(f"Standalone `{arch}` config requires `{arch}.attention.layer_norm_rms_epsilon` or `{arch}.attention.layer_norm_epsilon`")
        vocab = md.get(f"{arch}.vocab_size")
        if vocab is None:
            vocab = len(md["tokenizer.ggml.tokens"]) if "tokenizer.ggml.tokens" in md else self.tensors["token_embd.weight"].shape[0]
        # head_dim = attention.key_length when present, else embedding_length / head_count, which must divide (normal_config.rs:780-791);
        # rope.dimension_count is the ROTATED width (partial rotary), not the head size
        key_len = md.get(f"{arch}.attention.key_length")
        if key_len is not None:
            head_dim = int(key_len)
        else:
            if d % heads:
                raise GgufError(f"GGUF metadata: embedding length / attention head count ({d} / {heads}) is not an integer")
            head_dim = d // heads
        rope_dim = int(md.get(f"{arch}.rope.dimension_count", head_dim))
        if rope_dim != head_dim:
            raise GgufError(f"partial rotary embeddings ({arch}.rope.dimension_count = {rope_dim}, head size {head_dim}) are outside this runner")
        window = md.get(f"{arch}.attention.sliding_window")  # 0 = absent (normal_config.rs:792-796)
        window = int(window) if window else None
        scaling = None
        if md.get(f"{arch}.rope.scaling.type") == "linear":
            scaling = RopeScaling("linear", float(md.get(f"{arch}.rope.scaling.factor", 1.0)))
        kw = dict(hidden_size=d, intermediate_size=req_uint("feed_forward_length", uniform=True), num_layers=req_uint("block_count"), num_heads=heads,
                  num_kv_heads=kv_heads, vocab_size=int(vocab), head_dim=head_dim,
                  rms_eps=float(eps), rope_theta=float(md.get(f"{arch}.rope.freq_base", 10000.0)),
                  rope_scaling=scaling, rope_interleaved=True,  # GGUF llama/mistral: adjacent pairs (normal_registry.rs:446-461)
                  max_position_embeddings=ctx_len, sliding_window=window,
                  # Mixtral ships as architecture "llama" with expert_count / expert_used_count (gguf/normal_config.rs)
                  num_experts=int(md.get(f"{arch}.expert_count", 0) or 0), num_experts_per_tok=int(md.get(f"{arch}.expert_used_count", 2) or 2))
        kw.update(overrides)
        return LlamaConfig(**kw)


def load_llama(path: str, device, **cfg_overrides):
    """GGUF file -> runner.  Binds every tensor under its GGUF name (normal_bindings.rs:40-220): quantized 2-D weights stay packed,
    F32 norms go up as they are, `rope_freqs.weight` (Llama-3 frequency factors) feeds the RoPE table, a missing `output.weight`
    ties lm_head to the embeddings."""
    import torch
    from ..llama import Llama
    from .qtensor import QTensor
    ar = GgufArchive(path)
    cfg = ar.llama_config(**cfg_overrides)
    ff = ar.tensor_bytes("rope_freqs.weight").view(np.float32).copy() if "rope_freqs.weight" in ar.tensors else None
    m = Llama(cfg, device, freq_factors=ff)
    for name, t in ar.tensors.items():
        if name == "rope_freqs.weight":
            continue
        raw = torch.from_numpy(ar.tensor_bytes(name).copy())
        if t.dtype == GgmlDType.F32 and len(t.shape) == 1:
            m.set_tensor(name, raw.view(torch.float32))
        elif t.dtype == GgmlDType.F32 and len(t.shape) == 2:  # MoE router ffn_gate_inp [experts, hidden]
            m.set_tensor(name, raw.view(torch.float32).reshape(t.shape))
        elif len(t.shape) == 2:
            m.set_tensor(name, QTensor(t.dtype, t.shape, raw.to(device)))
        elif len(t.shape) == 3 and name.endswith("_exps.weight"):  # stacked experts [E, n, k] -> [E * n, k] (same bytes)
            m.set_tensor(name, QTensor(t.dtype, (t.shape[0] * t.shape[1], t.shape[2]), raw.to(device)))
        else:
            raise GgufError(f"tensor {name}: no binding for shape {t.shape} / {t.dtype.name}")
    if "output.weight" not in ar.tensors:
        e = ar.tensors["token_embd.weight"]
        m.set_tensor("output.weight", QTensor(e.dtype, e.shape, torch.from_numpy(ar.tensor_bytes("token_embd.weight").copy()).to(device)))
    ar.close()
    return m


# ---- writer (tests, synthetic checkpoints) -------------------------------------------------------------------------------
def _enc_str(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _enc_value(v) -> bytes:
    if isinstance(v, bool):
        return struct.pack("<I", _BOOL) + struct.pack("<?", v)
    if isinstance(v, int):
        return struct.pack("<I", _U32) + struct.pack("<I", v) if 0 <= v < 2 ** 32 else struct.pack("<I", _I64) + struct.pack("<q", v)
    if isinstance(v, float):
        return struct.pack("<I", _F32) + struct.pack("<f", v)
    if isinstance(v, str):
        return struct.pack("<I", _STR) + _enc_str(v)
    if isinstance(v, (list, tuple)) and all(isinstance(x, str) for x in v):
        return struct.pack("<I", _ARR) + struct.pack("<I", _STR) + struct.pack("<Q", len(v)) + b"".join(_enc_str(x) for x in v)
    if isinstance(v, np.ndarray) and v.dtype == np.float32:
        return struct.pack("<I", _ARR) + struct.pack("<I", _F32) + struct.pack("<Q", v.size) + v.tobytes()
    raise GgufError(f"cannot encode metadata value of type {type(v)}")


def write_gguf(path: str, metadata: dict, tensors: dict, alignment: int = DEFAULT_ALIGNMENT, version: int = 3) -> None:
    """tensors: name -> (GgmlDType, logical shape outermost-first, bytes-like of the packed data)."""
    md = dict(metadata)
    if alignment != DEFAULT_ALIGNMENT:
        md["general.alignment"] = alignment
    head = GGUF_MAGIC + struct.pack("<I", version) + struct.pack("<Q", len(tensors)) + struct.pack("<Q", len(md))
    for k, v in md.items():
        head += _enc_str(k) + _enc_value(v)
    off, infos, blobs = 0, b"", []
    for name, (dt, shape, data) in tensors.items():
        data = np.ascontiguousarray(np.frombuffer(bytes(data) if not isinstance(data, np.ndarray) else data.tobytes(), dtype=np.uint8))
        if data.size != _nbytes(dt, tuple(shape)):
            raise GgufError(f"tensor {name}: {data.size} bytes for {dt.name} {tuple(shape)}")
        infos += _enc_str(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", s) for s in reversed(shape)) + struct.pack("<I", dt.id) + struct.pack("<Q", off)
        blobs.append((off, data))
        off = (off + data.size + alignment - 1) // alignment * alignment
    hdr = head + infos
    pad = (-len(hdr)) % alignment
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * pad)
        pos = 0
        for o, data in blobs:
            f.write(b"\0" * (o - pos))
            f.write(data.tobytes())
            pos = o + data.size
