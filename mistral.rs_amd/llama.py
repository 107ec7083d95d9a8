"""Python handle on the C++ model runner (csrc/host/runtime.cpp) + the host-side pieces either side of it:
RoPE tables (mirror of mistralrs-core/src/layers.rs:1044-1182 `Llama3RotaryEmbedding`), paged-KV bookkeeping
for a fixed batch (slot mapping rule of pipeline/inputs_processor.rs:900-922), and a HIP-graph decode loop
(role of pipeline/cuda_graph.rs).  PyTorch only owns device buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .gguf.qtensor import GgmlDType, QTensor


@dataclass
class RopeScaling:  # Llama3RopeConfig (layers.rs:1030-1042)
    rope_type: str = "default"  # default | llama3 | linear
    factor: float = 1.0
    low_freq_factor: float | None = None
    high_freq_factor: float | None = None
    original_max_position_embeddings: int | None = None


@dataclass
class LlamaConfig:
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    vocab_size: int
    head_dim: int | None = None
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: RopeScaling | None = None
    rope_interleaved: bool = True  # GGUF llama/mistral: adjacent pairs (gguf/normal_registry.rs:446-461)
    max_position_embeddings: int = 8192
    block_size: int = 32  # paged_attention/mod.rs:44
    max_batch: int = 1
    max_context_len: int = 1024
    use_fused: bool = True
    # decode engine (csrc/ext_dec.hip: the reference CPU path's arithmetic, Q8_K / Q8_0 activations).  None = whenever the model allows it
    # (use_fused, interleaved RoPE, head_dim 128, block 32, q4_k / q5_k / q6_k / q8_0 linears); False = the round-1 fused kernels (Q8_1)
    decode_engine: bool | None = None
    kv_dtype: str = "bf16"  # "f16": the reference CPU path's default KV dtype (decode engine only; the MFMA prefill needs bf16 pages)
    tp_world_size: int = 1  # tensor parallel: num_heads / num_kv_heads / intermediate_size are the LOCAL (per-rank) sizes
    tp_rank: int = 0
    num_experts: int = 0           # > 0: Mixtral-style sparse MoE FFN in every layer (models/mixtral.rs:236-304)
    num_experts_per_tok: int = 2   # router top-k
    # Mistral: a query attends the last `sliding_window` positions, itself included (GGUF <arch>.attention.sliding_window, normal_config.rs:792-796;
    # mask rule paged_attention/layers/paged_attention.rs:551-553); None = full causal attention
    sliding_window: int | None = None

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_heads

    @property
    def max_blocks_per_seq(self) -> int:
        return (self.max_context_len + self.block_size - 1) // self.block_size + 1

    @classmethod
    def mixtral_8x7b(cls, **kw):
        return cls(hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8, vocab_size=32000,
                   rope_theta=1e6, num_experts=8, num_experts_per_tok=2, **kw)

    @classmethod
    def llama3_70b(cls, **kw):  # BASELINE.json configs[3]
        return cls(hidden_size=8192, intermediate_size=28672, num_layers=80, num_heads=64, num_kv_heads=8, vocab_size=128256,
                   rope_theta=500000.0, rope_scaling=RopeScaling("llama3", 8.0, 1.0, 4.0, 8192), **kw)

    @classmethod
    def llama3_8b(cls, **kw):
        return cls(hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8, vocab_size=128256,
                   rope_theta=500000.0, rope_scaling=RopeScaling("llama3", 8.0, 1.0, 4.0, 8192), **kw)


def rope_tables(cfg: LlamaConfig, freq_factors: np.ndarray | None = None):
    """cos/sin [max_pos, head_dim/2] f32, following new_llama3_with_factors (layers.rs:1071-1182):
    f32 inv_freq, t (f32) outer inv_freq (f32), then cos/sin."""
    hd = cfg.head_dim
    inv = (1.0 / np.power(np.float32(cfg.rope_theta), np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd))).astype(np.float32)
    rs = cfg.rope_scaling
    if freq_factors is not None:
        inv = (inv / np.asarray(freq_factors, dtype=np.float32)).astype(np.float32)
    elif rs is not None and rs.rope_type == "llama3":
        lo = np.float32(rs.original_max_position_embeddings) / np.float32(rs.low_freq_factor)
        hi = np.float32(rs.original_max_position_embeddings) / np.float32(rs.high_freq_factor)
        out = []
        for f in inv:
            wl = np.float32(2.0 * math.pi) / f
            if wl < hi:
                out.append(f)
            elif wl > lo:
                out.append(f / np.float32(rs.factor))
            else:
                sm = (np.float32(rs.original_max_position_embeddings) / wl - np.float32(rs.low_freq_factor)) / \
                     (np.float32(rs.high_freq_factor) - np.float32(rs.low_freq_factor))
                out.append((np.float32(1.0) - sm) * f / np.float32(rs.factor) + sm * f)
        inv = np.array(out, dtype=np.float32)
    elif rs is not None and rs.rope_type == "linear":
        inv = (inv / np.float32(rs.factor)).astype(np.float32)
    t = np.arange(cfg.max_position_embeddings, dtype=np.float32)[:, None]
    freqs = (t * inv[None, :]).astype(np.float32)
    return np.cos(freqs).astype(np.float32), np.sin(freqs).astype(np.float32)


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
                                         "vocab_size", "rot_dim", "rope_interleaved")] + [("rms_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("block_size", "max_blocks_per_seq", "max_batch", "max_context_len", "use_fused", "world_size", "rank",
                                         "num_experts", "num_experts_per_tok", "kv_f16", "sliding_window")]


class _PrefillArgs(C.Structure):
    _fields_ = [("token_ids", C.c_void_p), ("positions", C.c_void_p), ("slot_mapping", C.c_void_p), ("block_tables", C.c_void_p),
                ("context_lens", C.c_void_p), ("logits", C.c_void_p), ("start_pos", C.c_int32), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class _Bufs(C.Structure):
    _fields_ = [("input_ids", C.c_void_p), ("positions", C.c_void_p), ("context_lens", C.c_void_p), ("slot_mapping", C.c_void_p),
                ("block_tables", C.c_void_p), ("tokens_out", C.c_void_p), ("tokens_out_stride", C.c_int32), ("step_counter", C.c_void_p),
                ("cos_table", C.c_void_p), ("sin_table", C.c_void_p), ("logits", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class Llama:
    """Quantized Llama/Mistral decoder on one MI355X.  Weights are registered under their GGUF tensor names."""

    def __init__(self, cfg: LlamaConfig, device: torch.device, max_new_tokens: int = 4096, freq_factors=None):
        self.cfg, self.device = cfg, device
        L = _lib.load("ext")
        _lib.load("quant"); _lib.load("paged_attn"); _lib.load("core")
        self._L = L
        L.mrs_llama_create.restype = C.c_void_p
        L.mrs_llama_create.argtypes = [C.POINTER(_Cfg)]
        L.mrs_llama_workspace_bytes.restype = C.c_size_t
        L.mrs_llama_workspace_bytes.argtypes = [C.POINTER(_Cfg)]
        L.mrs_llama_destroy.argtypes = [C.c_void_p]
        L.mrs_llama_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64]
        L.mrs_llama_set_kv_cache.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mrs_llama_set_buffers.argtypes = [C.c_void_p, C.POINTER(_Bufs)]
        L.mrs_llama_decode_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mrs_llama_forward_logits.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mrs_llama_decode_bytes.restype = C.c_double
        L.mrs_llama_decode_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mrs_llama_prefill_workspace_bytes.restype = C.c_size_t
        L.mrs_llama_prefill_workspace_bytes.argtypes = [C.POINTER(_Cfg), C.c_int]
        L.mrs_llama_prefill.argtypes = [C.c_void_p, C.POINTER(_PrefillArgs), C.c_int, C.c_void_p]
        L.mrs_llama_prefill_flops.restype = C.c_double
        L.mrs_llama_prefill_flops.argtypes = [C.c_void_p, C.c_int]
        L.mrs_llama_set_comm.argtypes = [C.c_void_p, C.c_void_p]
        L.mrs_llama_set_dec_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.mrs_llama_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.mrs_dec_repack_bytes.restype = C.c_size_t
        L.mrs_dec_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
        L.mrs_dec_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
        L.mrs_gemm_qi_repack_bytes.restype = C.c_size_t
        L.mrs_gemm_qi_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
        L.mrs_gemm_qi_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
        L.mrs_llama_set_qi_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        self._exact_prefill_wanted = os.environ.get("MRS_PREFILL_EXACT", "1") not in ("", "0")
        # bf16 shadow copy of the dense linears for the SELECTABLE bf16 prompt path (csrc/ext_gemm_lt.hip: plain library GEMMs instead of the fused block dequant): 2 bytes
        # per weight.  MRS_PREFILL_BF16_SHADOW = 1 / 0 / auto (default): auto takes it when the copy stays below MRS_BF16_SHADOW_MAX_GB (24): Llama-3-8B yes (14 GB),
        # Llama-3-70B no (137 GB next to three other copies of the weights)
        d_, ff_ = cfg.hidden_size, cfg.intermediate_size
        shadow_bytes = 2.0 * cfg.num_layers * (2 * cfg.num_heads * cfg.head_dim * d_ + 2 * cfg.num_kv_heads * cfg.head_dim * d_ + 3 * ff_ * d_)
        want = os.environ.get("MRS_PREFILL_BF16_SHADOW", "auto")
        self._bf16_shadow_wanted = (want == "1" or (want == "auto" and shadow_bytes <= float(os.environ.get("MRS_BF16_SHADOW_MAX_GB", "24")) * 2 ** 30)) \
            and not cfg.num_experts and device.type == "cuda"
        L.mrs_dequantize.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.mrs_llama_set_bf16_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.mrs_last_error.restype = C.c_char_p
        c = _Cfg(cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim,
                 cfg.vocab_size, cfg.head_dim, int(cfg.rope_interleaved), cfg.rms_eps, cfg.block_size, cfg.max_blocks_per_seq,
                 cfg.max_batch, cfg.max_context_len, int(cfg.use_fused), cfg.tp_world_size, cfg.tp_rank, cfg.num_experts,
                 cfg.num_experts_per_tok if cfg.num_experts else 0, int(cfg.kv_dtype == "f16"), int(cfg.sliding_window or 0))
        self._c = c
        self._h = L.mrs_llama_create(C.byref(c))
        if not self._h:
            raise ValueError(self._err())
        self._keep: dict = {}
        if cfg.max_context_len > cfg.max_position_embeddings:
            raise ValueError(f"max_context_len {cfg.max_context_len} exceeds max_position_embeddings {cfg.max_position_embeddings} (the RoPE tables)")
        if cfg.kv_dtype not in ("bf16", "f16"):
            raise ValueError("kv_dtype must be bf16 or f16")
        # decode engine: wanted unless switched off; possible while every linear registered so far has a decode layout
        # rotate-half ("neox") RoPE (safetensors Llama / Mistral, RotaryEmbedding::forward with is_gpt_neox): the engine takes it with the q / k rows in pair
        # order (set_tensor permutes them for the decode layout), full rotary only
        self._engine_wanted = cfg.decode_engine is not False and cfg.use_fused and cfg.head_dim == 128 and cfg.block_size == 32 \
            and cfg.num_heads // cfg.num_kv_heads in (1, 2, 4, 8) and (cfg.rope_interleaved or getattr(cfg, "rot_dim", cfg.head_dim) == cfg.head_dim)
        if cfg.decode_engine and not self._engine_wanted:
            raise ValueError("decode_engine needs use_fused, head_dim 128, block_size 32, a GQA group of 1 / 2 / 4 / 8 and (with rotate-half RoPE) full rotary")
        if cfg.kv_dtype == "f16" and not self._engine_wanted:
            raise ValueError("f16 KV pages are a decode-engine feature")
        self._engine_ok = self._engine_wanted
        self._mode_set = None
        B, dev = cfg.max_batch, device
        # paged KV cache, bf16, reference layout (cache_engine.rs:458-484); blocks for max_batch full sequences
        self.num_blocks = B * cfg.max_blocks_per_seq
        x = 8
        kvt = torch.float16 if cfg.kv_dtype == "f16" else torch.bfloat16
        self.key_caches = [torch.zeros(self.num_blocks, cfg.num_kv_heads, cfg.head_dim // x, cfg.block_size, x, dtype=kvt, device=dev)
                           for _ in range(cfg.num_layers)]
        self.value_caches = [torch.zeros(self.num_blocks, cfg.num_kv_heads, cfg.head_dim, cfg.block_size, dtype=kvt, device=dev)
                             for _ in range(cfg.num_layers)]
        for i, (k, v) in enumerate(zip(self.key_caches, self.value_caches)):
            self._chk(L.mrs_llama_set_kv_cache(self._h, i, k.data_ptr(), v.data_ptr()))
        cos, sin = rope_tables(cfg, freq_factors)
        self.cos, self.sin = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
        self.input_ids = torch.zeros(B, dtype=torch.int32, device=dev)
        self.positions = torch.zeros(B, dtype=torch.int32, device=dev)
        self.context_lens = torch.zeros(B, dtype=torch.int32, device=dev)
        self.slot_mapping = torch.zeros(B, dtype=torch.int64, device=dev)
        # sequence s owns blocks [s*max_blocks, (s+1)*max_blocks)
        self.block_tables = (torch.arange(B * cfg.max_blocks_per_seq, dtype=torch.int32, device=dev)).reshape(B, cfg.max_blocks_per_seq).contiguous()
        self.tokens_out = torch.zeros(B, max_new_tokens, dtype=torch.int32, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.logits = torch.zeros(B, cfg.vocab_size, dtype=torch.float32, device=dev)
        self.workspace = torch.empty(L.mrs_llama_workspace_bytes(C.byref(c)), dtype=torch.uint8, device=dev)
        b = _Bufs(self.input_ids.data_ptr(), self.positions.data_ptr(), self.context_lens.data_ptr(), self.slot_mapping.data_ptr(),
                  self.block_tables.data_ptr(), self.tokens_out.data_ptr(), max_new_tokens, self.step_counter.data_ptr(),
                  self.cos.data_ptr(), self.sin.data_ptr(), self.logits.data_ptr(), self.workspace.data_ptr(), self.workspace.numel())
        self._chk(L.mrs_llama_set_buffers(self._h, C.byref(b)))
        self._graph = None
        self._replays_left = None
        self._graph_chained = False  # the captured step is the chained one (mrs_llama_decode_step_chained): it expects the hidden-state buffer to hold embedding(input_ids)
        self._need_embed = True      # ... which is stale whenever the host changed input_ids or ran an eager step since the last replay

    # -------------------------------------------------------------------------------------------------
    def _err(self) -> str:
        return (self._L.mrs_last_error() or b"").decode()

    def _chk(self, rc: int):
        if rc != 0:
            raise ValueError(self._err())

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.mrs_llama_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_comm(self, comm) -> None:
        """Attach the RCCL communicator (distributed.RcclComm) used for the row-parallel sum all-reduces."""
        self._comm = comm
        self._chk(self._L.mrs_llama_set_comm(self._h, comm.handle))

    def set_p2p(self, p2p) -> None:
        """Attach the one-shot peer-mailbox all-reduce (distributed.P2PAllReduce): decode-sized row-parallel all-reduces take it, larger ones RCCL.
        None detaches it (every all-reduce on RCCL again); a captured decode graph must be re-captured afterwards."""
        self._p2p = p2p
        self._L.mrs_llama_set_p2p.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self._L.mrs_llama_set_p2p(self._h, p2p.handle if p2p is not None else None))
        # a captured decode graph has the OLD route's kernels and its Comm struct baked in by value: replaying it after a detach would keep running the dead route
        # (sticky error state -> NaN sums -> greedy tokens from NaN logits, and nobody reads the error word any more).  Drop it; replay() raises until
        # capture_decode_graph() is called again.
        self._graph = None
        self._replays_left = None

    def p2p_error(self) -> int:
        """THIS rank's error word of the peer-mailbox route (blocking device read; call where the host synchronises anyway -- after a run of decode steps, before
        tokens are handed out): non-zero = some granule never arrived within the bounded spin (ext_p2p.hip) and the affected sums are NaN.  Read only: the word is
        per rank and the route must be dropped on EVERY rank in the same step -- use `p2p_sync_error()` unless you reduce the word over the ranks yourself."""
        if getattr(self, "_p2p", None) is None:
            return 0
        self._L.mrs_llama_check_p2p.argtypes = [C.c_void_p]
        return int(self._L.mrs_llama_check_p2p(self._h))

    def p2p_sync_error(self, group=None) -> bool:
        """The mandatory protocol around the error word (include/mrs_hip_ext.h, mrs_llama_check_p2p): MAX-reduce it over the tensor-parallel ranks; if any rank saw
        a time-out, EVERY rank detaches the route in this same call (RCCL from now on).  Returns True when the route was dropped: the caller re-captures its decode
        graph and repeats the steps since its last check.  The P2PAllReduce object stays alive (peers may still hold its mailbox mapped) until its close()."""
        if getattr(self, "_p2p", None) is None:
            return False
        import torch
        import torch.distributed as dist
        bad = torch.tensor([int(self.p2p_error() != 0)], device=getattr(self, "device", None) or getattr(self, "dev", "cuda"))
        if dist.is_available() and dist.is_initialized():
            if bad.device.type == "cpu" or dist.get_backend(group) == "gloo":
                bad = bad.cpu()
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if int(bad.item()):
            keep = self._p2p
            self.set_p2p(None)
            self._p2p_dropped = keep  # not freed here: close() after every rank has detached and re-captured
            return True
        return False

    def set_tensor(self, name: str, t) -> None:
        """t: QTensor (packed GGUF blocks) or an f32 torch tensor (norm weights)."""
        if isinstance(t, QTensor):
            self._keep[name] = t
            self._chk(self._L.mrs_llama_set_tensor(self._h, name.encode(), t.data.data_ptr(), t.dtype.id, t.shape[0], t.shape[1]))
            if self._engine_wanted and name != "token_embd.weight":
                # decode layout (mrs_dec_repack): same bits in row-major planes, made once; the GGUF blocks stay for the prefill GEMMs
                nb = self._L.mrs_dec_repack_bytes(t.dtype.id, t.shape[0], t.shape[1])
                if nb == 0:
                    if self.cfg.decode_engine:
                        raise ValueError(f"decode engine: {name} has ggml dtype {t.dtype.name}; supported: Q4K Q5K Q6K Q8_0")
                    self._engine_ok = False
                else:
                    planes = torch.empty(nb, dtype=torch.uint8, device=self.device)
                    src = t.data
                    if not self.cfg.rope_interleaved and name.endswith((".attn_q.weight", ".attn_k.weight")):
                        # pair order for mrs_dec_qkv_neox: inside every head the rows 0, hd/2, 1, hd/2 + 1, ... (rows are whole byte ranges of the packed tensor)
                        hd = self.cfg.head_dim
                        rows = t.data.view(t.shape[0] // hd, hd, -1)
                        order = torch.stack([torch.arange(hd // 2), torch.arange(hd // 2) + hd // 2], dim=1).reshape(-1).to(self.device)
                        src = rows[:, order, :].contiguous().view(-1)
                    self._chk(self._L.mrs_dec_repack(src.data_ptr(), t.dtype.id, t.shape[0], t.shape[1], planes.data_ptr(), self._stream()))
                    if src is not t.data:
                        torch.cuda.current_stream().synchronize()  # the permuted copy dies with this call
                    self._keep[name + "#dec"] = planes
                    self._chk(self._L.mrs_llama_set_dec_tensor(self._h, name.encode(), planes.data_ptr()))
            # MFMA-order copy for the exact-integer prompt GEMM (ext_gemm_qi.hip): dense per-layer linears of the types it takes; with every one present the
            # runner prefills in the decode engine's arithmetic (Llama::prefill_exact).  A third copy of the same bits, made once.  Batched decode steps (2..8 sequences)
            # stream THIS copy through the matrix cores (csrc/ext_dec_mm.hip), lm_head included when the model is sized for batches.
            if self._engine_wanted and self._exact_prefill_wanted and (name.startswith("blk.") or (name == "output.weight" and self.cfg.max_batch > 1)):  # dense linears and stacked expert tensors (round 6: sparse-MoE prompts too)
                nb2 = self._L.mrs_gemm_qi_repack_bytes(t.dtype.id, t.shape[0], t.shape[1])
                if nb2:
                    qi = torch.empty(nb2, dtype=torch.uint8, device=self.device)
                    self._chk(self._L.mrs_gemm_qi_repack(t.data.data_ptr(), t.dtype.id, t.shape[0], t.shape[1], qi.data_ptr(), self._stream()))
                    self._keep[name + "#qi"] = qi
                    self._chk(self._L.mrs_llama_set_qi_tensor(self._h, name.encode(), qi.data_ptr()))
            if self._bf16_shadow_wanted and name.startswith("blk.") and "_exps" not in name:
                self._make_bf16_shadow(name, t)
        else:
            t = t.to(self.device, torch.float32).contiguous()
            self._keep[name] = t
            rows, cols = (1, t.numel()) if t.dim() == 1 else t.shape
            self._chk(self._L.mrs_llama_set_tensor(self._h, name.encode(), t.data_ptr(), 0, rows, cols))

    def _make_bf16_shadow(self, name: str, t) -> None:
        # q / k / v and gate / up of a layer share one buffer, rows back to back: the runner then multiplies them in ONE library GEMM (fused_qkv / fused_glu role)
        layer, role = name.split(".")[1], name.split(".")[2]
        cfg = self.cfg
        nq, nkv, ff, d = cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim, cfg.intermediate_size, cfg.hidden_size
        group = {"attn_q": ("qkv", 0, nq + 2 * nkv), "attn_k": ("qkv", nq, nq + 2 * nkv), "attn_v": ("qkv", nq + nkv, nq + 2 * nkv),
                 "ffn_gate": ("gu", 0, 2 * ff), "ffn_up": ("gu", ff, 2 * ff)}.get(role)
        if group is not None and t.shape[1] == d:
            key = f"blk.{layer}.{group[0]}#bf16"
            if key not in self._keep:
                self._keep[key] = torch.empty(group[2], d, dtype=torch.bfloat16, device=self.device)
            sh = self._keep[key][group[1]: group[1] + t.shape[0]]
        else:
            sh = torch.empty(t.shape[0], t.shape[1], dtype=torch.bfloat16, device=self.device)
            self._keep[name + "#bf16"] = sh
        self._chk(self._L.mrs_dequantize(t.data.data_ptr(), t.dtype.id, t.shape[0], t.shape[1], sh.data_ptr(), 30, self._stream()))
        self._chk(self._L.mrs_llama_set_bf16_tensor(self._h, name.encode(), sh.data_ptr()))

    def build_bf16_shadow(self) -> bool:
        """Make the bf16 shadow copy of every dense linear AFTER loading (a model that was built without it: `MRS_PREFILL_BF16_SHADOW=auto` skips models whose copy exceeds
        MRS_BF16_SHADOW_MAX_GB).  Raises torch.OutOfMemoryError when it does not fit; returns `bf16_shadow`."""
        if self.cfg.num_experts:
            return False
        for name, t in list(self._keep.items()):
            if isinstance(t, QTensor) and name.startswith("blk.") and "#" not in name and "_exps" not in name:
                self._make_bf16_shadow(name, t)
        torch.cuda.current_stream().synchronize()
        return self.bf16_shadow

    def drop_bf16_shadow(self) -> None:
        """Release the shadow copies (the runner falls back to the fused block-dequant kernels on the bf16 path)."""
        self._L.mrs_llama_set_bf16_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        for name in [n for n in self._keep if isinstance(self._keep[n], QTensor) and n.startswith("blk.") and "_exps" not in n]:
            self._L.mrs_llama_set_bf16_tensor(self._h, name.encode(), None)
        for k in [k for k in self._keep if k.endswith("#bf16")]:
            del self._keep[k]

    def decode_bytes(self, b: int, context_len: int) -> float:
        return float(self._L.mrs_llama_decode_bytes(self._h, b, context_len))

    def _stream(self) -> int:
        return torch.cuda.current_stream().cuda_stream

    def set_state(self, token_ids, positions) -> None:
        """Point the decode state at `token_ids[s]` to be processed at `positions[s]` (context = position + 1)."""
        b = len(token_ids)
        cfg = self.cfg
        pos = np.asarray(positions, dtype=np.int64)
        if pos.max() >= cfg.max_context_len:
            raise ValueError("position beyond max_context_len")
        bt = self.block_tables[:b].cpu().numpy()
        slots = bt[np.arange(b), pos // cfg.block_size].astype(np.int64) * cfg.block_size + pos % cfg.block_size
        self.input_ids[:b] = torch.tensor(np.asarray(token_ids, dtype=np.int32), device=self.device)
        self.positions[:b] = torch.tensor(pos.astype(np.int32), device=self.device)
        self.context_lens[:b] = torch.tensor((pos + 1).astype(np.int32), device=self.device)
        self.slot_mapping[:b] = torch.tensor(slots, device=self.device)
        self._replays_left = None  # recomputed from the device state at the next replay()
        self._need_embed = True

    @property
    def decode_path(self) -> str:
        """'engine' (ext_dec.hip, reference CPU-path arithmetic), 'fused' (round-1 kernels, Q8_1) or 'reference-sequence'."""
        return "engine" if (self._engine_wanted and self._engine_ok) else ("fused" if self.cfg.use_fused else "reference-sequence")

    def _set_mode(self) -> None:
        mode = 2 if (self._engine_wanted and self._engine_ok) else int(bool(self.cfg.use_fused))
        if mode != self._mode_set:
            if self.cfg.kv_dtype == "f16" and mode != 2:
                raise ValueError("f16 KV pages need the decode engine (a linear has a dtype it does not support)")
            self._chk(self._L.mrs_llama_set_mode(self._h, mode))
            self._mode_set = mode

    def forward_logits(self, b: int) -> torch.Tensor:
        self._set_mode()
        self._need_embed = True  # the eager step overwrites the hidden-state buffer
        self._chk(self._L.mrs_llama_forward_logits(self._h, b, self._stream()))
        return self.logits[:b]

    def decode_step(self, b: int = 1) -> None:
        self._set_mode()
        self._need_embed = True
        self._chk(self._L.mrs_llama_decode_step(self._h, b, self._stream()))

    def _chained_ok(self, b: int) -> bool:
        """Round 6: a greedy batch-1 step of the decode engine can run "chained" -- the next token's embedding row is gathered by the launch that samples it and lm_head folds
        the arg-max into its epilogue (two launches fewer per step).  The graph then relies on the hidden-state buffer between replays: replay() refreshes it after any
        host-side change of the state."""
        self._set_mode()
        self._L.mrs_llama_chained_ok.argtypes = [C.c_void_p, C.c_int]
        return bool(self._L.mrs_llama_chained_ok(self._h, b))

    def _step_for_graph(self, b: int) -> None:
        if self._graph_chained:
            self._L.mrs_llama_decode_step_chained.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
            self._chk(self._L.mrs_llama_decode_step_chained(self._h, b, self._stream()))
        else:
            self.decode_step(b)

    def _embed_state(self, b: int) -> None:
        self._L.mrs_llama_embed_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self._L.mrs_llama_embed_state(self._h, b, self._stream()))

    def capture_decode_graph(self, b: int = 1) -> None:
        """Capture one decode step (forward + greedy sample + state advance) into a HIP graph."""
        self._graph_batch = b
        self._graph_chained = self._chained_ok(b)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up outside capture (lazy module loads, attribute sets)
            saved = [t.clone() for t in (self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.step_counter)]
            if self._graph_chained:
                self._embed_state(b)
            self._step_for_graph(b)
            torch.cuda.current_stream().synchronize()
            for t, v in zip((self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.step_counter), saved):
                t.copy_(v)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_for_graph(b)
        self._graph = g
        for t, v in zip((self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.step_counter), saved):
            t.copy_(v)
        self._need_embed = True  # the restored input_ids are not what the warm-up / capture left in the hidden-state buffer

    def replay(self) -> None:
        if self._graph is None:
            raise RuntimeError("replay(): no decode graph is captured (set_p2p() / p2p_sync_error() drop the graph of the old all-reduce route: capture_decode_graph() again)")
        # the captured step advances device-side state; refuse to run it past the buffers it indexes (tokens_out, block table, RoPE tables)
        if self._replays_left is None:  # one read-back per set_state(): replays that stay inside the context window and the token buffer
            b = getattr(self, "_graph_batch", None) or self.positions.numel()  # only the sequences of the captured batch count (stale entries of earlier, larger batches do not)
            self._replays_left = int(min(self.cfg.max_context_len - 1 - int(self.positions[:b].max().item()), self.tokens_out.shape[1] - int(self.step_counter.item())))
        if self._replays_left <= 0:
            raise ValueError("replay(): the decode graph would run past max_new_tokens / max_context_len; set_state() to a new position first")
        self._replays_left -= 1
        if self._graph_chained and self._need_embed:
            self._embed_state(self._graph_batch)  # eager, on the current stream, ahead of the graph: the chained step starts from embedding(input_ids)
            self._need_embed = False
        self._graph.replay()

    def prefill(self, tokens, start_pos: int = 0, seq: int = 0) -> torch.Tensor:
        """Prompt processing of one sequence (mrs_llama_prefill): returns the last token's logits [vocab] and leaves K/V of every prompt token in the
        sequence's pages.  With the decode engine and Q4_K / Q5_K / Q6_K linears (`prefill_is_exact`) the prompt runs in the decode engine's arithmetic -- Q8_K
        activation rows x exact-integer MFMA GEMMs, the decode kernels' attention per query -- and every logit and KV page equals what a token-by-token
        decode produces; otherwise on the bf16 matrix cores (dequantized weights).  Mirrors the prompt branch of the reference
        (PagedAttention::forward try_regular_prompt + reshape_and_cache, paged_attention.rs:1413-1475)."""
        self._set_mode()
        cfg, dev = self.cfg, self.device
        T = len(tokens)
        if cfg.kv_dtype != "bf16" and not self.prefill_is_exact:
            raise ValueError("the bf16 MFMA prefill reads and writes bf16 KV pages; use prefill_chunked() with f16 pages")
        if start_pos + T > cfg.max_context_len:
            raise ValueError("prompt does not fit max_context_len")
        pos = torch.arange(start_pos, start_pos + T, dtype=torch.int32, device=dev)
        bt_row = self.block_tables[seq].to(torch.int64)
        slots = (bt_row[(pos // cfg.block_size).long()] * cfg.block_size + (pos % cfg.block_size).long()).to(torch.int64)
        bts = self.block_tables[seq:seq + 1].expand(T, -1).contiguous()
        ctx = (pos + 1).to(torch.int32)
        ids = torch.tensor(np.asarray(tokens, dtype=np.int32), device=dev)
        need = self._L.mrs_llama_prefill_workspace_bytes(C.byref(self._c), T)
        if getattr(self, "_prefill_ws", None) is None or self._prefill_ws.numel() < need:
            self._prefill_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(cfg.vocab_size, dtype=torch.float32, device=dev)
        a = _PrefillArgs(ids.data_ptr(), pos.data_ptr(), slots.data_ptr(), bts.data_ptr(), ctx.data_ptr(), out.data_ptr(), start_pos,
                         self._prefill_ws.data_ptr(), self._prefill_ws.numel())
        self._chk(self._L.mrs_llama_prefill(self._h, C.byref(a), T, self._stream()))
        self._prefill_keep = (ids, pos, slots, bts, ctx)  # keep alive until the stream has consumed them
        return out

    @property
    def prefill_is_exact(self) -> bool:
        """True when prefill() runs in the decode engine's arithmetic (csrc/ext_gemm_qi.hip; Llama::prefill_exact in host/runtime.cpp)."""
        self._set_mode()
        self._L.mrs_llama_prefill_is_exact.argtypes = [C.c_void_p]
        return bool(self._L.mrs_llama_prefill_is_exact(self._h))

    def set_prefill_mode(self, exact: int) -> None:
        """1: prompts in the decode engine's arithmetic (default where available), 0: bf16-operand MFMA GEMMs + flash attention (plain library GEMMs on the bf16 shadow
        copy of the weights when the model has one, else the fused block-dequant kernels), 2: as 0 with the fused block-dequant kernels forced, -1: MRS_PREFILL_EXACT."""
        self._L.mrs_llama_set_prefill_mode.argtypes = [C.c_void_p, C.c_int]
        self._chk(self._L.mrs_llama_set_prefill_mode(self._h, int(exact)))

    @property
    def bf16_shadow(self) -> bool:
        """True when every dense linear has a bf16 shadow copy: set_prefill_mode(0) then runs the prompt GEMMs as plain bf16 library GEMMs (csrc/ext_gemm_lt.hip)."""
        self._L.mrs_llama_bf16_shadow_ok.argtypes = [C.c_void_p]
        return bool(self._L.mrs_llama_bf16_shadow_ok(self._h))

    def prefill_flops(self, T: int) -> float:
        return float(self._L.mrs_llama_prefill_flops(self._h, T))

    # chunked prefill through the batch<=8 decode kernels: the chunk's tokens act as `b` sequences that share one
    # block table, each attending to [0, its position] (K/V of the whole chunk are in the cache before attention).
    def prefill_chunked(self, tokens, start_pos: int = 0, chunk: int = 8) -> torch.Tensor:
        cfg = self.cfg
        chunk = min(chunk, cfg.max_batch, 8)
        saved_bt = self.block_tables.clone()
        self.block_tables[:] = self.block_tables[0:1].expand_as(self.block_tables)
        last = None
        for i in range(0, len(tokens), chunk):
            ids = tokens[i:i + chunk]
            self.set_state(ids, [start_pos + i + j for j in range(len(ids))])
            last = self.forward_logits(len(ids))[len(ids) - 1].clone()
        self.block_tables.copy_(saved_bt)
        return last


def random_qtensor(dtype: GgmlDType, n: int, k: int, device, seed: int, w_std: float = 0.02) -> QTensor:
    """Random-but-valid GGUF blocks generated ON THE GPU (no CPU quantizer, no oracle): random quants and
    sub-scales, block super-scales chosen so the dequantized weights are ~zero-mean with std ~ w_std.
    Used for synthetic benchmark models (there is no network for real checkpoints)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ts, blk = dtype.type_size, dtype.block_size
    nb = n * (k // blk)
    raw = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=device, generator=g)
    half = raw.view(torch.int16).view(nb, ts // 2)

    def f16bits(val: float, jitter: bool = True):
        v = torch.full((nb,), val, dtype=torch.float32, device=device)
        if jitter:
            v = v * (0.75 + 0.5 * torch.rand(nb, device=device, generator=g))
        return v.to(torch.float16).view(torch.int16)
    if dtype == GgmlDType.Q4K or dtype == GgmlDType.Q5K:
        qmax = 15 if dtype == GgmlDType.Q4K else 31
        d = w_std / (31.5 * qmax * 0.36)
        half[:, 0] = f16bits(d)
        half[:, 1] = f16bits(d * qmax / 2.0)
    elif dtype == GgmlDType.Q6K:
        half[:, 104] = f16bits(w_std / (74.0 * 18.5))
    elif dtype == GgmlDType.Q8_0:
        half[:, 0] = f16bits(w_std / 74.0)
    elif dtype in (GgmlDType.Q4_0, GgmlDType.Q5_0):
        half[:, 0] = f16bits(w_std / (4.6 if dtype == GgmlDType.Q4_0 else 9.2))
    else:
        raise ValueError(f"random_qtensor: {dtype.name} not implemented")
    return QTensor(dtype, (n, k), raw.view(-1))
