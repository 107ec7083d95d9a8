"""oracle/llama_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the Llama / Mistral decoder graph as the reference drives it
(mistralrs-core/src/models/llama.rs:68-157 CausalSelfAttention, :243-260 Block, :487-518 forward_embeds),
token by token with an eager KV cache.  `mode` selects the arithmetic:
  "q8_1"  the GPU path's dataflow: activations quantized to Q8_1 (mmvq_gguf.cu), integer dots (oracle C); f64 glue ops
  "exact" dequantized weights, f64 accumulation (oracle A); f64 glue ops
  "cpu"   the reference CPU path: candle QMatMul = Q8_K / Q8_0 activations with ggml's generic 8-lane f32 order (oracle B; parity unpinned),
          candle's rms_norm (f32 sums in order, x / sqrt(mean + eps) * w), the in-tree CPU attention (single_q.rs + fast_exp,
          cpu_path_oracle.c), SiLU with libm exp
  "cpu_fast"  as "cpu", but every GEMV row = one f32 term per superblock added in superblock order (llama_oracle.c; OpenMP rows: the
          cpu_baseline kernel) -- a second CPU evaluation of the same path, used to measure what f32 summation order alone does
  "engine"    the decode engine's orders: GEMV = the same integers and products in the kernel's summation order (orc_gemv_engine), and the SAME rms_norm / attention / SiLU expressions with the summation trees of
          the HIP kernels and the reference's fast_exp wherever an exponential is taken (cpu_path_oracle.c orc_*_engine): the HIP engine must
          equal this mode bit for bit
Everything else is f32 like the CPU path: interleaved/neox RoPE, residual adds.
`kv_dtype` in {"f32", "bf16", "f16"} rounds K/V on the way into the cache; `attn_bpw` = 32-token blocks per attention split ("engine").
"""
from __future__ import annotations

import numpy as np

from . import oracle as O


def _round(x, dt):
    if dt == "bf16":
        return O.round_bf16(x)
    if dt == "f16":
        return x.astype(np.float16).astype(np.float32)
    return x


def p_moe_free(w: dict) -> bool:
    return not any(n.endswith("ffn_gate_inp.weight") for n in w)


class LlamaRef:
    def __init__(self, cfg, weights: dict, cos: np.ndarray, sin: np.ndarray, mode: str = "q8_1", kv_dtype: str = "bf16", attn_bpw: int = 1,
                 n_kv_chunks: int = 1):
        """cfg: object with hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads, head_dim, vocab_size,
        rms_eps, rope_interleaved.  weights: GGUF name -> (ggml_type, packed uint8 [N, row_bytes]) or f32 array."""
        self.cfg, self.w, self.cos, self.sin, self.mode, self.kv_dtype = cfg, weights, cos, sin, mode, kv_dtype
        self.attn_bpw, self.n_kv_chunks = attn_bpw, n_kv_chunks
        self.k = [[] for _ in range(cfg.num_layers)]
        self.v = [[] for _ in range(cfg.num_layers)]
        self.trace = None  # a list: every linear() appends (tensor name, the int8 activation quants its matvec multiplies by) -- bench.py's first-flip measurement

    def linear(self, name: str, x: np.ndarray) -> np.ndarray:
        t, packed = self.w[name]
        n = packed.shape[0]
        k = x.shape[-1]
        if self.trace is not None and self.mode in ("cpu", "cpu_fast", "engine"):
            xq = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, k)
            if t == O.Q8_0:  # Q8_0 weights take Q8_0 activation blocks (34 B: f16 d + 32 int8)
                img = O.quantize(O.Q8_0, xq).reshape(xq.shape[0], -1, 34)[:, :, 2:]
            else:            # K-quants take Q8_K (292 B: f32 d + 256 int8 + 16 int16 sums)
                img = O.quantize_q8_K(xq).reshape(xq.shape[0], -1, 292)[:, :, 4:260]
            self.trace.append((name, img.copy()))
        if self.mode == "q8_1":
            return O.matmul_q8_1(t, packed, n, k, O.quantize_q8_1(x.reshape(-1, k)))
        if self.mode == "cpu":
            return O.matmul_cpu(t, packed, n, k, x.reshape(-1, k))
        if self.mode == "cpu_fast":  # one f32 term per superblock, superblock order (bench.py cpu_baseline)
            return np.concatenate([O.gemv_cpu_fast(t, packed, n, k, r) for r in x.reshape(-1, k)], axis=0)
        if self.mode == "engine":  # the decode engine's summation order
            return np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in x.reshape(-1, k)], axis=0)
        return O.matmul_exact(t, packed, n, k, x.reshape(-1, k))

    # ---- glue ops by mode
    def norm(self, x, w):
        if self.mode == "engine":
            return O.rms_norm_engine(x, w, self.cfg.rms_eps)
        if self.mode in ("cpu", "cpu_fast"):
            return O.rms_norm_candle(x, w, self.cfg.rms_eps)
        return O.rms_norm(x, w, self.cfg.rms_eps)

    def attend(self, q, k, v, scale):
        """q [1, H, hd]; k, v [S, KVH, hd]"""
        w = getattr(self.cfg, "sliding_window", None) or 0
        if self.mode == "engine":
            return O.attention_engine(q[0], k, v, scale, self.attn_bpw, w)[None]
        if w and k.shape[0] > w:  # the reference gathers the last W positions and runs its SDPA over them (DecodePlan::GatherSdpa, paged_attention/plan.rs:116-138)
            k, v = k[-w:], v[-w:]
        if self.mode in ("cpu", "cpu_fast"):
            return O.attention_single_q_cpu(q[0], k, v, scale, self.n_kv_chunks)[None]
        return O.attention(q, k, v, scale)

    def glu(self, g, u):
        return O.fused_glu_engine(g, u) if self.mode == "engine" else O.fused_glu(g, u, 0)

    def embed(self, ids) -> np.ndarray:
        t, packed = self.w["token_embd.weight"]
        return O.dequantize(t, packed[np.asarray(ids)], self.cfg.hidden_size)

    def step(self, token: int, pos: int) -> np.ndarray:
        """Process one token at position `pos` (the cache must hold positions < pos). Returns logits [vocab]."""
        c = self.cfg
        hd, H, KVH = c.head_dim, c.num_heads, c.num_kv_heads
        h = self.embed([token]).astype(np.float32)  # [1, d]
        for l in range(c.num_layers):
            p = f"blk.{l}."
            xn = self.norm(h, self.w[p + "attn_norm.weight"])
            q = self.linear(p + "attn_q.weight", xn).reshape(1, H, hd)
            k = self.linear(p + "attn_k.weight", xn).reshape(1, KVH, hd)
            v = self.linear(p + "attn_v.weight", xn).reshape(1, KVH, hd)
            posv = np.array([pos], dtype=np.int32)
            q = O.rope(q, self.cos, self.sin, posv, not c.rope_interleaved)
            k = O.rope(k, self.cos, self.sin, posv, not c.rope_interleaved)
            assert len(self.k[l]) == pos, "cache out of sync with position"
            self.k[l].append(_round(k[0], self.kv_dtype))
            self.v[l].append(_round(v[0], self.kv_dtype))
            att = self.attend(q, np.stack(self.k[l]), np.stack(self.v[l]), np.float32(1.0 / np.sqrt(np.float32(hd))))
            h = h + self.linear(p + "attn_output.weight", att.reshape(1, H * hd))
            xn = self.norm(h, self.w[p + "ffn_norm.weight"])
            if p + "ffn_gate_inp.weight" in self.w:
                h = h + self.moe(p, xn)
                continue
            g = self.linear(p + "ffn_gate.weight", xn)
            u = self.linear(p + "ffn_up.weight", xn)
            h = h + self.linear(p + "ffn_down.weight", self.glu(g, u))
        xn = self.norm(h, self.w["output_norm.weight"])
        return self.linear("output.weight", xn)[0]

    def prefill(self, tokens, start_pos: int = 0) -> np.ndarray:
        """The whole prompt in one pass, layer by layer (models/llama.rs:487-518 with seq_len = T): every linear is the same per-row arithmetic as
        `step` (candle QMatMul quantizes each activation row on its own: one Q8_K / Q8_0 image per token), RoPE / residuals per row, K / V of all T
        positions enter the cache, then attention:
          cpu / cpu_fast  attention/backends/cpu/full.rs (q_len > 1: Q_BLOCK x KV_BLOCK tiles, causal rows as binary ranges; cpu_path_oracle.c
                          orc_attention_full_cpu) -- NOT the single_q.rs order `step` uses, so prefill != step in the last bits on a CPU too;
          engine          per query the decode attention over positions <= its own (csrc/ext_gemm_qi.hip prefill_attn_exact_kernel = the decode
                          kernels' split / merge per query): prefill == the step loop bit for bit, by construction;
          others          the f64 reference attention per query.
        Returns the logits of the LAST position [vocab] (the reference only projects the last row: llama.rs:513-516)."""
        c = self.cfg
        hd, H, KVH = c.head_dim, c.num_heads, c.num_kv_heads
        T = len(tokens)
        assert all(len(kl) == start_pos for kl in self.k), "cache out of sync with start_pos"
        assert p_moe_free(self.w), "prefill(): dense models only"
        h = self.embed([int(t) for t in tokens]).astype(np.float32)  # [T, d]
        posv = np.arange(start_pos, start_pos + T, dtype=np.int32)
        scale = np.float32(1.0 / np.sqrt(np.float32(hd)))
        w = getattr(c, "sliding_window", None) or 0
        for l in range(c.num_layers):
            p = f"blk.{l}."
            xn = self.norm(h, self.w[p + "attn_norm.weight"])
            q = self.linear(p + "attn_q.weight", xn).reshape(T, H, hd)
            k = self.linear(p + "attn_k.weight", xn).reshape(T, KVH, hd)
            v = self.linear(p + "attn_v.weight", xn).reshape(T, KVH, hd)
            q = O.rope(q, self.cos, self.sin, posv, not c.rope_interleaved)
            k = O.rope(k, self.cos, self.sin, posv, not c.rope_interleaved)
            for t in range(T):
                self.k[l].append(_round(k[t], self.kv_dtype))
                self.v[l].append(_round(v[t], self.kv_dtype))
            K, V = np.stack(self.k[l]), np.stack(self.v[l])
            if self.mode in ("cpu", "cpu_fast"):
                att = O.attention_full_cpu(q, K, V, scale, w)
            else:
                att = np.concatenate([self.attend(q[t:t + 1], K[: start_pos + t + 1], V[: start_pos + t + 1], scale) for t in range(T)], axis=0)
            h = h + self.linear(p + "attn_output.weight", att.reshape(T, H * hd))
            xn = self.norm(h, self.w[p + "ffn_norm.weight"])
            g = self.linear(p + "ffn_gate.weight", xn)
            u = self.linear(p + "ffn_up.weight", xn)
            h = h + self.linear(p + "ffn_down.weight", self.glu(g, u))
        xn = self.norm(h[-1:], self.w["output_norm.weight"])
        return self.linear("output.weight", xn)[0]

    def moe(self, p: str, xn: np.ndarray) -> np.ndarray:
        """SparseMoeBlock::forward (models/mixtral.rs:280-304) for one token: router logits = xn @ Wr^T (f32), softmax over all experts,
        top-k (ties: lower index first), renormalise (moe_router_topk, ops.rs:259-336); y = sum_j w_j * down_j(silu(gate_j xn) * up_j xn)."""
        c = self.cfg
        logits = (xn.astype(np.float32) @ self.w[p + "ffn_gate_inp.weight"].astype(np.float32).T)[0].astype(np.float64)
        pr = np.exp(logits - logits.max())
        pr /= pr.sum()
        ids = sorted(range(len(pr)), key=lambda e: (-pr[e], e))[: c.num_experts_per_tok]
        wts = pr[ids] / pr[ids].sum()
        self.last_route = (ids, wts)
        out = np.zeros_like(xn, dtype=np.float64)
        ff = c.intermediate_size
        for e, wt in zip(ids, wts):
            tg, pg = self.w[p + "ffn_gate_exps.weight"]
            tu, pu = self.w[p + "ffn_up_exps.weight"]
            td, pd = self.w[p + "ffn_down_exps.weight"]
            self.w["_g"], self.w["_u"], self.w["_d"] = (tg, pg[e * ff:(e + 1) * ff]), (tu, pu[e * ff:(e + 1) * ff]), (td, pd[e * c.hidden_size:(e + 1) * c.hidden_size])
            act = self.glu(self.linear("_g", xn), self.linear("_u", xn))
            out += np.float64(np.float32(wt)) * self.linear("_d", act).astype(np.float64)
        return out.astype(np.float32)

    def run(self, tokens, start_pos: int = 0) -> np.ndarray:
        return np.stack([self.step(int(t), start_pos + i) for i, t in enumerate(tokens)])


def synth_weights(cfg, types: dict, seed: int = 0, w_std: float = 0.05) -> dict:
    """Deterministic synthetic checkpoint quantized with the oracle's quantizers.
    types: role -> ggml type id for {embd, q, k, v, o, gate, up, down, output}."""
    rng = np.random.default_rng(seed)
    d, ff, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    nq, nkv = cfg.num_heads * hd, cfg.num_kv_heads * hd

    def lin(role, n, k):
        return (types[role], O.quantize(types[role], (rng.standard_normal((n, k)) * w_std).astype(np.float32)))

    w = {"token_embd.weight": lin("embd", cfg.vocab_size, d), "output.weight": lin("output", cfg.vocab_size, d),
         "output_norm.weight": (1 + 0.01 * rng.standard_normal(d)).astype(np.float32)}
    for l in range(cfg.num_layers):
        p = f"blk.{l}."
        w[p + "attn_norm.weight"] = (1 + 0.01 * rng.standard_normal(d)).astype(np.float32)
        w[p + "ffn_norm.weight"] = (1 + 0.01 * rng.standard_normal(d)).astype(np.float32)
        w[p + "attn_q.weight"] = lin("q", nq, d)
        w[p + "attn_k.weight"] = lin("k", nkv, d)
        w[p + "attn_v.weight"] = lin("v", nkv, d)
        w[p + "attn_output.weight"] = lin("o", d, nq)
        E = getattr(cfg, "num_experts", 0)
        if E:  # Mixtral: F32 router + experts stacked over the leading axis
            w[p + "ffn_gate_inp.weight"] = (rng.standard_normal((E, d)) * 0.5).astype(np.float32)
            w[p + "ffn_gate_exps.weight"] = lin("gate", E * ff, d)
            w[p + "ffn_up_exps.weight"] = lin("up", E * ff, d)
            w[p + "ffn_down_exps.weight"] = lin("down", E * d, ff)
            continue
        w[p + "ffn_gate.weight"] = lin("gate", ff, d)
        w[p + "ffn_up.weight"] = lin("up", ff, d)
        w[p + "ffn_down.weight"] = lin("down", d, ff)
    return w
