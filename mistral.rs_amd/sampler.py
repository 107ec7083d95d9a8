"""Top-k / top-p / min-p sampling: the role of `Sampler::sample_topk_on_device` (mistralrs-core/src/sampler.rs:1171-1290).

Device half (csrc/sampling.hip, reference ABI `topk_large_f32_packed[_batched]`, ops.rs:691-1000): the k largest logits of a row in (value descending, index
ascending) order + the pieces of the full-softmax normaliser, ONE small device -> host copy of `2k + 2` floats per row.  Host half (this file): probabilities of the
candidates under the FULL softmax, the top-p cut, the min-p cut, the weighted draw.  The reference draws with `rand`'s Isaac64Rng + WeightedIndex; this module draws
with numpy's Generator from the SAME weights, so the distribution is the reference's and the random stream is not (documented in DESIGN.md)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

CHUNK_SIZE = 2048      # CUDA_TOPK_CHUNK_SIZE (ops.rs:12)
MAX_K = 128            # CUDA_TOPK_MAX_K (ops.rs:18)
MAX_STAGE2 = 47 * 1024  # CUDA_TOPK_MAX_STAGE2_CANDIDATES (ops.rs:20)


class TopK:
    """Workspace + launcher for rows of `vocab` f32 logits (cuda_topk_logits_f32_packed / _batched, ops.rs:691-1000)."""

    def __init__(self, vocab: int, k: int, device, max_rows: int = 1):
        k = min(int(k), int(vocab))
        if vocab <= 0:
            raise ValueError("top-k: empty logits")
        if k == 0 or k > MAX_K:
            raise ValueError(f"top-k: k={k} must be in [1, {MAX_K}]")
        self.vocab, self.k, self.max_rows, self.device = vocab, k, max_rows, device
        self.nblocks = (vocab + CHUNK_SIZE - 1) // CHUNK_SIZE
        if self.nblocks * k > MAX_STAGE2:
            raise ValueError(f"top-k workspace too large: {self.nblocks * k} candidates")
        f32 = dict(dtype=torch.float32, device=device)
        self.block_values = torch.empty(max_rows, self.nblocks, k, **f32)
        self.block_indices = torch.empty(max_rows, self.nblocks, k, dtype=torch.int32, device=device)
        self.block_maxes = torch.empty(max_rows, self.nblocks, **f32)
        self.block_sums = torch.empty(max_rows, self.nblocks, **f32)
        self.packed = torch.empty(max_rows, 2 * k + 2, **f32)
        self._inv_t = torch.empty(max_rows, **f32)
        vp, i, f, ll = C.c_void_p, C.c_int, C.c_float, C.c_int64
        self._one = _lib.sym("core", "topk_large_f32_packed", [vp, vp, vp, vp, vp, vp, i, i, i, i, f, ll])
        self._many = _lib.sym("core", "topk_large_f32_packed_batched", [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, ll])

    def __call__(self, logits: torch.Tensor, temperature) -> torch.Tensor:
        """logits f32 [vocab] or [rows, vocab] (contiguous, on the device); temperature: a positive finite float, or one per row.  Returns the packed rows
        [rows, 2k + 2] on the device (a view of this object's buffer: consume it before the next call)."""
        x = logits.reshape(-1, self.vocab) if logits.dim() > 1 else logits.reshape(1, self.vocab)
        rows = x.shape[0]
        if x.dtype != torch.float32 or not x.is_contiguous() or rows > self.max_rows:
            raise ValueError("top-k: logits must be contiguous f32 with at most max_rows rows")
        temps = np.broadcast_to(np.asarray(temperature, dtype=np.float64), (rows,))
        if not np.all(np.isfinite(temps) & (temps > 0)):
            raise ValueError("top-k requires a positive finite temperature")
        st = torch.cuda.current_stream().cuda_stream
        if rows == 1:
            self._one(x.data_ptr(), self.block_values.data_ptr(), self.block_indices.data_ptr(), self.block_maxes.data_ptr(), self.block_sums.data_ptr(),
                      self.packed.data_ptr(), self.vocab, self.k, CHUNK_SIZE, self.nblocks, float(np.float32(1.0 / temps[0])), st)
        else:
            self._inv_t[:rows].copy_(torch.from_numpy((1.0 / temps).astype(np.float32)), non_blocking=False)
            self._many(x.data_ptr(), self._inv_t.data_ptr(), self.block_values.data_ptr(), self.block_indices.data_ptr(), self.block_maxes.data_ptr(),
                       self.block_sums.data_ptr(), self.packed.data_ptr(), rows, self.vocab, self.k, CHUNK_SIZE, self.nblocks, st)
        return self.packed[:rows]


class Top1:
    """Greedy rows: `top1_large_f32_packed[_batched]` (cuda_top1_logits_f32_*, ops.rs:1232-2050) -- packed [rows][2] = (max logit, token id as f32); no temperature."""

    def __init__(self, vocab: int, device, max_rows: int = 1):
        if vocab <= 0:
            raise ValueError("top-1: empty logits")
        self.vocab, self.max_rows, self.device = vocab, max_rows, device
        self.nblocks = (vocab + CHUNK_SIZE - 1) // CHUNK_SIZE
        self.block_values = torch.empty(max_rows, self.nblocks, dtype=torch.float32, device=device)
        self.block_indices = torch.empty(max_rows, self.nblocks, dtype=torch.int32, device=device)
        self.packed = torch.empty(max_rows, 2, dtype=torch.float32, device=device)
        vp, i, ll = C.c_void_p, C.c_int, C.c_int64
        self._many = _lib.sym("core", "top1_large_f32_packed_batched", [vp, vp, vp, vp, vp, i, i, i, i, ll])

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        x = logits.reshape(-1, self.vocab)
        rows = x.shape[0]
        if x.dtype != torch.float32 or not x.is_contiguous() or rows > self.max_rows:
            raise ValueError("top-1: logits must be contiguous f32 with at most max_rows rows")
        self._many(x.data_ptr(), self.block_values.data_ptr(), self.block_indices.data_ptr(), self.packed.data_ptr(), None, rows, self.vocab, CHUNK_SIZE, self.nblocks,
                   torch.cuda.current_stream().cuda_stream)
        return self.packed[:rows]


def top1_token(packed2) -> int:
    """Sampler::cuda_top1_token (sampler.rs:1284-1297): the packed pair must be finite and hold a non-negative integer"""
    mx, ix = float(packed2[0]), float(packed2[1])
    if not (np.isfinite(mx) and np.isfinite(ix)) or ix < 0 or ix != np.floor(ix):
        raise ValueError(f"invalid CUDA top-1 output: max_logit={mx} argmax={ix}")
    return int(ix)


def filtered_probs(packed: np.ndarray, k: int, temperature: float, top_p: float = 1.0, min_p: float = 0.0):
    """sampler.rs:1189-1236 for one packed row: (token ids [k], reporting probabilities [k], weights after the top-p and min-p cuts [k]); f32 like the reference."""
    vals, ids = packed[:k].astype(np.float32), packed[k:2 * k].astype(np.uint32)
    denom, gmax = np.float32(packed[2 * k]), np.float32(packed[2 * k + 1])
    if not (denom > 0 and np.isfinite(denom) and np.isfinite(gmax)):
        raise ValueError("invalid top-k softmax normalizer")
    inv_t = np.float32(1.0 / temperature)
    with np.errstate(over="ignore", invalid="ignore"):
        rep = (np.exp((vals * inv_t - gmax).astype(np.float32), dtype=np.float32) / denom).astype(np.float32)
    probs = rep.copy()
    if 0.0 < top_p < 1.0:
        total = np.float32(0)
        for p in probs:
            total = np.float32(total + p)
        cutoff, cum = np.float32(top_p) * total, np.float32(0)
        for j in range(k):
            if cum >= cutoff:
                probs[j] = 0.0
            else:
                cum = np.float32(cum + probs[j])
    if 0.0 < min_p < 1.0 and k:
        probs[np.float32(probs[0] * np.float32(min_p)) >= probs] = 0.0
    return ids, rep, probs


def sample(packed: np.ndarray, k: int, temperature: float, top_p: float, min_p: float, rng: np.random.Generator):
    """One draw: (token id, its reporting probability).  Raises like the reference when every weight is zero."""
    ids, rep, probs = filtered_probs(packed, k, temperature, top_p, min_p)
    # WeightedIndex::new (sampler.rs:1238-1258) refuses NaN / infinite / negative weights and an all-zero set; nothing is silently zeroed
    if float(np.where(np.isfinite(probs) & (probs > 0), probs, 0).astype(np.float64).sum()) == 0.0:
        raise ValueError("All sampling probabilities are zero after CUDA top-k filtering.")
    if not np.all(np.isfinite(probs)) or np.any(probs < 0):
        raise ValueError("Failed to construct CUDA top-k multinomial sampler: invalid weight")
    w = probs.astype(np.float64)
    j = int(rng.choice(k, p=w / w.sum()))
    return int(ids[j]), float(rep[j])


def generate(model, prompt, max_new_tokens: int, top_k: int, temperature: float = 1.0, top_p: float = 1.0, min_p: float = 0.0, seed: int = 0):
    """Sampled decoding on a `Llama` runner (the loop of `Sampler::sample` with top_k set, sampler.rs:1262-1290): prefill, then per token one decode step, the device
    top-k over the logits row, `2k + 2` floats to the host, the top-p / min-p cuts and the draw there.  top_k == 1 takes the arg-max through the top-1 kernels, no temperature, probability 1
    (sample_cuda_top1_row).  Returns (tokens, reporting probabilities)."""
    rng = np.random.default_rng(seed)
    k = min(int(top_k), int(model.cfg.vocab_size))
    t1 = Top1(model.cfg.vocab_size, model.device) if k == 1 else None
    tk = None if k == 1 else TopK(model.cfg.vocab_size, top_k, model.device)
    logits = model.prefill(list(prompt), 0).float().reshape(1, -1)  # sequence 0: the decode steps below run batch row 0
    toks, probs = [], []
    for i in range(max_new_tokens):
        if hasattr(model, "p2p_sync_error") and model.p2p_sync_error():  # tensor parallel: a timed-out peer-mailbox sum is NaN -- never hand out a token from it
            raise RuntimeError("p2p all-reduce timed out: the route has been dropped on every rank (RCCL from now on); re-run the request")
        if k == 1:  # sample_cuda_top1_row (sampler.rs:767-781): the arg-max, no temperature, logprob 0 (probability 1)
            tok, p = top1_token(t1(logits.contiguous()).cpu().numpy()[0]), 1.0
        else:
            packed = tk(logits.contiguous(), temperature).cpu().numpy()[0]
            tok, p = sample(packed, tk.k, temperature, top_p, min_p, rng)
        toks.append(tok)
        probs.append(p)
        if i + 1 < max_new_tokens:
            model.set_state([tok], [len(prompt) + i])
            logits = model.forward_logits(1)[0:1].float()
    return toks, probs
