"""The oracle's prompt pass (oracle/llama_ref.py LlamaRef.prefill) and the restatement of the reference's CPU prompt attention
(attention/backends/cpu/full.rs + mask.rs -> cpu_path_oracle.c orc_attention_full_cpu).  CPU only."""
import types

import numpy as np
import pytest

from oracle import llama_ref, oracle as O


def _attention_f64(q, k, v, scale, window=0):
    """Plain softmax attention per query in f64 (causal; queries are the last T positions)."""
    T, H, hd = q.shape
    S, KVH, _ = k.shape
    out = np.zeros((T, H, hd))
    for t in range(T):
        qp = S - T + t
        lo = qp - window + 1 if window and qp >= window else 0
        for h in range(H):
            kk, vv = k[lo:qp + 1, h // (H // KVH)].astype(np.float64), v[lo:qp + 1, h // (H // KVH)].astype(np.float64)
            s = kk @ q[t, h].astype(np.float64) * scale
            p = np.exp(s - s.max())
            out[t, h] = (p / p.sum()) @ vv
    return out


@pytest.mark.parametrize("T,S,H,KVH,hd,window", [(1, 1, 4, 2, 64, 0), (5, 5, 4, 1, 128, 0), (19, 19, 8, 2, 128, 0), (140, 140, 4, 2, 64, 0), (9, 300, 4, 4, 32, 0),
                                                 (70, 70, 2, 1, 128, 16), (33, 200, 4, 2, 64, 150)])
def test_full_attention_restatement_vs_f64(T, S, H, KVH, hd, window):
    rng = np.random.default_rng(T * 1000 + S)
    q = rng.standard_normal((T, H, hd)).astype(np.float32)
    k = rng.standard_normal((S, KVH, hd)).astype(np.float32)
    v = rng.standard_normal((S, KVH, hd)).astype(np.float32)
    scale = np.float32(1 / np.sqrt(hd))
    got = O.attention_full_cpu(q, k, v, scale, window)
    want = _attention_f64(q, k, v, float(scale), window)
    assert np.abs(got - want).max() < 2e-5  # f32 sums + the Cephes fast_exp (elem.rs:417-433, ~2e-7 relative)


def test_full_attention_q_blocks_do_not_interact():
    """Each query row's result depends only on its own live range: rows of a Q_BLOCK computed together equal the same rows computed in a block of
    their own whenever their tiles start at the same kv position (kv_lo = 0 for causal rows)."""
    rng = np.random.default_rng(3)
    T, H, KVH, hd = 24, 2, 1, 64
    q = rng.standard_normal((T, H, hd)).astype(np.float32)
    k = rng.standard_normal((T, KVH, hd)).astype(np.float32)
    v = rng.standard_normal((T, KVH, hd)).astype(np.float32)
    full = O.attention_full_cpu(q, k, v, 0.125)
    for t in (0, 7, 8, 23):
        one = O.attention_full_cpu(q[t:t + 1], k[:t + 1], v[:t + 1], 0.125)
        assert np.array_equal(one[0], full[t])


def _tiny():
    cfg = types.SimpleNamespace(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, head_dim=128, vocab_size=96,
                                rms_eps=1e-5, rope_interleaved=True, num_experts=0)
    T = 12  # ggml type ids: Q4_K = 12, Q6_K = 14
    w = llama_ref.synth_weights(cfg, {"embd": 12, "q": 12, "k": 12, "v": 14, "o": 12, "gate": 12, "up": 12, "down": 14, "output": 14}, seed=5)
    half = cfg.head_dim // 2
    inv = 1.0 / (10000.0 ** (np.arange(half) / half))
    ang = np.arange(64)[:, None] * inv[None]
    return cfg, w, np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def test_engine_prefill_equals_step_loop():
    cfg, w, cos, sin = _tiny()
    toks = [3, 17, 40, 8, 91, 2, 55, 60, 11]
    a = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine")
    b = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine")
    la = a.prefill(toks)
    lb = b.run(toks)[-1]
    assert np.array_equal(la, lb)
    for l in range(cfg.num_layers):
        assert np.array_equal(np.stack(a.k[l]), np.stack(b.k[l])) and np.array_equal(np.stack(a.v[l]), np.stack(b.v[l]))
    # and the cache it leaves continues identically
    assert np.array_equal(a.step(7, len(toks)), b.step(7, len(toks)))


def test_cpu_prefill_close_to_step_loop():
    """On the CPU path the prompt (full.rs) and the token-by-token loop (single_q.rs) differ only in f32 summation order."""
    cfg, w, cos, sin = _tiny()
    toks = list(range(20, 37))
    a = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu")
    b = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu")
    la, lb = a.prefill(toks), b.run(toks)[-1]
    assert np.abs(la - lb).max() / np.abs(lb).max() < 5e-3
