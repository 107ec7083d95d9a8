"""How far apart are two f32 summation orders of the SAME CPU-path arithmetic (oracle modes "cpu" and "cpu_fast") as the synthetic model gets deeper?
Llama-3-8B layer shapes, random valid Q4_K_M blocks (the weights of tests/test_dec_model.py::_mk_8b_dims), vocab 4096; CPU only.
usage: python profiles/experiments/order_spread.py [layers ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import llama_ref, oracle as O
from mistralrs_amd.llama import LlamaConfig, rope_tables

O.build()
O.set_threads(min(32, os.cpu_count() or 1))
for layers in [int(v) for v in sys.argv[1:]] or [2, 4, 8]:
    d, ff, nq, nkv, vocab = 4096, 14336, 4096, 1024, 4096
    cfg = LlamaConfig(hidden_size=d, intermediate_size=ff, num_layers=layers, num_heads=32, num_kv_heads=8, vocab_size=vocab, head_dim=128,
                      rope_theta=500000.0, max_position_embeddings=512, max_batch=1, max_context_len=256, decode_engine=True, kv_dtype="bf16")
    w, k = {}, [3]

    def blocks(t, n, kk, scale):
        k[0] += 1
        return (t, O.random_blocks(t, n, kk, seed=k[0], d_scale=scale))
    rng = np.random.default_rng(3)
    w["token_embd.weight"] = blocks(O.Q4_K, vocab, d, 1.0)
    w["output.weight"] = blocks(O.Q6_K, vocab, d, 0.02)
    w["output_norm.weight"] = (1 + 0.05 * rng.standard_normal(d)).astype(np.float32)
    for l in range(layers):
        p = f"blk.{l}."
        w[p + "attn_norm.weight"] = (1 + 0.05 * rng.standard_normal(d)).astype(np.float32)
        w[p + "ffn_norm.weight"] = (1 + 0.05 * rng.standard_normal(d)).astype(np.float32)
        w[p + "attn_q.weight"] = blocks(O.Q4_K, nq, d, 0.02)
        w[p + "attn_k.weight"] = blocks(O.Q4_K, nkv, d, 0.02)
        w[p + "attn_v.weight"] = blocks(O.Q6_K, nkv, d, 0.02)
        w[p + "attn_output.weight"] = blocks(O.Q4_K, d, nq, 0.02)
        w[p + "ffn_gate.weight"] = blocks(O.Q4_K, ff, d, 0.02)
        w[p + "ffn_up.weight"] = blocks(O.Q4_K, ff, d, 0.02)
        w[p + "ffn_down.weight"] = blocks(O.Q6_K if l % 2 == 0 else O.Q4_K, d, ff, 0.02)
    cos, sin = rope_tables(cfg)
    a = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu", kv_dtype="bf16")
    b = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu_fast", kv_dtype="bf16")
    tok, rels, t0 = 1000 % vocab, [], time.time()
    for pos in range(6):
        x, y = a.step(tok, pos), b.step(tok, pos)
        rels.append(float(np.abs(x - y).max() / np.abs(x).max()))
        tok = int(x.argmax())
    print(f"layers {layers:2d}: cpu vs cpu_fast, max |dlogit| / max |logit| per position: " + " ".join(f"{r:.2e}" for r in rels) + f"   ({time.time() - t0:.0f} s)", flush=True)
