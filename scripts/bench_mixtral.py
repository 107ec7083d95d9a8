#!/usr/bin/env python
"""BASELINE configs[4] on ONE MI355X (the reference quotes it at TP=2): Mixtral-8x7B GGUF Q4_K_M-shaped synthetic model, batch-1 decode
through the C++ runner's MoE path (router top-2 on the device, expert-indexed fused GEMVs), HIP-graph replay.  One JSON line.

    python scripts/bench_mixtral.py [--steps 128] [--prompt-len 32] [--layers 32]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--prompt-path", choices=["grouped", "chunked"], default="grouped",
                    help="grouped: mrs_llama_prefill (MFMA attention half + route dispatch / grouped expert GEMMs); chunked: 8 tokens per step through the decode kernels")
    ap.add_argument("--layers", type=int, default=32)
    a = ap.parse_args()
    import torch
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.gguf import GgmlDType as T
    from mistralrs_amd.llama import Llama, LlamaConfig, random_qtensor
    import bench
    dev = torch.device("cuda:0")
    max_ctx = (a.prompt_len + a.warmup + a.steps + 2 + 63) // 64 * 64
    cfg = LlamaConfig.mixtral_8x7b(max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
    cfg.num_layers = a.layers
    m = Llama(cfg, dev, max_new_tokens=a.warmup + a.steps + 8)
    d, ff, hd, E = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim, cfg.num_experts
    nq, nkv = cfg.num_heads * hd, cfg.num_kv_heads * hd
    types = bench.q4_k_m_types(cfg.num_layers)
    g = torch.Generator(device="cpu").manual_seed(0)
    seed = 0
    for name, t in types.items():
        seed += 1
        role = name.split(".")[2] if name.startswith("blk.") else name
        if role in ("token_embd.weight", "output.weight"):
            m.set_tensor(name, random_qtensor(t, cfg.vocab_size, d, dev, seed))
        elif role in ("ffn_gate", "ffn_up"):
            m.set_tensor(name.replace(role, role + "_exps"), random_qtensor(t, E * ff, d, dev, seed))
        elif role == "ffn_down":
            m.set_tensor(name.replace(role, role + "_exps"), random_qtensor(t, E * d, ff, dev, seed))
        else:
            n, k = {"attn_q": (nq, d), "attn_k": (nkv, d), "attn_v": (nkv, d), "attn_output": (d, nq)}[role]
            m.set_tensor(name, random_qtensor(t, n, k, dev, seed))
    for i in range(cfg.num_layers):
        for nm in ("attn_norm", "ffn_norm"):
            m.set_tensor(f"blk.{i}.{nm}.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
        m.set_tensor(f"blk.{i}.ffn_gate_inp.weight", 0.05 * torch.randn(E, d, generator=g))
    m.set_tensor("output_norm.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
    torch.cuda.synchronize()
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(a.prompt_len)]
    run_prompt = (lambda: m.prefill(prompt, 0)) if a.prompt_path == "grouped" else (lambda: m.prefill_chunked(prompt, 0, chunk=8))
    run_prompt()  # warm-up (lazy code-object loads, workspace allocation); the timed run overwrites the same pages
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run_prompt()
    first = int(last.argmax())
    ttft = time.perf_counter() - t0
    m.set_state([first], [a.prompt_len])
    m.step_counter.zero_()
    m.capture_decode_graph(1)
    for _ in range(a.warmup):
        m.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_bytes = m.decode_bytes(1, a.prompt_len + a.warmup + a.steps // 2)
    print(json.dumps({"metric": "decode_tokens_per_sec", "value": round(a.steps / dt, 2), "unit": "tokens/s", "n_gpus": 1, "steps": a.steps,
                      "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "dtype": ("q4_k/q6_k weights x q8_k activations (decode engine: the reference CPU path's arithmetic)" if m.decode_path == "engine" else "q4_k/q6_k weights x q8_1 activations"), "decode_path": m.decode_path,
                      "data": "synthetic", "config": {"workload": f"Mixtral-8x7B-shaped GGUF Q4_K_M ({cfg.num_layers} layers, 8 experts top-2), TP=1, "
                                                                  f"{a.prompt_len}-token prompt ({a.prompt_path} prompt path) / {a.steps} decode, batch 1"},
                      "step_bytes": int(step_bytes), "step_roofline_frac": round(step_bytes * a.steps / dt / 8e12, 4),
                      "prompt_tokens_per_sec": round(a.prompt_len / ttft, 1)}))


if __name__ == "__main__":
    main()
