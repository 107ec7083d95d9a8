#!/usr/bin/env python
"""bench.py -- Llama-3-8B GGUF Q4_K_M, 512-token prefill / 256-token greedy decode on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE decode step (one new token through embedding -> 32 blocks -> norm -> lm_head -> greedy sample) at
batch 1, replayed from a HIP graph; the timed region is exactly K steps after W warm-up steps, bracketed by
barrier + torch.cuda.synchronize(); `value` = decoded tokens/s over all ranks (max time over ranks).
Method mirrors `mistralrs bench` (mistralrs-cli/src/commands/bench.rs:52-55,253-305): synthetic prompt tokens
1000 + (start+i) % 2048, EOS disabled, greedy; prefill tok/s = prompt_len / TTFT is reported next to it.
Weights are synthetic (no network): random valid GGUF blocks with the llama.cpp Q4_K_M tensor-type map; inputs are
resident in HBM when the timed region starts.  Multi-GPU (round 1): N independent replicas (weak scaling), see DESIGN.md.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK = 2.5e15  # dense bf16 FLOP/s (same guide)


def q4_k_m_types(n_layers: int):
    """llama.cpp "Q4_K_M" tensor-type map: Q4_K everywhere, Q6_K for output.weight and for
    attn_v / ffn_down in the use_more_bits layers."""
    from mistralrs_amd.gguf import GgmlDType as T

    def more_bits(i):
        return i < n_layers // 8 or i >= 7 * n_layers // 8 or (i - n_layers // 8) % 3 == 2
    out = {"token_embd.weight": T.Q4K, "output.weight": T.Q6K}
    for i in range(n_layers):
        p = f"blk.{i}."
        out[p + "attn_q.weight"] = T.Q4K
        out[p + "attn_k.weight"] = T.Q4K
        out[p + "attn_v.weight"] = T.Q6K if more_bits(i) else T.Q4K
        out[p + "attn_output.weight"] = T.Q4K
        out[p + "ffn_gate.weight"] = T.Q4K
        out[p + "ffn_up.weight"] = T.Q4K
        out[p + "ffn_down.weight"] = T.Q6K if more_bits(i) else T.Q4K
    return out


def build_model(cfg, device, seed=0, max_new_tokens=4096):
    import torch
    from mistralrs_amd.llama import Llama, random_qtensor
    m = Llama(cfg, device, max_new_tokens=max_new_tokens)
    d, ff, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    nq, nkv = cfg.num_heads * hd, cfg.num_kv_heads * hd
    shapes = {"attn_q": (nq, d), "attn_k": (nkv, d), "attn_v": (nkv, d), "attn_output": (d, nq),
              "ffn_gate": (ff, d), "ffn_up": (ff, d), "ffn_down": (d, ff)}
    types = q4_k_m_types(cfg.num_layers)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for i, (name, t) in enumerate(types.items()):
        if name in ("token_embd.weight", "output.weight"):
            n, k = cfg.vocab_size, d
        else:
            n, k = shapes[name.split(".")[2]]
        m.set_tensor(name, random_qtensor(t, n, k, device, seed * 1000 + i))
    for i in range(cfg.num_layers):
        for nm in ("attn_norm", "ffn_norm"):
            m.set_tensor(f"blk.{i}.{nm}.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
    m.set_tensor("output_norm.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
    return m


def cpu_baseline(model, cfg, budget_s=12.0):
    """Reference CPU path (oracle B restatement, llama_oracle.c): decode a few tokens of the SAME synthetic model on the
    host cores.  Bounded sample; thread count picked by a quick calibration (cgroup quotas make nproc unreliable)."""
    import numpy as np
    from mistralrs_amd.llama import rope_tables
    from oracle import llama_ref, oracle as O
    O.build()
    w = {}
    for name, t in model._keep.items():
        if "#" in name:  # decode-layout copies of the same tensors
            continue
        if hasattr(t, "dtype") and hasattr(t, "shape") and not hasattr(t, "data_ptr"):  # QTensor
            w[name] = (t.dtype.id, t.data.cpu().numpy().reshape(t.shape[0], -1))
        else:
            w[name] = t.cpu().numpy()
    cos, sin = rope_tables(cfg)
    # calibrate threads on one big matvec
    tname = "blk.0.ffn_gate.weight"
    x = np.random.default_rng(0).standard_normal((1, cfg.hidden_size)).astype(np.float32)
    best = (1e9, 1)
    for thr in sorted({1, 2, 4, 8, 16, 32, 64, os.cpu_count() or 1}):
        if thr > (os.cpu_count() or 1):
            continue
        O.set_threads(thr)
        O.gemv_cpu_fast(w[tname][0], w[tname][1], w[tname][1].shape[0], cfg.hidden_size, x)
        t0 = time.perf_counter()
        for _ in range(3):
            O.gemv_cpu_fast(w[tname][0], w[tname][1], w[tname][1].shape[0], cfg.hidden_size, x)
        dt = (time.perf_counter() - t0) / 3
        if dt < best[0]:
            best = (dt, thr)
    O.set_threads(best[1])
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu_fast", kv_dtype="f32")
    tok, n, t0 = 1000, 0, time.perf_counter()
    while True:
        lg = ref.step(tok, n)
        tok, n = int(lg.argmax()), n + 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 16:
            break
    return {"value": round(n / el, 3), "unit": "tokens/s", "cores": best[1], "kind": "port",
            "sample": f"{n} greedy decode tokens from an empty context, same synthetic Llama-3-8B Q4_K_M weights, "
                      f"oracle-B restatement of the candle CPU path (Q8_K activations, OpenMP rows, gcc -O3 -march=native); "
                      f"host reports {os.cpu_count()} logical CPUs"}


def measured_traffic(model_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE on this same command,
    x2 gfx950 correction; scripts/profile_round.sh -> profiles/round1_hbm_traffic.json).  Counters cannot be read from inside the
    timed process, so the figure is the last profiled one for this kernel and workload; null for any other workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_hbm_traffic.json")
    try:
        k = json.load(open(path))["kernels"]["void mrs::decode_gemv_kernel<1, 1, 2, 512>(mrs::DecodeGemvArgs)"]
    except (OSError, KeyError, ValueError):
        return {"traffic": None}
    if "8B" not in model_name:
        return {"traffic": None}
    return {"traffic": int(k["read_bytes_per_launch"] + k["write_bytes_per_launch"]), "traffic_source": "profiles/round1_hbm_traffic.json"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small", action="store_true", help="tiny config (smoke / CI), not the benchmark")
    ap.add_argument("--tp", action="store_true", help="N > 1: ONE model sharded tensor-parallel over the N GPUs (RCCL all-reduce) instead of N replicas")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.llama import LlamaConfig
    ctx_needed = a.prompt_len + a.warmup + a.steps + 2
    max_ctx = (ctx_needed + 63) // 64 * 64
    if a.small:
        cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2, vocab_size=2048,
                          head_dim=64, max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "tiny-llama (smoke)"
    else:
        cfg = LlamaConfig.llama3_8b(max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "Llama-3-8B"
    tp = a.tp and world > 1
    if tp:  # column / row parallel shards (mistralrs-quant/src/distributed/layers.rs): local heads, kv heads, ffn
        from mistralrs_amd import distributed as D
        cfg.head_dim = cfg.head_dim  # keep the global head_dim
        cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size = D.local_dims(cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size, world)
        cfg.tp_world_size, cfg.tp_rank = world, rank
    model = build_model(cfg, dev, seed=rank, max_new_tokens=a.warmup + a.steps + 8)
    if tp:
        from mistralrs_amd import distributed as D
        model.set_comm(D.RcclComm(rank, world, dev))
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- prefill (TTFT), reference method: prompt_len / time-to-first-token
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(a.prompt_len)]
    model.prefill(prompt, 0)  # warm-up (lazy code-object loads, workspace allocation); the timed run overwrites the same pages
    sync()
    t0 = time.perf_counter()
    last = model.prefill(prompt, 0)
    first_tok = int(last.argmax())  # device -> host read-back of the first token: end of TTFT
    ttft = time.perf_counter() - t0
    prefill_flops = model.prefill_flops(a.prompt_len)

    # ---------------- decode: HIP graph of one step, replayed
    model.set_state([first_tok], [a.prompt_len])
    model.step_counter.zero_()
    model.capture_decode_graph(1)
    for _ in range(a.warmup):
        model.replay()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        model.replay()
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    dev_s = ev0.elapsed_time(ev1) / 1e3
    tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    t_all = float(tmax.item())
    toks = model.tokens_out[0, : a.warmup + a.steps].cpu().numpy()
    assert int(model.positions[0]) == a.prompt_len + a.warmup + a.steps, "decode state did not advance as expected"

    # ---------------- roofline of the dominant kernel: the fused gate/up GEMV (2 x [ffn, d] weights per launch)
    import ctypes as C
    from mistralrs_amd import _lib
    ext = _lib.load("ext")
    ext.mrs_decode_gate_up.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    h = torch.randn(1, cfg.hidden_size, device=dev)
    y = torch.zeros(((cfg.intermediate_size + 511) // 512 * 512) // 32 * 36, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    layers = [(model._keep[f"blk.{i}.ffn_gate.weight"], model._keep[f"blk.{i}.ffn_up.weight"], model._keep[f"blk.{i}.ffn_norm.weight"])
              for i in range(cfg.num_layers)]

    def gate_up_pass():
        for g, u, nw in layers:
            ext.mrs_decode_gate_up(g.data.data_ptr(), u.data.data_ptr(), g.dtype.id, cfg.intermediate_size, cfg.hidden_size,
                                   h.data_ptr(), nw.data_ptr(), cfg.rms_eps, 0, y.data_ptr(), y.numel() // 36, 1, st)
    gate_up_pass()
    torch.cuda.synchronize()
    reps = 8
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for _ in range(reps):
        gate_up_pass()
    k1.record()
    torch.cuda.synchronize()
    kern_s = k0.elapsed_time(k1) / 1e3 / (reps * len(layers))
    kern_bytes = layers[0][0].nbytes() + layers[0][1].nbytes()  # algorithmic bytes per launch: the two packed weight tensors
    achieved = kern_bytes / kern_s

    avg_ctx = a.prompt_len + a.warmup + a.steps / 2
    step_bytes = model.decode_bytes(1, int(avg_ctx))
    tok_s = (1 if tp else world) * a.steps / t_all  # TP: the N GPUs decode ONE sequence
    out = {
        "metric": "decode_tokens_per_sec", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * t_all / a.steps, 4), "higher_is_better": True, "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "q4_k/q6_k weights x q8_1 activations (int8 dot, f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"{name} GGUF Q4_K_M, TP=1, {a.prompt_len} prefill / {a.steps} decode, batch 1, paged KV bf16 (block 32)",
                   "parallelism": "tp1" if world == 1 else (f"tp{world}" if tp else f"replicas x{world}")},
        "prefill_tokens_per_sec": round(a.prompt_len / ttft, 1), "ttft_ms": round(1e3 * ttft, 2),
        "prefill_roofline": {"bound": "mfma", "achieved": round(prefill_flops / ttft / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                             "frac": round(prefill_flops / ttft / MFMA_PEAK, 4), "flops": prefill_flops,
                             "note": "fused block-dequant -> bf16 MFMA GEMMs (mrs_gemm_q_bf16_multi) + MFMA flash attention over the paged cache; whole prompt incl. the host read-back of the first token"},
        "device_ms_per_step": round(1e3 * dev_s / a.steps, 4),
        "step_bytes": int(step_bytes), "step_roofline_frac": round(step_bytes * (a.steps / t_all) / HBM_PEAK, 4),
        "roofline": {"bound": "hbm", "kernel": "decode_gemv_kernel<1, PRO_NORM, EPI_GLU_Q8_1> (fused RMSNorm+Q8_1+gate/up GEMV+SiLU*mul+Q8_1)",
                     "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                     "bytes_per_launch": int(kern_bytes), "us_per_launch": round(kern_s * 1e6, 2), **measured_traffic(name)},
        "greedy_tokens_head": [int(t) for t in toks[a.warmup: a.warmup + 8]],
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(model, cfg)
        except Exception as e:  # the baseline is a reported extra, never fatal
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
