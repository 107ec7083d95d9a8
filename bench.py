#!/usr/bin/env python
"""bench.py -- Llama-3-8B GGUF Q4_K_M, 512-token prefill / 256-token greedy decode on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE decode step (one new token through embedding -> 32 blocks -> norm -> lm_head -> greedy sample) at
batch 1, replayed from a HIP graph; the timed region is exactly K steps after W warm-up steps, bracketed by
barrier + torch.cuda.synchronize(); `value` = decoded tokens/s over all ranks (max time over ranks).
Method mirrors `mistralrs bench` (mistralrs-cli/src/commands/bench.rs:52-55,253-305): synthetic prompt tokens
1000 + (start+i) % 2048, EOS disabled, greedy; prefill tok/s = prompt_len / TTFT is reported next to it.
Weights are synthetic (no network), SURVEY 8(d): N(0, 0.02^2) per tensor (seed = f(tensor index)), norm weights 1 + N(0, 0.01), quantized ON THE GPU by the
device ISQ quantizers (csrc/ext_isq.hip: bit-identical to GGML's reference quantizers) with the llama.cpp Q4_K_M tensor-type map (`--weights blocks` = random
valid block bytes, the fast variant for pure kernel timing); inputs are resident in HBM when the timed region starts.
Multi-GPU: `--gpus N` with N > 1 runs ONE model tensor-parallel over the N GPUs (one process per GPU, RCCL all-reduce after every
row-parallel projection; "scaling": "strong"); launched by torch.distributed.run, or self-spawned when WORLD_SIZE is not set.
`--model auto` = Llama-3-8B (BASELINE configs[1]) for N < 8 and Llama-3-70B, 2048 prefill (configs[3]) for N = 8; `--replicas` = N independent copies.
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


def build_model(cfg, device, seed=0, max_new_tokens=4096, tp=None, quant="q4_k_m", weights="gaussian"):
    """Synthetic model of `cfg`'s (per-rank) dims.  tp = (rank, world): the column / row-parallel shards (q / k / v / gate / up rows, o / down
    columns: distributed/layers.rs:695-975,1160-1616) get rank-specific random blocks of the SHARD's shape -- the bytes and the arithmetic of a
    real shard without materialising the unsharded 40 GB tensor on every GPU -- while the replicated tensors (embedding, norms, lm_head) use
    the same seed on every rank, so all ranks compute the same logits and sample the same token."""
    import torch
    from mistralrs_amd.llama import Llama, random_qtensor
    m = Llama(cfg, device, max_new_tokens=max_new_tokens)
    d, ff, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    nq, nkv = cfg.num_heads * hd, cfg.num_kv_heads * hd
    shapes = {"attn_q": (nq, d), "attn_k": (nkv, d), "attn_v": (nkv, d), "attn_output": (d, nq),
              "ffn_gate": (ff, d), "ffn_up": (ff, d), "ffn_down": (d, ff)}
    types = q4_k_m_types(cfg.num_layers)
    g = torch.Generator(device="cpu").manual_seed(seed)
    E = cfg.num_experts
    for i, (name, t) in enumerate(types.items()):
        sharded = name not in ("token_embd.weight", "output.weight")
        n, k = shapes[name.split(".")[2]] if sharded else (cfg.vocab_size, d)
        role = name.split(".")[2] if sharded else ""
        if E and role in ("ffn_gate", "ffn_up", "ffn_down"):  # Mixtral: experts stacked along the row axis, [E * n][k] packed blocks (weight_source.rs:1835)
            name, n = name.replace(role, role + "_exps"), E * n
        tseed = seed * 1000 + i + (7919 * (tp[0] + 1) if tp and sharded else 0)
        if quant == "q8_0_isq":  # in-situ quantisation: bf16 weights -> Q8_0 blocks on the device (utils/isq.rs:323-361 does this on the host cores)
            from mistralrs_amd import isq
            from mistralrs_amd.gguf import GgmlDType
            gw = torch.Generator(device=device).manual_seed(tseed)
            m.set_tensor(name, isq.quantize((torch.randn(n, k, device=device, generator=gw) * 0.02).to(torch.bfloat16), GgmlDType.Q8_0))
        elif weights == "gaussian":  # N(0, 0.02^2) through the GGML quantizer of the tensor's type (on the device, bit-identical to GGML)
            from mistralrs_amd import isq
            gw = torch.Generator(device=device).manual_seed(tseed)
            m.set_tensor(name, isq.quantize(torch.randn(n, k, device=device, generator=gw) * 0.02, t))
        else:
            m.set_tensor(name, random_qtensor(t, n, k, device, tseed))
    for i in range(cfg.num_layers if E else 0):
        m.set_tensor(f"blk.{i}.ffn_gate_inp.weight", 0.05 * torch.randn(E, d, generator=g))  # F32 router (models/mixtral.rs:262-304)
    for i in range(cfg.num_layers):
        for nm in ("attn_norm", "ffn_norm"):
            m.set_tensor(f"blk.{i}.{nm}.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
    m.set_tensor("output_norm.weight", 1.0 + 0.01 * torch.randn(d, generator=g))
    return m


def host_weights(model):
    """The model's tensors as the CPU restatement takes them (GGUF name -> (ggml type id, packed uint8 [N, row_bytes]) or f32 array)."""
    w = {}
    for name, t in model._keep.items():
        if "#" in name:  # decode-layout copies of the same tensors
            continue
        if hasattr(t, "dtype") and hasattr(t, "shape") and not hasattr(t, "data_ptr"):  # QTensor
            w[name] = (t.dtype.id, t.data.cpu().numpy().reshape(t.shape[0], -1))
        else:
            w[name] = t.cpu().numpy()
    return w


def cpu_baseline(model, cfg, prompt, positions=16):
    """Reference CPU path on the host cores, the SAME synthetic model (oracle/llama_ref.py + llama_oracle.c / cpu_path_oracle.c): the parity prompt
    (`prompt`, >= 64 tokens, in ONE pass: per-row Q8_K activations, full.rs attention) followed by `positions` greedy decode steps.
      * timing ("port"): mode "cpu_fast" (= cpu order b: one f32 term per superblock, 2 kv chunks, OpenMP rows); the timed sample is its decode steps;
      * parity material, teacher-forced on that run's tokens:
          "cpu"    order a: ggml's generic GEMV order, candle's in-order rms sum, full.rs (prompt) / single_q.rs (decode) attention
          "engine" the same arithmetic in the decode engine's documented summation orders: the HIP engine must equal it bit for bit (prompt and decode)
          "exact"  dequantized weights, NO activation quantization, f64 accumulation and glue: the model every quantized evaluation approximates
    Every mode rounds K / V to bf16 on the way into the cache (the page format).  Thread count picked by a quick calibration (cgroup quotas make nproc unreliable).
    Returns (baseline record, fed tokens, {mode: [positions + 1 logit vectors: last prompt position, then after each fed token]}, {mode: LlamaRef})."""
    import numpy as np
    from mistralrs_amd.llama import rope_tables
    from oracle import llama_ref, oracle as O
    O.build()
    w = host_weights(model)
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
    P = len(prompt)
    nblk = (cfg.max_context_len + 31) // 32
    kw = {"cpu_b": dict(mode="cpu_fast", n_kv_chunks=2), "cpu": dict(mode="cpu"), "exact": dict(mode="exact"),
          "engine": dict(mode="engine", attn_bpw=1 if nblk <= 64 else (nblk + 63) // 64)}
    runs = {k: llama_ref.LlamaRef(cfg, w, cos, sin, kv_dtype="bf16", **v) for k, v in kw.items()}
    f32 = lambda lg: np.asarray(lg, dtype=np.float32).copy()
    # same token rule as the GPU run; the GPU side repeats exactly this (prefill of the prompt, then token by token through the decode engine)
    t0 = time.perf_counter()
    lb = [f32(runs["cpu_b"].prefill(prompt))]
    t_prompt = time.perf_counter() - t0
    fed, t0 = [], time.perf_counter()
    for i in range(positions):
        fed.append(int(lb[-1].argmax()))
        lb.append(f32(runs["cpu_b"].step(fed[-1], P + i)))
    el = time.perf_counter() - t0
    logits = {"cpu_b": lb}
    for k in ("cpu", "engine", "exact"):
        logits[k] = [f32(runs[k].prefill(prompt))] + [f32(runs[k].step(t, P + i)) for i, t in enumerate(fed)]
    base = {"value": round(positions / el, 3), "unit": "tokens/s", "cores": best[1], "kind": "port", "oracle_pinned": False,
            "prompt_tokens_per_sec": round(P / t_prompt, 2),
            "sample": f"{positions} greedy decode tokens after a {P}-token prompt (context {P}..{P + positions - 1}), same synthetic {cfg.num_layers}-layer Q4_K_M weights, "
                      f"restatement of the candle CPU path (Q8_K activations, integer block dots, OpenMP rows, gcc -O3 -march=native); "
                      f"host reports {os.cpu_count()} logical CPUs",
            "oracle_pinned_note": "candle / ggml sources are not in /root/reference (Cargo dependency): the CPU-path arithmetic is restated from the published ggml "
                                  "algorithms and pinned on the reference's own CUDA sources compiled on the host (oracle/_ref) for the block formats, not on a candle run"}
    return base, fed, logits, runs


def dropin_rate(model, cfg, prompt, steps, device):
    """What an UNMODIFIED mistralrs-core host gets from the drop-in libraries: the literal reference launch sequence of a decode step through the
    reference ABI (launch_mmvq_gguf_quantize_q8_1_* + launch_mmvq_gguf_<t>_* per projection, rotary_embedding, reshape_and_cache, paged_attention_v1 / v2,
    add_rms_norm_*; runner mode use_fused = 0) on the same weights, replayed from a HIP graph like the engine."""
    import copy
    import torch
    from mistralrs_amd.llama import Llama
    c2 = copy.copy(cfg)
    c2.use_fused, c2.decode_engine = False, False
    m2 = Llama(c2, device, max_new_tokens=steps + 16)
    for name, t in model._keep.items():
        if "#" not in name:
            m2.set_tensor(name, t)  # the same device tensors (GGUF blocks); no decode-layout copies in this mode
    assert m2.decode_path == "reference-sequence", m2.decode_path
    last = m2.prefill(prompt, 0)
    m2.set_state([int(last.argmax())], [len(prompt)])
    m2.step_counter.zero_()
    m2.capture_decode_graph(1)
    for _ in range(4):
        m2.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m2.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    avg_ctx = len(prompt) + 4 + steps / 2
    return steps / dt, m2.decode_bytes(1, int(avg_ctx)) * (steps / dt) / HBM_PEAK


def source_digest():
    """sha256 over the kernel / runtime sources (mistral.rs_amd/csrc/**, include/*.h), file names included: identifies the BUILD a profile was taken from.  The GPU box
    has no .git, so a commit hash cannot be compared there; the committed profiles carry this digest (scripts/make_profile_summary.py) and bench.py only quotes a
    profile whose digest equals the tree it runs from."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    files = []
    for root, sub in ((os.path.join(here, "mistral.rs_amd", "csrc"), True), (os.path.join(here, "include"), False)):
        for dp, dn, fn in os.walk(root):
            if os.path.basename(dp) == "build":
                dn[:] = []
                continue
            files += [os.path.join(dp, f) for f in fn if f.endswith((".hip", ".cuh", ".cpp", ".h"))]
            if not sub:
                break
    for f in sorted(files):
        h.update(os.path.relpath(f, here).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_profile(model_name, quant="q4_k_m"):
    """Side fields of `roofline` from the committed rocprofv3 passes of THIS build (profiles/round6_{hbm_traffic,kernel_stats}.json, scripts/profile_round.sh ->
    scripts/make_profile_summary.py): HBM bytes per launch of the dominant kernel (--pmc FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, separate passes) and
    its average duration INSIDE the captured decode graph (kernel trace; includes the launch boundary).  Counters cannot be read from inside the timed process.  A
    profile is only quoted when its recorded source digest equals source_digest() of the running tree: otherwise both are null."""
    here = os.path.dirname(os.path.abspath(__file__))
    out = {"traffic": None, "in_graph": None}
    if "8B" not in model_name or quant != "q4_k_m":  # the profiled workload is the Q4_K_M model; ISQ Q8_0 streams twice the bytes
        return out
    dig = source_digest()
    pick = lambda ks: next(v for name, v in ks.items() if "dec_gemv_kernel<1, 2" in name.replace("(int)", "").replace("(mrs::dec::)", ""))  # NCOLS = 1, EPI_GLU
    try:
        j = json.load(open(os.path.join(here, "profiles/round6_hbm_traffic.json")))
        if j.get("source_digest") == dig:
            k = pick(j["kernels"])
            out["traffic"] = int(k["read_bytes_per_launch"] + k["write_bytes_per_launch"])
            out["traffic_source"] = "profiles/round6_hbm_traffic.json"
        else:
            out["traffic_note"] = f"profiles/round6_hbm_traffic.json is of build {j.get('source_digest')}, this tree is {dig}: not quoted"
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    try:
        j = json.load(open(os.path.join(here, "profiles/round6_kernel_stats.json")))
        if j.get("source_digest") == dig:
            kk = pick(j["kernels"])
            out["in_graph"] = {"us_per_launch": round(kk["avg_us"], 2), "source": "profiles/round6_kernel_stats.json", "source_digest": dig, "commit": j.get("commit")}
        else:
            out["in_graph_note"] = f"profiles/round6_kernel_stats.json is of build {j.get('source_digest')}, this tree is {dig}: not quoted"
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    return out


def timed_run(model, cfg, prompt_len, steps, warmup, batch, sync, world, dev):
    """One benchmark pass on an already built model: TTFT of a `prompt_len` prompt (reference method: prompt_len / time-to-first-token, host read-back of
    the first token included), then the HIP graph of one decode step replayed `warmup` + `steps` times, EXACTLY `steps` of them timed between
    barrier + synchronize on both sides (max over ranks)."""
    import torch
    import torch.distributed as dist
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(prompt_len)]
    model.prefill(prompt, 0)  # warm-up (lazy code-object loads, workspace allocation); the timed run overwrites the same pages
    sync()
    t0 = time.perf_counter()
    last = model.prefill(prompt, 0)
    first_tok = int(last.argmax())  # device -> host read-back of the first token: end of TTFT
    ttft = time.perf_counter() - t0
    B = max(1, min(8, batch))
    for sq in range(1, B):  # the other sequences of a batched run: same prompt into their own pages (untimed)
        model.prefill(prompt, 0, seq=sq)
    # ---------------- decode: HIP graph of one step, replayed
    model.set_state([first_tok] * B, [prompt_len] * B)
    model.step_counter.zero_()
    model.capture_decode_graph(B)
    for _ in range(warmup):
        model.replay()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        model.replay()
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    dev_s = ev0.elapsed_time(ev1) / 1e3
    tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    toks = model.tokens_out[0, : warmup + steps].cpu().numpy()
    assert int(model.positions[0]) == prompt_len + warmup + steps, "decode state did not advance as expected"
    return {"prompt": prompt, "ttft": ttft, "prefill_flops": model.prefill_flops(prompt_len), "B": B, "t_all": float(tmax.item()), "dev_s": dev_s, "toks": toks,
            "first_tok": first_tok}


def parity_leg(model, cfg, a, prefill_exact):
    """The parity record of the bench line: the SAME model on the GPU and on the host cores, a `--parity-prompt` (64) token prompt prefilled in ONE pass on both
    sides followed by `--parity-positions` (16) decode steps teacher-forced on the CPU run's greedy tokens.
      (1) engine vs the engine-order restatement (oracle mode "engine"): logits of the last prompt position and of every decode position, and the K / V pages the
          prefill wrote, must be IDENTICAL; the pages a token-by-token decode of the prompt writes must equal the prefill's bit for bit;
      (2) distances: engine / cpu order a / cpu order b each against the EXACT model (dequantized weights, no activation quantization, f64), per position
          (`vs_exact`): is the engine an equally valid sample of the reference arithmetic, not merely as far from a as b is;
      (3) engine vs cpu order a and the a-vs-b calibration (as in earlier rounds), argmax agreement, free-running greedy ids."""
    import numpy as np
    import torch
    P, D = a.parity_prompt, a.parity_positions
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(P)]
    base, fed, cl, runs = cpu_baseline(model, cfg, prompt, positions=D)
    nb = (P + cfg.block_size - 1) // cfg.block_size

    def pages():  # K / V of positions [0, P) of sequence 0 as [P, KVH, hd] bf16 bit patterns, every layer
        ks, vs = [], []
        for kc, vc in zip(model.key_caches, model.value_caches):
            blk = model.block_tables[0, :nb].long()
            k = kc[blk].permute(0, 3, 1, 2, 4).reshape(nb * cfg.block_size, cfg.num_kv_heads, cfg.head_dim)[:P]
            v = vc[blk].permute(0, 3, 1, 2).reshape(nb * cfg.block_size, cfg.num_kv_heads, cfg.head_dim)[:P]
            ks.append(k.contiguous().view(torch.int16).cpu().numpy())
            vs.append(v.contiguous().view(torch.int16).cpu().numpy())
        return np.stack(ks), np.stack(vs)

    f32 = lambda t: t.float().cpu().numpy()
    gl = [f32(model.prefill(prompt, 0))]
    kp, vp = pages()
    for i, t in enumerate(fed):
        model.set_state([t], [P + i])
        gl.append(f32(model.forward_logits(1)[0]))
    # the same prompt token by token through the decode engine, into the same pages
    for pos, t in enumerate(prompt):
        model.set_state([t], [pos])
        ld = f32(model.forward_logits(1)[0])
    kd, vd = pages()
    from oracle import oracle as O
    ref = runs["engine"]
    ko = np.stack([O.to_bf16_bits(np.stack(ref.k[l][:P])).view(np.int16) for l in range(cfg.num_layers)])
    vo = np.stack([O.to_bf16_bits(np.stack(ref.v[l][:P])).view(np.int16) for l in range(cfg.num_layers)])
    rel = lambda x, y, r: float(np.abs(x - y).max() / np.abs(r).max())
    ex = cl["exact"]
    e_x = [rel(g, c, c) for g, c in zip(gl, ex)]
    a_x = [rel(g, c, c) for g, c in zip(cl["cpu"], ex)]
    b_x = [rel(g, c, c) for g, c in zip(cl["cpu_b"], ex)]
    e_a = [rel(g, c, c) for g, c in zip(gl, cl["cpu"])]
    e_b = [rel(g, c, ca) for g, c, ca in zip(gl, cl["cpu_b"], cl["cpu"])]
    a_b = [rel(c, d, c) for c, d in zip(cl["cpu"], cl["cpu_b"])]
    ident = [bool(np.array_equal(g, c)) for g, c in zip(gl, cl["engine"])]
    e_e = [float(np.abs(g - c).max()) for g, c in zip(gl, cl["engine"])]
    ids = lambda ls: [int(x.argmax()) for x in ls]
    ratio = [e / max(x, y, 1e-12) for e, x, y in zip(e_x, a_x, b_x)]
    flips = [{"position": P - 1 + p, "cpu_a_top2_margin_over_max_logit": float((np.sort(c)[-1] - np.sort(c)[-2]) / np.abs(c).max()), "engine_vs_cpu_a": e_a[p]}
             for p, (g, c) in enumerate(zip(gl, cl["cpu"])) if int(g.argmax()) != int(c.argmax())]
    # free-running greedy from the prefilled prompt: engine vs a free-running CPU-order-a run (continues from runs["cpu"]'s prompt cache)
    cpu_a = runs["cpu"]
    for l in range(cfg.num_layers):
        del cpu_a.k[l][P:], cpu_a.v[l][P:]
    tg = tc = int(gl[0].argmax())
    tc = int(cl["cpu"][0].argmax())
    gpu_toks, cpu_toks = [tg], [tc]
    model.prefill(prompt, 0)
    for i in range(D):
        model.set_state([tg], [P + i])
        tg = int(model.forward_logits(1)[0].argmax())
        gpu_toks.append(tg)
        if gpu_toks[:-1] == cpu_toks:  # the CPU run only has to continue while the prefixes agree
            tc = int(cpu_a.step(tc, P + i).argmax())
            cpu_toks.append(tc)
    n_cmp = len(cpu_toks)
    rnd = lambda v: [round(x, 6) for x in v]
    try:
        flip = first_flip_leg(model, cfg, runs["cpu"].w)
    except Exception as e:  # reported extra
        flip = {"error": repr(e)}
    out = {
        "greedy_match": gpu_toks[:n_cmp] == cpu_toks and n_cmp == D + 1,  # vs CPU order a (the reference's own summation orders), free-running after the shared prompt
        "greedy_match_engine_order_restatement": bool(all(ident)),  # identical logits => identical ids: the engine vs the CPU evaluation of the same arithmetic in its order
        "greedy_match_between_cpu_orders": ids(cl["cpu_b"]) == ids(cl["cpu"]),  # calibration: do two CPU summation orders pick the same ids on this model (teacher-forced)
        "parity": {
            "weights": "N(0, 0.02^2) per tensor through the GGML quantizers (device ISQ, bit-identical to GGML), Q4_K_M type map" if a.weights == "gaussian" else "random valid block bytes",
            "prompt_tokens": P, "decode_positions": D, "positions": "index 0 = last prompt position, then one per decode step",
            "prompt_arithmetic": "decode engine's (mrs_gemm_qi + per-query decode attention)" if prefill_exact else "bf16-operand MFMA (NOT the reference arithmetic)",
            "oracle_pinned": False,
            "engine_vs_engine_order_restatement": {
                "bit_identical_positions": int(sum(ident)), "of": len(ident), "max_abs_logit_diff": max(e_e), "greedy_ids_identical": ids(gl) == ids(cl["engine"]),
                "kv_pages_prefill_equal_oracle": bool(np.array_equal(kp, ko) and np.array_equal(vp, vo)),
                "kv_pages_prefill_equal_token_by_token_decode": bool(np.array_equal(kp, kd) and np.array_equal(vp, vd)),
                "prefill_logits_equal_token_by_token_decode": bool(np.array_equal(gl[0], ld)),
                "what": "oracle/cpu_path_oracle.c: the reference CPU path's arithmetic (Q8_K activations, integer block dots, candle rms_norm, single_q.rs softmax with fast_exp) in the "
                        "engine's documented f32 summation orders; LlamaRef.prefill == its own step loop"},
            "vs_exact": {"engine": rnd(e_x), "cpu_order_a": rnd(a_x), "cpu_order_b": rnd(b_x), "engine_over_max_cpu": [round(r, 3) for r in ratio],
                         "engine_within_1.05x_of_max_cpu_at_every_position": bool(max(ratio) <= 1.05),
                         "mean": {"engine": round(float(np.mean(e_x)), 6), "cpu_order_a": round(float(np.mean(a_x)), 6), "cpu_order_b": round(float(np.mean(b_x)), 6)},
                         "what": "max |logits - exact| / max |exact| per position; exact = dequantized weights, f32 activations without quantization, f64 accumulation and glue, same bf16 K / V rounding"},
            "teacher_forced_max_logit_error_over_max_logit": {"engine_vs_cpu_order_a": rnd(e_a), "engine_vs_cpu_order_b": rnd(e_b), "cpu_order_a_vs_cpu_order_b": rnd(a_b)},
            "mean": {"engine_vs_cpu_order_a": round(float(np.mean(e_a)), 6), "cpu_order_a_vs_cpu_order_b": round(float(np.mean(a_b)), 6),
                     "ratio": round(float(np.mean(e_a) / max(np.mean(a_b), 1e-12)), 3)},
            "argmax_agree_with_cpu_order_a": {"engine": int(sum(x == y for x, y in zip(ids(gl), ids(cl["cpu"])))), "cpu_order_b": int(sum(x == y for x, y in zip(ids(cl["cpu_b"]), ids(cl["cpu"])))),
                                              "of": len(gl)},
            "argmax_flips": flips[:4],
            "first_flip": flip, "first_position_vs_cpu_order_a": flip.get("first_position_vs_cpu_order_a"), "positions_before_first_flip": flip.get("positions_before_first_flip"),
            "greedy_free_running": {"tokens_compared": n_cmp, "first_difference": next((i for i, (x, y) in enumerate(zip(gpu_toks, cpu_toks)) if x != y), None)},
            "note": "cpu order a = ggml generic 8-lane GEMV order + candle in-order rms sum + full.rs (prompt) / single_q.rs (decode, 1 kv chunk) attention; b = one f32 term per "
                    "superblock + 2 kv chunks. Orders that differ ONLY in f32 summation agree to ~1e-6 until a rounding difference moves one int8 activation quant across a "
                    "rounding step, then sit at the int8 noise floor of this random-weight model (profiles/round3_parity.md); the engine's arithmetic is pinned by the bit-identical "
                    "restatement, its validity as a sample of the reference arithmetic by vs_exact."}}
    return base, out


def first_flip_leg(model, cfg, ref_w, positions=6):
    """the one parity measurement that was still inferred from 2-layer models.  On THIS (32-layer) model:
      * `first_position_vs_cpu_order_a`: a 1-token prompt -- no int8 activation quant can have moved yet between two f32 summation orders except through f32 rounding
        itself -- on the engine and on LlamaRef(mode="cpu") (the reference's own orders): max |logit difference| / max |logit|;
      * `positions_before_first_flip`: the oracle in the ENGINE's order (which the HIP engine equals bit for bit, checked above) and in CPU order a decode the same
        tokens (teacher-forced on order a's greedy ids); every linear's int8 activation quants are compared, position by position: how many positions (and, inside the
        first differing position, how many linears) pass before ONE quant differs, and how many of the quants differ there."""
    import numpy as np
    from mistralrs_amd.llama import rope_tables
    from oracle import llama_ref
    cos, sin = rope_tables(cfg)
    nblk = (cfg.max_context_len + 31) // 32
    ra = llama_ref.LlamaRef(cfg, ref_w, cos, sin, kv_dtype="bf16", mode="cpu")
    re = llama_ref.LlamaRef(cfg, ref_w, cos, sin, kv_dtype="bf16", mode="engine", attn_bpw=1 if nblk <= 64 else (nblk + 63) // 64)
    tok = 1000 % cfg.vocab_size
    g0 = model.prefill([tok], 0).float().cpu().numpy().reshape(-1)
    out = {}
    first = None
    for pos in range(positions):
        ra.trace, re.trace = [], []
        la = np.asarray(ra.step(tok, pos), dtype=np.float32)
        le = np.asarray(re.step(tok, pos), dtype=np.float32)
        if pos == 0:
            out["first_position_vs_cpu_order_a"] = float(np.abs(g0 - la).max() / np.abs(la).max())
            out["first_position_engine_equals_engine_order_restatement"] = bool(np.array_equal(g0, le))
        if first is None:
            for li, ((na, qa), (ne, qe)) in enumerate(zip(ra.trace, re.trace)):
                nd = int((qa != qe).sum())
                if nd:
                    first = {"position": pos, "linears_before_it_in_that_position": li, "of_linears_per_position": len(ra.trace), "tensor": na,
                             "quants_differing": nd, "of_quants": int(qa.size), "largest_quant_step": int(np.abs(qa.astype(np.int16) - qe.astype(np.int16)).max())}
                    break
        out.setdefault("per_position_vs_cpu_order_a", []).append(round(float(np.abs(le - la).max() / np.abs(la).max()), 9))
        tok = int(la.argmax())
    ra.trace = re.trace = None
    out["positions_before_first_flip"] = positions if first is None else first["position"]
    out["first_flip"] = first
    out["what"] = ("engine-order restatement vs CPU order a on the bench model, teacher-forced on order a's ids from a 1-token prompt; per_position = max |logit difference| "
                   "/ max |logit| (engine order vs order a); a flip = one int8 activation quant of one linear differing between the two orders")
    return out


def moe_parity_leg(dev, layers=4, P=16, D=4):
    """Parity record of the configs[4] leg: a `layers`-layer model of Mixtral-8x7B's layer shape (8 experts, top-2, hidden 4096, ffn 14336),
    N(0, 0.02^2) weights through the device ISQ quantizers, on the GPU and on the host cores:
      (1) bit-exact claims, GPU vs GPU: prefill(prompt) == decoding the prompt token by token (last-position logits and every K / V page) -- the prompt runs in the decode
          engine's arithmetic (grouped exact GEMMs over the expert-sorted routes, csrc/ext_gemm_qi.hip);
      (2) distances to the CPU restatement (oracle/llama_ref.py mode "engine"; its router / expert combination is an f64 restatement of SparseMoeBlock::forward, so this is a
          tolerance, not bit identity): last prompt position + D teacher-forced decode positions, max |dlogit| / max |logit| and arg-max agreement."""
    import gc
    import numpy as np
    import torch
    from mistralrs_amd.llama import LlamaConfig, rope_tables
    from oracle import llama_ref, oracle as O
    O.build()
    cfg = LlamaConfig.mixtral_8x7b(max_batch=1, max_context_len=64, max_position_embeddings=8192)
    cfg.num_layers = layers
    model = build_model(cfg, dev, seed=0, max_new_tokens=8, quant="q4_k_m", weights="gaussian")
    exact = bool(model.prefill_is_exact)
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(P)]
    nb = (P + cfg.block_size - 1) // cfg.block_size

    def pages():
        out = []
        for kc, vc in zip(model.key_caches, model.value_caches):
            blk = model.block_tables[0, :nb].long()
            out.append((kc[blk].clone().view(torch.int16), vc[blk].clone().view(torch.int16)))
        return out
    f32 = lambda t: t.float().cpu().numpy()
    lp = f32(model.prefill(prompt, 0))
    pp = pages()
    for pos, t in enumerate(prompt):
        model.set_state([t], [pos])
        ld = f32(model.forward_logits(1)[0])
    pd = pages()
    same_pages = all(bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) for a, b in zip(pp, pd))
    w = host_weights(model)
    cos, sin = rope_tables(cfg)
    O.set_threads(min(64, os.cpu_count() or 1))
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, kv_dtype="bf16", mode="engine", attn_bpw=1)
    for pos, t in enumerate(prompt):
        lc = np.asarray(ref.step(t, pos), dtype=np.float32)
    rel, agree = [float(np.abs(ld - lc).max() / np.abs(lc).max())], [int(ld.argmax() == lc.argmax())]
    tok = int(lc.argmax())
    for i in range(D):
        lc = np.asarray(ref.step(tok, P + i), dtype=np.float32)
        model.set_state([tok], [P + i])
        g = f32(model.forward_logits(1)[0])
        rel.append(float(np.abs(g - lc).max() / np.abs(lc).max()))
        agree.append(int(g.argmax() == lc.argmax()))
        tok = int(lc.argmax())
    out = {"model": f"{layers} layers of Mixtral-8x7B's shape (8 experts, top-2), N(0, 0.02^2) through the device ISQ quantizers (Q4_K_M type map)", "prompt_tokens": P, "decode_positions": D,
           "prompt_arithmetic": "decode engine's (exact)" if exact else "bf16-operand MFMA",
           "prefill_logits_equal_token_by_token_decode": bool(np.array_equal(lp, ld)), "kv_pages_prefill_equal_token_by_token_decode": bool(same_pages),
           "vs_cpu_restatement_max_logit_error_over_max_logit": [round(x, 6) for x in rel], "argmax_agree_with_cpu_restatement": f"{sum(agree)} / {len(agree)}",
           "oracle_pinned": False}
    del model, ref, w
    gc.collect()
    torch.cuda.empty_cache()
    return out


def extra_config(kind, dev, steps=64, warmup=4, prompt_len=512):
    """A second BASELINE.json configuration on the same GPU, after the headline run (N = 1, default flags only): the same timed_run() on a freshly built model.
    kind: "70b" = configs[3]'s model (Llama-3-70B Q4_K_M, 2048-token prompt) on ONE GPU (40 GB of weights); "q8_0_isq" = configs[2] (Llama-3-8B, every linear quantized in situ from bf16 to Q8_0 on the GPU), "mixtral" = configs[4]'s model on ONE GPU
    (Mixtral-8x7B-shaped Q4_K_M, 26 GB: fits one MI355X; the TP = 2 form is `bench.py --model mixtral --gpus 2`)."""
    import gc
    import torch
    from mistralrs_amd.llama import LlamaConfig
    t_build = time.perf_counter()
    max_ctx = (prompt_len + warmup + steps + 2 + 63) // 64 * 64
    if kind == "70b":
        prompt_len = 2048
        max_ctx = (prompt_len + warmup + steps + 2 + 63) // 64 * 64
        cfg = LlamaConfig.llama3_70b(max_batch=1, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        model = build_model(cfg, dev, seed=0, max_new_tokens=warmup + steps + 8, quant="q4_k_m", weights="gaussian")
        name, wdesc = "Llama-3-70B GGUF Q4_K_M on ONE GPU (configs[3]'s model; its TP = 8 form is `bench.py --gpus 8`)", "N(0, 0.02^2) through the device ISQ quantizers"
    elif kind == "mixtral":
        cfg = LlamaConfig.mixtral_8x7b(max_batch=1, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        model = build_model(cfg, dev, seed=0, max_new_tokens=warmup + steps + 8, quant="q4_k_m", weights="gaussian")
        name, wdesc = "Mixtral-8x7B-shaped (8 experts, top-2) GGUF Q4_K_M, TP=1", "N(0, 0.02^2) through the device ISQ quantizers (46.7 B parameters, expert stacks quantized as [8 x 14336, 4096] tensors)"
    else:
        cfg = LlamaConfig.llama3_8b(max_batch=1, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        model = build_model(cfg, dev, seed=0, max_new_tokens=warmup + steps + 8, quant="q8_0_isq", weights="gaussian")
        name, wdesc = "Llama-3-8B ISQ Q8_0 (in situ from bf16, on the GPU), TP=1", "N(0, 0.02^2) bf16 weights quantized by the device ISQ pass"
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    def sync():
        torch.cuda.synchronize()
    r = timed_run(model, cfg, prompt_len, steps, warmup, 1, sync, 1, dev)
    step_bytes = model.decode_bytes(1, int(prompt_len + warmup + steps / 2))
    out = {"workload": f"{name}, {prompt_len} prefill / {steps} decode, batch 1", "weights": wdesc, "decode_tokens_per_sec": round(steps / r["t_all"], 2),
           "ms_per_step": round(1e3 * r["t_all"] / steps, 4), "steps": steps, "warmup": warmup, "step_bytes": int(step_bytes),
           "step_roofline_frac": round(step_bytes * (steps / r["t_all"]) / HBM_PEAK, 4), "prefill_tokens_per_sec": round(prompt_len / r["ttft"], 1),
           "ttft_ms": round(1e3 * r["ttft"], 2), "prefill_arithmetic": "decode engine's (exact)" if model.prefill_is_exact else "bf16-operand MFMA",
           "decode_path": model.decode_path, "build_s": round(t_build, 1)}
    if kind == "70b" and model.prefill_is_exact:  # long prompt: also the selectable bf16-operand path
        def bf16_ttft():
            model.set_prefill_mode(0)
            model.prefill(r["prompt"], 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            int(model.prefill(r["prompt"], 0).argmax())
            return time.perf_counter() - t0
        tb = bf16_ttft()
        out["prefill_bf16"] = {"tokens_per_sec": round(prompt_len / tb, 1), "ttft_ms": round(1e3 * tb, 2), "frac": round(r["prefill_flops"] / tb / MFMA_PEAK, 4),
                               "gemm": "fused block dequant -> bf16 MFMA (the 137 GB bf16 shadow copy is not taken by default next to three other copies of the weights)"}
        # the MI355X-first variant of the same leg: 288 GB of HBM hold the bf16 shadow copy of the 70B model as well (GGUF 41 + decode tiles 43 + MFMA-order panels 44 +
        # shadow 137 GB): built after everything above has been recorded -- if it does not fit, the note says so and nothing else is affected
        try:
            if model.build_bf16_shadow():
                ts = bf16_ttft()
                out["prefill_bf16_shadow"] = {"tokens_per_sec": round(prompt_len / ts, 1), "ttft_ms": round(1e3 * ts, 2), "frac": round(r["prefill_flops"] / ts / MFMA_PEAK, 4),
                                              "gemm": "hipBLASLt on a bf16 shadow copy of the dense linears (137 GB), built after loading", "hbm_in_use_gb": round(torch.cuda.memory_allocated() / 2 ** 30, 1)}
        except Exception as e:  # torch.OutOfMemoryError or a library refusal
            out["prefill_bf16_shadow"] = {"failed": f"{type(e).__name__}: {str(e)[:200]}"}
        try:
            model.drop_bf16_shadow()
        except Exception:
            pass
    del model
    gc.collect()
    torch.cuda.empty_cache()
    if kind == "mixtral":
        try:
            out["parity"] = moe_parity_leg(dev)
        except Exception as e:  # reported extra
            out["parity"] = {"failed": f"{type(e).__name__}: {e}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded concurrently (1..8; BASELINE configs use 1): value = batch * steps / time")
    ap.add_argument("--shard-shapes", type=int, default=0, help="single GPU, no collectives: run ONE rank's shard shapes of a TP = N model (shape smoke test for --gpus N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in launch sequence leg (dropin_tokens_per_sec)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (configs[2] ISQ Q8_0 and the configs[4] model on one GPU)")
    ap.add_argument("--parity-prompt", type=int, default=64, help="prompt length of the CPU-path parity leg (prefilled in one pass on both sides)")
    ap.add_argument("--small", action="store_true", help="tiny config (smoke / CI), not the benchmark")
    ap.add_argument("--tp", action="store_true", help="(default for N > 1) ONE model sharded tensor-parallel over the N GPUs")
    ap.add_argument("--replicas", action="store_true", help="N > 1: N independent replicas (weak scaling) instead of tensor parallelism")
    ap.add_argument("--model", choices=["auto", "8b", "70b", "mixtral"], default="auto",
                    help="auto: 70b (configs[3]) when N == 8, else 8b (configs[1]); mixtral: Mixtral-8x7B-shaped sparse MoE (configs[4]; TP = 2 with --gpus 2)")
    ap.add_argument("--weights", choices=["gaussian", "blocks"], default="gaussian",
                    help="gaussian (default): N(0, 0.02^2) through the device ISQ quantizers (SURVEY 8d); blocks: random valid block bytes (fast variant for pure kernel timing)")
    ap.add_argument("--parity-positions", type=int, default=16, help="greedy positions of the CPU-path parity leg")
    ap.add_argument("--quant", choices=["q4_k_m", "q8_0_isq"], default="q4_k_m",
                    help="q8_0_isq = BASELINE configs[2]: every linear quantized in situ from bf16 weights to Q8_0 on the GPU (mistralrs_amd.isq, role of generate_isq!)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # driver contract: `python bench.py --gpus N` must use N GPUs -- re-launch ourselves as one process per GPU (mistralrs-core/src/distributed.rs:569-795
        # spawns its ranks the same way)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    if os.environ.get("MRS_BENCH_DRY_RUN"):
        # launcher check without GPUs (tests/test_distributed.py): every rank joins a gloo group, rank 0 prints the line's identity fields
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
        tp = world > 1 and not a.replicas
        big = a.model == "70b" or (a.model == "auto" and world == 8 and not a.replicas)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "decode_tokens_per_sec", "dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "scaling": "strong" if tp else "weak",
                              "config": {"workload": ("Llama-3-70B" if big else "Llama-3-8B") + " GGUF Q4_K_M", "parallelism": "tp1" if world == 1 else (f"tp{world}" if tp else f"replicas x{world}")}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: measuring {world} rank(s)", file=sys.stderr)
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.llama import LlamaConfig
    big = a.model == "70b" or (a.model == "auto" and world == 8 and not a.replicas)
    if big and a.prompt_len == 512:
        a.prompt_len = 2048  # configs[3]: 2048 prefill / 256 decode
    BATCH_LEG = (8, 64)  # warm-up / timed steps of the batched_decode side leg: the model is sized for it whatever --steps says
    leg_steps = max(a.warmup + a.steps, sum(BATCH_LEG))
    ctx_needed = a.prompt_len + leg_steps + 2
    max_ctx = (ctx_needed + 63) // 64 * 64
    if a.small:
        cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2, vocab_size=2048,
                          head_dim=64, max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "tiny-llama (smoke)"
    elif a.model == "mixtral":
        cfg = LlamaConfig.mixtral_8x7b(max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "Mixtral-8x7B-shaped (8 experts, top-2)"
    elif big:
        cfg = LlamaConfig.llama3_70b(max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "Llama-3-70B"
    else:
        cfg = LlamaConfig.llama3_8b(max_batch=8, max_context_len=max_ctx, max_position_embeddings=max(8192, max_ctx))
        name = "Llama-3-8B"
    if a.shard_shapes > 1 and world == 1:
        from mistralrs_amd import distributed as D
        cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size = D.local_dims(cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size, a.shard_shapes)
        name += f" (shapes of one rank of TP={a.shard_shapes}, no collectives)"
    tp = world > 1 and not a.replicas
    if tp:  # column / row parallel shards (mistralrs-quant/src/distributed/layers.rs): local heads, kv heads, ffn
        from mistralrs_amd import distributed as D
        cfg.head_dim = cfg.head_dim  # keep the global head_dim
        cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size = D.local_dims(cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size, world)
        cfg.tp_world_size, cfg.tp_rank = world, rank
    model = build_model(cfg, dev, seed=0 if tp else rank, max_new_tokens=leg_steps + 8, tp=(rank, world) if tp else None, quant=a.quant, weights=a.weights)
    comm, p2p = None, None
    if tp:
        from mistralrs_amd import distributed as D
        comm = D.RcclComm(rank, world, dev)
        assert comm.nranks() == world, f"RCCL communicator has {comm.nranks()} ranks, expected {world}"
        model.set_comm(comm)
        # decode-sized all-reduces: one-shot peer-mailbox route (csrc/ext_p2p.hip) when it is available AND agrees with RCCL on a probe vector on
        # every rank; otherwise RCCL carries everything
        p2p, p2p_note = None, "rccl only"
        if not os.environ.get("MRS_NO_P2P"):
            try:
                p2p = D.P2PAllReduce(rank, world, dev)
                probe = (torch.arange(4096, device=dev, dtype=torch.float32) % 97) * (rank + 1)
                a1, a2 = probe.clone(), probe.clone()
                p2p.all_reduce_(a1)
                comm.all_reduce_(a2)
                torch.cuda.synchronize()
                good = torch.tensor([int(torch.equal(a1, a2) and p2p.error() == 0)], device=dev)
            except Exception as e:  # IPC not available on this box
                good, p2p_note = torch.tensor([0], device=dev), f"rccl only (p2p unavailable: {type(e).__name__})"
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if int(good.item()) == 1:
                model.set_p2p(p2p)
                p2p_note = "one-shot peer-mailbox all-reduce over xGMI (ext_p2p.hip) for [1, hidden]; RCCL for the prefill messages"
            else:
                p2p = None
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    r = timed_run(model, cfg, a.prompt_len, a.steps, a.warmup, a.batch, sync, world, dev)
    if tp and p2p is not None:
        # the peer-mailbox route reports a granule that never arrived through its error word (bounded spin, NaN sums): if ANY rank saw one, every rank drops
        # the route and the whole timed pass is repeated on RCCL -- a number measured over NaN sums is not a measurement
        if model.p2p_sync_error():  # MAX over the ranks, every rank detaches in the same step (mrs_hip_ext.h: mrs_llama_check_p2p)
            p2p, p2p_note = None, "rccl only (the p2p route raised its error word during the timed pass: dropped, pass repeated on RCCL)"
            r = timed_run(model, cfg, a.prompt_len, a.steps, a.warmup, a.batch, sync, world, dev)
    prompt, ttft, prefill_flops, B, t_all, dev_s, toks = r["prompt"], r["ttft"], r["prefill_flops"], r["B"], r["t_all"], r["dev_s"], r["toks"]
    prefill_exact = bool(model.prefill_is_exact)
    ttft_bf16, ttft_bf16_fused = None, None
    if prefill_exact and world == 1:  # the same prompt through the bf16-operand MFMA GEMMs + flash attention (selectable: Llama.set_prefill_mode(0) / MRS_PREFILL_EXACT=0)
        def timed_prefill(mode):
            model.set_prefill_mode(mode)
            model.prefill(prompt, 0, seq=min(1, cfg.max_batch - 1))
            sync()
            t0 = time.perf_counter()
            int(model.prefill(prompt, 0, seq=min(1, cfg.max_batch - 1)).argmax())
            return time.perf_counter() - t0
        ttft_bf16 = timed_prefill(0)  # library bf16 GEMMs on the bf16 shadow copy of the weights when the model has one (csrc/ext_gemm_lt.hip), else the fused block dequant
        if model.bf16_shadow:
            ttft_bf16_fused = timed_prefill(2)  # A / B: the fused block-dequant -> bf16 MFMA kernels of rounds 2-5 on the same prompt
        model.set_prefill_mode(-1)

    # ---------------- roofline of the dominant kernel: the decode engine's gate/up phase (RMSNorm + Q8_K quantize + 2 x [ffn, d] GEMV + SiLU*up),
    # timed with HIP events on the launch stream over every layer's weights (>= 1 GB: nothing Infinity-Cache resident)
    import ctypes as C
    from mistralrs_amd import _lib
    ext = _lib.load("ext")

    class Mat(C.Structure):
        _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]
    MP = C.POINTER(Mat)
    ext.mrs_dec_gate_up.argtypes = [MP, MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    h = torch.randn(1, cfg.hidden_size, device=dev)
    act = torch.empty(1, cfg.intermediate_size, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert model.decode_path == "engine", model.decode_path
    layers = []
    moe = cfg.num_experts > 0
    sfx = "_exps" if moe else ""
    for i in range(cfg.num_layers):
        g, u = model._keep[f"blk.{i}.ffn_gate{sfx}.weight"], model._keep[f"blk.{i}.ffn_up{sfx}.weight"]
        layers.append((Mat(model._keep[f"blk.{i}.ffn_gate{sfx}.weight#dec"].data_ptr(), g.dtype.id, g.shape[0], g.shape[1]),
                       Mat(model._keep[f"blk.{i}.ffn_up{sfx}.weight#dec"].data_ptr(), u.dtype.id, u.shape[0], u.shape[1]),
                       model._keep[f"blk.{i}.ffn_norm.weight"], (g.nbytes() + u.nbytes()) * (cfg.num_experts_per_tok / cfg.num_experts if moe else 1)))
    if moe:  # the top-k experts' gate / up rows in ONE launch (mrs_dec_gate_up_topk); expert ids on the device
        ext.mrs_dec_gate_up_topk.argtypes = [MP, MP, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        sel = torch.tensor([1, 5][: cfg.num_experts_per_tok], dtype=torch.int32, device=dev)
        act = torch.empty(cfg.num_experts_per_tok, cfg.intermediate_size, device=dev)

    def gate_up_pass():
        for mg, mu, nw, _ in layers:
            if moe:
                rc = ext.mrs_dec_gate_up_topk(C.byref(mg), C.byref(mu), cfg.intermediate_size, sel.data_ptr(), cfg.num_experts_per_tok, h.data_ptr(), nw.data_ptr(), cfg.rms_eps, 0,
                                              act.data_ptr(), cfg.intermediate_size, st)
            else:
                rc = ext.mrs_dec_gate_up(C.byref(mg), C.byref(mu), cfg.intermediate_size, None, h.data_ptr(), cfg.hidden_size, nw.data_ptr(), cfg.rms_eps, 0,
                                         act.data_ptr(), cfg.intermediate_size, 1, st)
            assert rc == 0
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
    kern_bytes = int(layers[0][3])  # algorithmic bytes per launch: the two GGUF weight tensors (the decode layout holds the same bits + 2.8 % for 8-bit scales)
    achieved = kern_bytes / kern_s

    # ---------------- tensor parallel: cost of the decode all-reduces (2 per layer, [1, hidden] f32) measured on the same communicator
    ar = None
    if tp:
        buf = torch.zeros(cfg.hidden_size, device=dev)
        for _ in range(10):
            comm.all_reduce_(buf)
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            comm.all_reduce_(buf)
        e1.record()
        sync()
        us = e0.elapsed_time(e1) * 1e3 / 200
        ar = {"rccl_us_per_call": round(us, 2), "calls_per_step": 2 * cfg.num_layers, "bytes": cfg.hidden_size * 4, "decode_route": p2p_note}
        if p2p is not None:
            for _ in range(10):
                p2p.all_reduce_(buf)
            sync()
            e0.record()
            for _ in range(200):
                p2p.all_reduce_(buf)
            e1.record()
            sync()
            ar["p2p_us_per_call"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
        used = ar.get("p2p_us_per_call", ar["rccl_us_per_call"])
        ar["frac_of_step"] = round(used * 2 * cfg.num_layers / (1e6 * t_all / a.steps), 4)  # back-to-back launches; in the step they sit inside the captured graph

    avg_ctx = a.prompt_len + a.warmup + a.steps / 2
    step_bytes = model.decode_bytes(B, int(avg_ctx))
    tok_s = (1 if tp else world) * B * a.steps / t_all  # TP: the N GPUs decode ONE sequence (B sequences with --batch)
    out = {
        "metric": "decode_tokens_per_sec", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * t_all / a.steps, 4), "higher_is_better": True, "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": ("q4_k/q6_k weights x q8_k activations" if a.quant == "q4_k_m" else "q8_0 weights x q8_0 activations") + " (int8 dot, f32 accumulate: the reference CPU path's arithmetic)", "data": "synthetic",
        "config": {"workload": f"{name} " + ("GGUF Q4_K_M" if a.quant == "q4_k_m" else "ISQ Q8_0 (in situ from bf16, on the GPU)") + f", TP={world if tp else 1}, {a.prompt_len} prefill / {a.steps} decode, batch {B}, paged KV bf16 (block 32)",
                   "parallelism": "tp1" if world == 1 else (f"tp{world}" if tp else f"replicas x{world}")},
        "prefill_tokens_per_sec": round(a.prompt_len / ttft, 1), "ttft_ms": round(1e3 * ttft, 2),
        "prefill_arithmetic": ("decode engine's: Q8_K activation rows x exact-integer f16 MFMA block dots, one f32 term per superblock (mrs_gemm_qi), decode attention per query "
                               "-- logits and KV pages identical to token-by-token decode" if prefill_exact else "bf16-operand MFMA (fused block dequant) + MFMA flash attention"),
        "prefill_roofline": {"bound": "mfma", "achieved": round(prefill_flops / ttft / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                             "frac": round(prefill_flops / ttft / MFMA_PEAK, 4), "flops": prefill_flops,
                             "note": ("exact-integer f16 MFMA GEMMs on Q8_K activation images (mrs_gemm_qi) + per-query decode attention over the paged cache"
                                      if prefill_exact else "fused block-dequant -> bf16 MFMA GEMMs (mrs_gemm_q_bf16_multi) + MFMA flash attention over the paged cache")
                                     + "; whole prompt incl. the host read-back of the first token"},
        "device_ms_per_step": round(1e3 * dev_s / a.steps, 4),
        "step_bytes": int(step_bytes), "step_roofline_frac": round(step_bytes * (a.steps / t_all) / HBM_PEAK, 4),
        # `frac` is what THIS process measured: HIP events on the launch stream around back-to-back launches of the kernel over every layer's weights (2.1 GB: nothing
        # cache-resident).  The in-graph average of the committed rocprofv3 trace (launch boundary included) is the side object `in_graph`, quoted only when the profile
        # was taken from this very build (source digest).
        "roofline": {"bound": "hbm", "kernel": "dec_gemv_kernel<1, EPI_GLU> (decode engine gate/up phase: RMSNorm + Q8_K quantize + gate/up GEMV + SiLU*up)",
                     "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                     "bytes_per_launch": int(kern_bytes), "us_per_launch": round(kern_s * 1e6, 2),
                     "frac_source": "live: HIP events on the launch stream over back-to-back launches of every layer's weights, this process", "source_digest": source_digest(),
                     **committed_profile(name, a.quant)},
        "greedy_tokens_head": [int(t) for t in toks[a.warmup: a.warmup + 8]],
    }
    rl = out["roofline"]
    if rl.get("in_graph"):
        rl["in_graph"]["achieved"] = round(kern_bytes / (rl["in_graph"]["us_per_launch"] * 1e-6) / 1e9, 1)
        rl["in_graph"]["frac"] = round(kern_bytes / (rl["in_graph"]["us_per_launch"] * 1e-6) / HBM_PEAK, 4)
    if ttft_bf16 is not None:
        out["prefill_bf16"] = {"tokens_per_sec": round(a.prompt_len / ttft_bf16, 1), "ttft_ms": round(1e3 * ttft_bf16, 2), "frac": round(prefill_flops / ttft_bf16 / MFMA_PEAK, 4),
                               "gemm": ("hipBLASLt bf16 x bf16 -> f32 on a bf16 shadow copy of the dense linears (dequantized once at load: 2 bytes per weight)" if model.bf16_shadow
                                        else "fused block dequant -> bf16 MFMA (mrs_gemm_q_bf16_multi)"),
                               **({"fused_dequant_kernels": {"tokens_per_sec": round(a.prompt_len / ttft_bf16_fused, 1), "ttft_ms": round(1e3 * ttft_bf16_fused, 2),
                                                             "frac": round(prefill_flops / ttft_bf16_fused / MFMA_PEAK, 4)}} if ttft_bf16_fused else {}),
                               "note": "same prompt through the selectable bf16-operand path (Llama.set_prefill_mode(0) / MRS_PREFILL_EXACT=0): faster, but its logits and KV pages are "
                                       "only close to (not identical with) the decode engine's -- the default keeps the reference CPU arithmetic"}
    if ar is not None:
        out["allreduce"] = ar
    if world == 1 and B == 1 and not a.no_dropin and not a.small and a.quant == "q4_k_m" and not moe:
        try:
            d_tok, d_frac = dropin_rate(model, cfg, prompt, min(a.steps, 64), dev)
            out["dropin_tokens_per_sec"] = round(d_tok, 2)
            out["dropin_step_roofline_frac"] = round(d_frac, 4)
            out["dropin_note"] = "the reference's own launch sequence through the drop-in C ABI (use_fused = 0: ~16 launches per layer, Q8_1 activations) -- what an unmodified Rust host gets; value / step_roofline_frac above are the MI355X-native engine (C++ runner)"
        except Exception as e:
            out["dropin_tokens_per_sec"] = None
            out["dropin_note"] = f"failed: {e}"
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"], par = parity_leg(model, cfg, a, prefill_exact)
            out.update(par)
        except Exception as e:  # the baseline is a reported extra, never fatal
            import traceback
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}: {traceback.format_exc()[-400:]}"}
    if rank == 0 and world == 1 and not a.no_extra and not a.small and B == 1 and a.quant == "q4_k_m" and not moe and cfg.max_batch >= 8 and not a.shard_shapes:
        # side field, never fatal: the same model decoding 4 and 8 sequences per step (the reference's MMVQ contract: batch 1-8 from one weight pass, mmvq_gguf.cu:724-792;
        # what the scheduler feeds).  `value` stays the batch-1 line BASELINE.json names.
        try:
            bw, bs_ = BATCH_LEG
            out["batched_decode"] = {"note": f"same weights, b sequences per step (same prompt in b sets of pages), {bs_} timed steps after {bw} warm-up steps; tokens/s = b * steps / time"}
            for bb in (2, 4, 8):
                rb = timed_run(model, cfg, a.prompt_len, bs_, bw, bb, sync, world, dev)
                out["batched_decode"][str(bb)] = {"tokens_per_sec": round(bb * bs_ / rb["t_all"], 2), "ms_per_step": round(1e3 * rb["t_all"] / bs_, 4),
                                                  "step_roofline_frac": round(model.decode_bytes(bb, int(a.prompt_len + bw + bs_ / 2)) * (bs_ / rb["t_all"]) / HBM_PEAK, 4)}
        except Exception as e:
            out["batched_decode"] = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not a.no_extra and not a.small and B == 1 and a.quant == "q4_k_m" and a.model in ("auto", "8b") and not a.shard_shapes:
        out["extra_configs"] = {}
        del model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        for kind in ("q8_0_isq", "mixtral", "70b"):
            try:
                out["extra_configs"][kind] = extra_config(kind, dev)
            except Exception as e:
                out["extra_configs"][kind] = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
