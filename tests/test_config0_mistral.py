"""BASELINE.json configs[0] as a named, runnable thing: "Mistral-7B-Instruct GGUF Q4_K_M, 128-token prompt greedy decode on the reference CPU path (plumbing)".

A Mistral-7B-shaped checkpoint (hidden 4096, 32 layers, 32 / 8 heads of 128, ffn 14336, vocab 32000, sliding window 4096, rope theta 10000 -- the shape of
Mistral-7B-Instruct-v0.1) is written as a GGUF file with the metadata a llama.cpp conversion carries (architecture "llama", gguf/normal_config.rs:780-796,
891-945), loaded back through mistralrs_amd.gguf.archive (config synthesis + tensor bindings), and run twice on the same 128-token prompt + 8 greedy tokens:
  * the engine on the MI355X (prompt in one pass through the decode engine's arithmetic, then the decode engine token by token),
  * the restatement of the reference CPU path on the host cores, reading its weights from the same file
    (oracle/llama_ref.py: mode "engine" = the CPU arithmetic in the engine's summation orders, must be IDENTICAL; mode "cpu" = ggml / candle orders).
Weights are synthetic (no network): N(0, 0.02^2) per tensor through the device ISQ quantizers (bit-identical to GGML's), llama.cpp's Q4_K_M type map.
MRS_CONFIG0_LAYERS (default 32) shortens the model for quick runs; MRS_CONFIG0_CPU_ORDER=1 adds the run in ggml's / candle's own summation orders (mode "cpu":
twice the host time; the distance it measures is the one bench.py's parity.vs_exact calibrates on the 32-layer bench model)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config0_mistral_7b_q4_k_m_gguf_128_token_prompt_8_greedy(oracle, dev, tmp_path, request):
    import torch
    if request.config.getoption("--host-emulation"):
        pytest.skip("7B shapes are for the device")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import q4_k_m_types
    from mistralrs_amd import isq
    from mistralrs_amd.gguf import GgmlDType, archive
    from mistralrs_amd.llama import rope_tables
    from oracle import llama_ref
    L = int(os.environ.get("MRS_CONFIG0_LAYERS", "32"))
    d, ff, hd, H, KVH, vocab = 4096, 14336, 128, 32, 8, 32000
    md = {"general.architecture": "llama", "general.name": "mistral-7b-instruct-v0.1 (synthetic weights)", "llama.context_length": 32768,
          "llama.embedding_length": d, "llama.block_count": L, "llama.feed_forward_length": ff, "llama.rope.dimension_count": hd,
          "llama.attention.head_count": H, "llama.attention.head_count_kv": KVH, "llama.attention.layer_norm_rms_epsilon": 1e-5,
          "llama.rope.freq_base": 10000.0, "llama.attention.sliding_window": 4096, "llama.vocab_size": vocab, "general.file_type": 15}
    shapes = {"attn_q": (H * hd, d), "attn_k": (KVH * hd, d), "attn_v": (KVH * hd, d), "attn_output": (d, H * hd), "ffn_gate": (ff, d), "ffn_up": (ff, d),
              "ffn_down": (d, ff)}
    tensors = {}
    g = torch.Generator(device="cpu").manual_seed(7)
    for i, (name, t) in enumerate(q4_k_m_types(L).items()):
        n, k = shapes[name.split(".")[2]] if name.startswith("blk.") else (vocab, d)
        gw = torch.Generator(device=dev).manual_seed(1000 + i)
        q = isq.quantize(torch.randn(n, k, device=dev, generator=gw) * 0.02, t)
        tensors[name] = (t, (n, k), q.data.cpu().numpy().reshape(-1))
    for i in range(L):
        for nm in ("attn_norm", "ffn_norm"):
            tensors[f"blk.{i}.{nm}.weight"] = (GgmlDType.F32, (d,), (1.0 + 0.01 * torch.randn(d, generator=g)).numpy().view(np.uint8))
    tensors["output_norm.weight"] = (GgmlDType.F32, (d,), (1.0 + 0.01 * torch.randn(d, generator=g)).numpy().view(np.uint8))
    path = os.path.join(str(tmp_path), "mistral-7b-instruct-q4_k_m.gguf")
    archive.write_gguf(path, md, tensors)
    del tensors
    print(f"wrote {os.path.getsize(path) / 2**30:.2f} GiB GGUF, {L} layers")

    # ---- load: config synthesis + bindings
    m = archive.load_llama(path, dev, max_context_len=192, max_batch=1)
    cfg = m.cfg
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, cfg.vocab_size) == (d, ff, L, H, KVH, hd, vocab)
    assert cfg.sliding_window == 4096 and cfg.rope_theta == 10000.0 and cfg.max_position_embeddings == 32768 and cfg.rope_interleaved
    assert m.decode_path == "engine" and m.prefill_is_exact

    # ---- the oracle reads the same file
    with archive.GgufArchive(path) as ar:
        w = {}
        for name, t in ar.tensors.items():
            raw = ar.tensor_bytes(name).copy()
            w[name] = raw.view(np.float32) if t.dtype == GgmlDType.F32 else (t.dtype.id, raw.reshape(t.shape[0], -1))
    cos, sin = rope_tables(cfg)
    cos, sin = cos[:256], sin[:256]
    eng = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype=cfg.kv_dtype)
    with_cpu = os.environ.get("MRS_CONFIG0_CPU_ORDER") == "1"
    cpu = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu", kv_dtype=cfg.kv_dtype) if with_cpu else None

    prompt = [(1000 + i % 2048) % vocab for i in range(128)]  # the `mistralrs bench` token rule (bench.rs:253-305)
    got = [m.prefill(prompt, 0).float().cpu().numpy()]
    want = [eng.prefill(prompt)]
    ref = [cpu.prefill(prompt)] if with_cpu else []
    toks = [int(got[0].argmax())]
    for i in range(8):
        m.set_state([toks[-1]], [128 + i])
        got.append(m.forward_logits(1)[0].float().cpu().numpy())
        want.append(eng.step(toks[-1], 128 + i))
        if with_cpu:
            ref.append(cpu.step(toks[-1], 128 + i))
        toks.append(int(got[-1].argmax()))
    for p, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), f"position {127 + p}: engine != engine-order restatement ({int((a != b).sum())} logits, max |d| {float(np.abs(a - b).max()):.3e})"
    print(f"configs[0]: 9 / 9 positions bit-identical to the engine-order restatement; greedy ids {toks}")
    if not with_cpu:
        return
    # against the reference's own summation orders: same int8-activation arithmetic, a different f32 order -> the distance of two CPU orders on this random-weight
    # model (bench.py parity.vs_exact), greedy ids equal wherever order a's top-2 margin exceeds it
    rel = [float(np.abs(a - c).max() / np.abs(c).max()) for a, c in zip(got, ref)]
    agree = 0
    for a, c, r in zip(got, ref, rel):
        top2 = np.sort(c)[-2:]
        if top2[1] - top2[0] > 2 * r * np.abs(c).max():
            assert int(a.argmax()) == int(c.argmax())
        agree += int(a.argmax()) == int(c.argmax())
    print(f"vs cpu order a: max {max(rel):.3e} of max |logit|, greedy ids equal at {agree} / 9")
    assert max(rel) < 0.5
