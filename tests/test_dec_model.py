"""north_star parity: the decode engine (ext_dec.hip, reference CPU-path arithmetic) through the C++ runner vs the whole-model restatement of the
reference CPU path (oracle/llama_ref.py + oracle/cpu_path_oracle.c).  Round 3: two bars.

(1) BIT EQUALITY with mode="engine": the reference CPU path's arithmetic -- Q8_K / Q8_0 activation quantizers, integer block dots, d_w * d_x products,
    candle's rms_norm expression, the in-tree CPU attention's online softmax with its fast_exp, SiLU, RoPE, residual adds -- evaluated in the f32
    summation orders the kernels document.  Every logit of every position must be identical, hence also every greedy id.
(2) Distance to mode="cpu" (ggml's generic 8-lane GEMV order, candle's in-order rms sum, single_q.rs's tile order): two evaluations that differ in f32
    summation order ONLY agree to ~1e-6 until the first time a rounding difference moves one int8 activation quant across a rounding step, and then
    sit at the int8 noise floor (oracle-only experiment, profiles/round3_parity.md: at 8B layer shapes ANY two orders, cpu vs cpu_fast included, are
    1.4e-2 .. 3e-2 apart from the second token on).  BASELINE's 1e-3 is therefore met before the first flip and unattainable after it for any
    implementation that does not reproduce one specific build's summation order; the tests hold the engine to the spread that two CPU orders show
    on the same model and tokens (factor 1.25 on the means), and to the size of one moved quant (3e-2) on the tiny model.
KV pages: f16 (the reference CPU path's default KV dtype) and bf16 (the dtype of the GPU pipelines), each against the oracle with the same KV rounding."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q4KM = lambda O: dict(embd=O.Q4_K, q=O.Q4_K, k=O.Q4_K, v=O.Q6_K, o=O.Q4_K, gate=O.Q4_K, up=O.Q4_K, down=O.Q6_K, output=O.Q6_K)
Q8 = lambda O: dict(embd=O.Q8_0, q=O.Q8_0, k=O.Q8_0, v=O.Q8_0, o=O.Q8_0, gate=O.Q8_0, up=O.Q8_0, down=O.Q8_0, output=O.Q8_0)
Q5 = lambda O: dict(embd=O.Q5_K, q=O.Q5_K, k=O.Q5_K, v=O.Q5_K, o=O.Q5_K, gate=O.Q5_K, up=O.Q5_K, down=O.Q6_K, output=O.Q6_K)


def _mk(oracle, dev, types, kv_dtype="bf16", heads=4, kvh=2, layers=2, hidden=512, ff=1024, vocab=512, max_batch=4, seed=0, experts=0, max_new=160, max_ctx=192,
        sliding_window=None, rope_interleaved=True):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor
    from mistralrs_amd.llama import Llama, LlamaConfig, rope_tables
    from oracle import llama_ref
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=ff, num_layers=layers, num_heads=heads, num_kv_heads=kvh, vocab_size=vocab, head_dim=128,
                      rope_theta=10000.0, max_position_embeddings=max(256, max_ctx), max_batch=max_batch, max_context_len=max_ctx, decode_engine=True, kv_dtype=kv_dtype,
                      num_experts=experts, num_experts_per_tok=2, sliding_window=sliding_window, rope_interleaved=rope_interleaved)
    w = llama_ref.synth_weights(cfg, types, seed=seed)
    m = Llama(cfg, dev, max_new_tokens=max_new)
    for name, val in w.items():
        if isinstance(val, tuple):
            t, packed = val
            dt = GgmlDType.from_id(t)
            m.set_tensor(name, QTensor.from_numpy(dt, (packed.shape[0], packed.shape[1] // dt.type_size * dt.block_size), packed, dev))
        else:
            m.set_tensor(name, torch.from_numpy(val))
    assert m.decode_path == "engine"
    cos, sin = rope_tables(cfg)
    return cfg, w, m, cos, sin


def _greedy_parity(oracle, m, ref, cfg, steps, bar, exact_frac=None):
    """Greedy decode from token 1000 % vocab: the engine picks the next token, the oracle is fed the same token; logits compared at every step."""
    tok, rels = 1000 % cfg.vocab_size, []
    for pos in range(steps):
        want = ref.step(tok, pos)
        m.set_state([tok], [pos])
        got = m.forward_logits(1)[0].float().cpu().numpy()
        rel = float(np.abs(got - want).max() / np.abs(want).max())
        rels.append(rel)
        assert rel <= bar, f"position {pos}: max |dlogit| = {rel:.3e} * max |logit| (bar {bar:g})"
        top2 = np.sort(want)[-2:]
        if bar <= 1e-3 or top2[1] - top2[0] > 2 * max(rel, 1e-5) * np.abs(want).max():
            assert int(got.argmax()) == int(want.argmax()), f"position {pos}: greedy ids differ"
        tok = int(got.argmax())
    if exact_frac is not None:
        # positions before the first moved quant agree to f32 noise: the engine's arithmetic is the oracle's (afterwards the two KV caches differ
        # by that one step and the rollouts stay ~1e-2 apart)
        assert rels[0] <= 1e-5 and np.sum(np.array(rels) <= 1e-5) >= exact_frac, rels
    return max(rels)


def _bit_parity(m, ref, cfg, steps):
    """Greedy decode from token 1000 % vocab; engine and engine-order oracle must produce identical logits (hence ids) at every position."""
    tok = 1000 % cfg.vocab_size
    for pos in range(steps):
        want = ref.step(tok, pos)
        m.set_state([tok], [pos])
        got = m.forward_logits(1)[0].float().cpu().numpy()
        assert np.array_equal(got, want), f"position {pos}: {int((got != want).sum())} of {got.size} logits differ, max |d| = {float(np.abs(got - want).max()):.3e}"
        tok = int(got.argmax())


@pytest.mark.parametrize("mix,kv", [("q4km", "f16"), ("q4km", "bf16"), ("q8", "f16"), ("q5", "bf16")])
def test_engine_bit_identical_to_engine_order_oracle_tiny_model(oracle, dev, request, mix, kv):
    """The decode engine == LlamaRef(mode="engine") bit for bit: every logit of every greedy position (all four weight-type mixes, both page dtypes)."""
    from oracle import llama_ref
    emu = request.config.getoption("--host-emulation")
    cfg, w, m, cos, sin = _mk(oracle, dev, {"q4km": Q4KM, "q8": Q8, "q5": Q5}[mix](oracle), kv)
    _bit_parity(m, llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype=kv), cfg, 5 if emu else 48)


@pytest.mark.parametrize("mix,kv", [("q4km", "bf16"), ("q8", "f16")])
def test_engine_rotate_half_rope_bit_identical_and_prefill_consistent(oracle, dev, request, mix, kv):
    """Rotate-half ("neox") RoPE, as safetensors Llama / Mistral checkpoints carry it (RotaryEmbedding::forward with is_gpt_neox, layers.rs:2978): the
    engine's qkv phase runs on q / k planes repacked in pair order (rows i, i + hd/2 adjacent: llama.py set_tensor) and writes the results back to the
    model's dim order -- every logit equals LlamaRef(mode="engine") with rope_interleaved = False, and differs from the interleaved model (the test
    exercises the switch); on the device a prompt through the MFMA prefill (rotary kernel in neox mode on the unpermuted weights) leaves pages the
    engine decodes from consistently."""
    from oracle import llama_ref
    emu = request.config.getoption("--host-emulation")
    types = {"q4km": Q4KM, "q8": Q8}[mix](oracle)
    cfg, w, m, cos, sin = _mk(oracle, dev, types, kv, rope_interleaved=False)
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype=kv)
    inter = llama_ref.LlamaRef(type("C", (), {**cfg.__dict__, "rope_interleaved": True})(), w, cos, sin, mode="engine", kv_dtype=kv)
    tok, differs = 1000 % cfg.vocab_size, False
    for pos in range(4 if emu else 40):
        want, other = ref.step(tok, pos), inter.step(tok, pos)
        m.set_state([tok], [pos])
        got = m.forward_logits(1)[0].float().cpu().numpy()
        assert np.array_equal(got, want), f"position {pos}: engine differs from the rotate-half engine-order oracle"
        differs = differs or not np.array_equal(want, other)
        tok = int(got.argmax())
    assert differs, "rotate-half and interleaved RoPE gave the same logits: the switch is not exercised"
    if emu or kv != "bf16":
        return
    import torch
    prompt = [(1000 + 3 * i) % cfg.vocab_size for i in range(40)]
    _, _, m2, _, _ = _mk(oracle, dev, types, kv, rope_interleaved=False)
    _, _, m3, _, _ = _mk(oracle, dev, types, kv, rope_interleaved=False)
    lp = m2.prefill(prompt, 0)
    for pos, t in enumerate(prompt):
        m3.set_state([t], [pos])
        ld = m3.forward_logits(1)[0].clone()
    if m2.prefill_is_exact:
        assert torch.equal(lp.float(), ld.float()), "prefill (decode engine's arithmetic) != token-by-token decode"
    else:
        assert float((lp - ld).abs().max()) <= 5e-2 * float(ld.abs().max())


@pytest.mark.parametrize("mix,kv", [("q4km", "f16"), ("q4km", "bf16"), ("q8", "f16"), ("q5", "bf16")])
def test_engine_vs_cpu_orders_against_the_exact_model_tiny_model(oracle, dev, request, mix, kv):
    """Tiny dims (hidden 512), the engine's GPU logits next to the reference's own summation orders, each measured against the EXACT model (dequantized
    weights, no activation quantization, f64): position 0 agrees with order a to f32 noise (no int8 quant has moved yet), and over the greedy rollout
    the engine's mean distance to the exact model is within 5 % of the larger CPU order's (tests/test_parity_calibration.py explains why the mean, and
    holds the same bar on the restatement without a GPU)."""
    from oracle import llama_ref
    emu = request.config.getoption("--host-emulation")
    steps = 5 if emu else 48
    cfg, w, m, cos, sin = _mk(oracle, dev, {"q4km": Q4KM, "q8": Q8, "q5": Q5}[mix](oracle), kv)
    _vs_exact(m, cfg, w, cos, sin, kv, steps, f"{mix}/{kv}", check=not emu)


def _vs_exact(m, cfg, w, cos, sin, kv, steps, label, check=True, factor=1.05):
    from oracle import llama_ref
    mk = lambda **k: llama_ref.LlamaRef(cfg, w, cos, sin, kv_dtype=kv, **k)
    runs = {"a": mk(mode="cpu"), "b": mk(mode="cpu_fast", n_kv_chunks=2), "x": mk(mode="exact")}
    tok, err = 1000 % cfg.vocab_size, {"engine": [], "a": [], "b": []}
    for pos in range(steps):
        lg = {k: r.step(tok, pos) for k, r in runs.items()}
        m.set_state([tok], [pos])
        lg["engine"] = m.forward_logits(1)[0].float().cpu().numpy()
        scale = np.abs(lg["x"]).max()
        for k in err:
            err[k].append(float(np.abs(lg[k] - lg["x"]).max() / scale))
        if pos == 0:
            assert np.abs(lg["engine"] - lg["a"]).max() <= 1e-5 * scale
        tok = int(lg["a"].argmax())
    mean = {k: float(np.mean(v)) for k, v in err.items()}
    print(f"{label}: mean distance to the exact model over {steps} positions: engine {mean['engine']:.4f}, cpu a {mean['a']:.4f}, cpu b {mean['b']:.4f}")
    if check:
        assert mean["engine"] <= factor * max(mean["a"], mean["b"]), mean


_W8B = {}  # (layers, vocab, seed) -> quantized weights: the GGML quantizer search over 0.5 G weights takes ~1 minute, once per session


def _mk_8b_dims(oracle, dev, kv_dtype, layers=2, vocab=4096, seed=3, max_ctx=256):
    """Llama-3-8B layer shapes (hidden 4096, 32 / 8 heads of 128, ffn 14336), Q4_K_M type mix, SURVEY 8(d) weights: N(0, 0.02^2) through the GGML
    quantizers (the oracle's, ~1 minute for two layers), `layers` layers and a small vocabulary so the CPU oracle finishes in seconds per token."""
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor
    from mistralrs_amd.llama import Llama, LlamaConfig, rope_tables
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_layers=layers, num_heads=32, num_kv_heads=8, vocab_size=vocab, head_dim=128,
                      rope_theta=500000.0, max_position_embeddings=max(512, max_ctx), max_batch=1, max_context_len=max_ctx, decode_engine=True, kv_dtype=kv_dtype)
    d, ff, nq, nkv = 4096, 14336, 4096, 1024
    O = oracle
    w, k = _W8B.get((layers, vocab, seed), {}), [seed]
    have = bool(w)

    def blocks(t, n, kk, scale):
        k[0] += 1
        return (t, O.quantize(t, (np.random.default_rng(k[0]).standard_normal((n, kk), dtype=np.float32) * np.float32(0.02))))
    rng = np.random.default_rng(seed)
    if not have:
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
    _W8B[(layers, vocab, seed)] = w
    m = Llama(cfg, dev, max_new_tokens=8)
    for name, val in w.items():
        if isinstance(val, tuple):
            t, packed = val
            dt = GgmlDType.from_id(t)
            m.set_tensor(name, QTensor.from_numpy(dt, (packed.shape[0], packed.shape[1] // dt.type_size * dt.block_size), packed, dev))
        else:
            m.set_tensor(name, torch.from_numpy(val))
    assert m.decode_path == "engine"
    cos, sin = rope_tables(cfg)
    return cfg, w, m, cos, sin


@pytest.mark.parametrize("kv", ["f16", "bf16"])
def test_north_star_parity_8b_layer_shapes(oracle, dev, request, kv):
    """Llama-3-8B layer shapes (2 layers, N(0, 0.02^2) weights through the GGML quantizers), 24 greedy tokens, teacher-forced on the CPU-order run:
    (1) engine == mode="engine" bit for bit at every position;
    (2) the exact-model calibration: engine, CPU order a (mode "cpu") and CPU order b (mode "cpu_fast" with two kv chunks) each against the exact model
        (dequantized weights, no activation quantization, f64).  All three sit at the int8-activation noise floor (2.5e-2 .. 4.5e-2 of max |logit| on
        this model); the engine's MEAN distance must be within 5 % of the larger CPU mean (oracle-only run of this test's model: 1.030 x with f16 pages,
        0.967 x with bf16 pages), and greedy ids must agree with order a wherever a's top-2 margin exceeds the distances involved."""
    from oracle import llama_ref
    if request.config.getoption("--host-emulation"):
        pytest.skip("8B layer shapes are for the device")
    cfg, w, m, cos, sin = _mk_8b_dims(oracle, dev, kv)
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu", kv_dtype=kv)
    alt = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu_fast", kv_dtype=kv, n_kv_chunks=2)
    mirror = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype=kv)
    truth = llama_ref.LlamaRef(cfg, w, cos, sin, mode="exact", kv_dtype=kv)
    tok, eng, spread, flips, cpu_flips, vx = 1000 % cfg.vocab_size, [], [], 0, 0, {"engine": [], "a": [], "b": []}
    for pos in range(24):
        want, other, exact, tr = ref.step(tok, pos), alt.step(tok, pos), mirror.step(tok, pos), truth.step(tok, pos)
        m.set_state([tok], [pos])
        got = m.forward_logits(1)[0].float().cpu().numpy()
        assert np.array_equal(got, exact), f"position {pos}: engine differs from the engine-order oracle ({int((got != exact).sum())} logits, max {float(np.abs(got - exact).max()):.3e})"
        for k, lg in (("engine", got), ("a", want), ("b", other)):
            vx[k].append(float(np.abs(lg - tr).max() / np.abs(tr).max()))
        scale = np.abs(want).max()
        eng.append(float(np.abs(got - want).max() / scale))
        spread.append(float(np.abs(other - want).max() / scale))
        top2 = np.sort(want)[-2:]
        if top2[1] - top2[0] > 2 * max(eng[-1], spread[-1], 1e-5) * scale:
            assert int(got.argmax()) == int(want.argmax()), f"position {pos}: greedy ids differ outside a near-tie"
        else:
            flips += 1
        if int(other.argmax()) != int(want.argmax()):
            cpu_flips += 1
        tok = int(want.argmax())
    print(f"8B layer shapes / kv {kv}: engine == engine-order oracle at 24 / 24 positions; engine-vs-cpu worst {max(eng):.2e} mean {np.mean(eng):.2e}; "
          f"cpu-vs-cpu_b worst {max(spread):.2e} mean {np.mean(spread):.2e}; near-ties {flips}; positions where the two CPU orders pick different ids: {cpu_flips}")
    mean = {k: float(np.mean(v)) for k, v in vx.items()}
    print(f"mean distance to the exact model: engine {mean['engine']:.4f}, cpu a {mean['a']:.4f}, cpu b {mean['b']:.4f}")
    assert mean["engine"] <= 1.05 * max(mean["a"], mean["b"]), mean


def test_prefill_512_tokens_at_8b_layer_shapes_vs_exact_oracle(oracle, dev, request):
    """A 512-token prompt through the MFMA prefill (fused block-dequant -> bf16 GEMMs, MFMA flash attention over the paged cache) at Llama-3-8B layer
    shapes (2 layers) vs oracle A for the whole prompt: exactly dequantized weights, f32 BLAS matmuls, f64 softmax -- the model the quantized file
    encodes.  The prefill's defined approximation is one bf16 rounding of both GEMM operands (2^-9 relative per product, f32 accumulate): logits of the
    last token within 2e-2 of max |logit|, same greedy id outside a near-tie; and the first decoded token after the prompt (CPU-path arithmetic on the
    KV pages the prefill wrote; int8 activations against exact weights) within 5e-2 of the exact model's next-position logits."""
    import torch
    if request.config.getoption("--host-emulation"):
        pytest.skip("8B layer shapes are for the device")
    from mistralrs_amd.llama import rope_tables
    cfg, w, m, cos, sin = _mk_8b_dims(oracle, dev, "bf16", max_ctx=640)
    T = 512
    prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(T)]
    assert m.prefill_is_exact
    got_engine = m.prefill(prompt, 0).float().cpu().numpy()  # default: the decode engine's arithmetic (int8 activation images, exact-integer MFMA)
    m.set_prefill_mode(0)  # the selectable bf16-operand path, whose defined approximation this test bounds
    got = m.prefill(prompt, 0).float().cpu().numpy()
    O = oracle
    d, H, KVH, hd = cfg.hidden_size, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    deq = lambda name: O.dequantize(w[name][0], w[name][1], w[name][1].shape[1] // O.type_size(w[name][0]) * O.block_size(w[name][0]))

    def forward(tokens):
        n = len(tokens)
        h = O.dequantize(w["token_embd.weight"][0], w["token_embd.weight"][1][np.asarray(tokens)], d)
        pos = np.arange(n, dtype=np.int32)
        for l in range(cfg.num_layers):
            p = f"blk.{l}."
            xn = O.rms_norm(h, w[p + "attn_norm.weight"], cfg.rms_eps)
            q = O.rope((xn @ deq(p + "attn_q.weight").T).reshape(n, H, hd), cos, sin, pos, False)
            k = O.rope((xn @ deq(p + "attn_k.weight").T).reshape(n, KVH, hd), cos, sin, pos, False)
            v = (xn @ deq(p + "attn_v.weight").T).reshape(n, KVH, hd)
            att = O.attention(q, k, v, 1.0 / np.sqrt(hd)).reshape(n, H * hd)
            h = h + att @ deq(p + "attn_output.weight").T
            xn = O.rms_norm(h, w[p + "ffn_norm.weight"], cfg.rms_eps)
            g, u = xn @ deq(p + "ffn_gate.weight").T, xn @ deq(p + "ffn_up.weight").T
            h = h + O.fused_glu(g, u, 0) @ deq(p + "ffn_down.weight").T
        return O.rms_norm(h[-1:], w["output_norm.weight"], cfg.rms_eps) @ deq("output.weight").T

    want = forward(prompt)[0]
    scale = np.abs(want).max()
    err = float(np.abs(got - want).max() / scale)
    top2 = np.sort(want)[-2:]
    print(f"512-token prompt, 8B layer shapes: prefill vs exact oracle {err:.2e} of max |logit|, top-2 margin {(top2[1] - top2[0]) / scale:.2e}")
    assert err <= 2e-2, err
    if top2[1] - top2[0] > 2 * err * scale:
        assert int(got.argmax()) == int(want.argmax())
    # the default (engine-arithmetic) prompt path sits where every int8-activation evaluation of this model sits (decode: 2.5e-2 .. 4.5e-2 of max |logit|,
    # test_north_star_parity_8b_layer_shapes); its bit-level pin is tests/test_prefill_exact.py, this is only the sanity bound against the exact model
    err_e = float(np.abs(got_engine - want).max() / scale)
    print(f"512-token prompt, engine-arithmetic prefill vs exact oracle {err_e:.2e}")
    assert err_e <= 6e-2, err_e
    nxt = int(want.argmax())
    m.set_state([nxt], [T])
    dec = m.forward_logits(1)[0].float().cpu().numpy()
    want2 = forward(prompt + [nxt])[0]
    err2 = float(np.abs(dec - want2).max() / np.abs(want2).max())
    print(f"first decode step after the prompt vs exact oracle: {err2:.2e}")
    assert err2 <= 5e-2, err2


def test_engine_graph_loop_batch_and_chunked_prefill(oracle, dev, request):
    """HIP-graph decode loop == eager loop token for token; a sequence decodes the same alone and inside a batch (bit-exact logits);
    chunked prefill through the batch kernels == token by token."""
    import torch
    if request.config.getoption("--host-emulation"):
        pytest.skip("graph capture needs the device")
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16")
    prompt = [(1000 + i) % cfg.vocab_size for i in range(20)]
    lt = None
    for pos, t in enumerate(prompt):
        m.set_state([t], [pos])
        lt = m.forward_logits(1)[0].clone()
    eager = []
    tok = int(lt.argmax())
    for i in range(24):
        m.set_state([tok], [len(prompt) + i])
        tok = int(m.forward_logits(1)[0].argmax())
        eager.append(tok)
    # graph loop from the same state
    cfg2, w2, m2, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16")
    lc = m2.prefill_chunked(prompt, chunk=4)
    assert torch.equal(lc, lt), "chunked prefill (b = 4 through the engine) != token by token"
    first = int(lc.argmax())
    m2.set_state([first], [len(prompt)])
    m2.step_counter.zero_()
    m2.capture_decode_graph(1)
    for _ in range(24):
        m2.replay()
    torch.cuda.synchronize()
    assert m2.tokens_out[0, :24].tolist() == eager
    # round 6: the captured batch-1 step is the CHAINED one (arg-max in lm_head's epilogue, the next token's embedding row gathered by the sampling launch): the graph
    # relies on the hidden-state buffer between replays -- a host-side state change (set_state) or an eager step in between must refresh it
    assert m2._graph_chained, "the decode engine's batch-1 graph should be the chained step"
    assert torch.equal(m2.logits[0], m.logits[0]), "logits of the last chained step != the eager step's"
    m2.set_state([eager[9]], [len(prompt) + 10])  # rewind both to position + 10 (pages of earlier positions hold the same tokens)
    m2.step_counter.zero_()
    for _ in range(3):
        m2.replay()
    m2.forward_logits(1)  # an eager step between replays: it recomputes the SAME position (the state was advanced by the graph) and clobbers the hidden-state buffer
    for _ in range(3):
        m2.replay()
    torch.cuda.synchronize()
    assert m2.tokens_out[0, :6].tolist() == eager[10:16], (m2.tokens_out[0, :6].tolist(), eager[10:16])
    # batch: sequence 0 alone vs sequences (0, 1) together
    cfg3, w3, m3, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16")
    for pos in range(6):
        m3.set_state([prompt[pos], prompt[pos + 3]], [pos, pos])
        both = m3.forward_logits(2).clone()
        m.set_state([prompt[pos]], [pos])  # m's sequence 0 is rewound: positions < pos hold the same tokens
        alone = m.forward_logits(1)[0]
        assert torch.equal(both[0], alone), f"batched != single at position {pos}"


def test_chained_greedy_step_equals_plain_step(oracle, dev, request):
    """Round 6: the chained batch-1 step (mrs_llama_decode_step_chained: no embedding launch -- the previous step's sampling launch gathered the row --, arg-max in lm_head's
    epilogue, ONE launch for next id + state advance + next embedding row) produces the same logits, tokens and device state as mrs_llama_decode_step, step by step
    (eager launches: runs on the host emulation too; the captured form is test_engine_graph_loop_batch_and_chunked_prefill)."""
    import torch
    n = 3 if request.config.getoption("--host-emulation") else 12
    cfg, w, m1, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16")
    _, _, m2, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16")
    assert m2._chained_ok(1) and not m2._chained_ok(2)
    for m in (m1, m2):
        m.set_state([1000 % cfg.vocab_size], [0])
        m.step_counter.zero_()
    m2._graph_chained = True
    m2._embed_state(1)
    for i in range(n):
        m1.decode_step(1)
        m2._step_for_graph(1)
        assert torch.equal(m1.logits[0], m2.logits[0]), f"step {i}: logits differ"
        for a, b in ((m1.input_ids, m2.input_ids), (m1.positions, m2.positions), (m1.context_lens, m2.context_lens), (m1.slot_mapping, m2.slot_mapping), (m1.step_counter, m2.step_counter)):
            assert torch.equal(a[:1], b[:1]), f"step {i}: device state differs"
    assert m1.tokens_out[0, :n].tolist() == m2.tokens_out[0, :n].tolist()


@pytest.mark.parametrize("mix", ["q4km", "q8"])
def test_engine_batched_steps_build_the_activation_image_once(oracle, dev, request, mix):
    """Batched decode steps (b >= 2: mrs_dec_act_image + the *_img entry points, one image per phase instead of one per GEMV workgroup) give every sequence the
    logits it gets alone (b = 1: the image is built inside the GEMV), bit for bit."""
    import torch
    types = {"q4km": Q4KM, "q8": Q8}[mix](oracle)
    cfg, w, m, cos, sin = _mk(oracle, dev, types, "bf16", max_batch=4)
    _, _, m1, _, _ = _mk(oracle, dev, types, "bf16", max_batch=4)
    npos = 3
    toks = [[(17 * s + 5 * p + 3) % cfg.vocab_size for p in range(npos)] for s in range(4)]
    alone = []
    for s in range(4):  # m1's sequence 0 plays sequence s from position 0 (its pages are overwritten)
        alone.append([])
        for p in range(npos):
            m1.set_state([toks[s][p]], [p])
            alone[s].append(m1.forward_logits(1)[0].clone())
    import ctypes as C
    from mistralrs_amd import _lib
    count = _lib.sym("ext", "mrs_dec_mm_launch_count", [], C.c_ulonglong)
    for b in (4, 3, 2):
        c0 = count()
        for pos in range(npos):
            m.set_state([toks[s][pos] for s in range(b)], [pos] * b)
            both = m.forward_logits(b)
            for s in range(b):
                assert torch.equal(both[s], alone[s][pos]), f"b = {b}: sequence {s} at position {pos} differs from its single-sequence step"
        # (round 6) batches of three and more stream the MFMA-order copy through the matrix cores (csrc/ext_dec_mm.hip): q / k / v, o_proj, gate / up, down per layer + lm_head
        # (Q8_0 weights: o_proj takes the attention kernel's f32 result on the vector-ALU kernel); smaller ones stay on the vector-ALU kernels -- the logits above are the same
        # bits either way
        took = count() - c0
        per_layer = 4 if mix == "q4km" else 3
        assert (took == npos * (per_layer * cfg.num_layers + 1)) if b >= 3 else (took == 0), (b, took)


def test_short_prompt_in_long_context_and_replay_guard(oracle, dev, request):
    """(advisor, round 1) a prompt of <= 16 tokens must prefill when max_context_len > 512 (the v1 / v2 rule of the fallback attention used to refuse
    before the MFMA flash kernel was even tried), and a captured decode graph must not be replayed past max_new_tokens / max_context_len."""
    import torch
    if request.config.getoption("--host-emulation"):
        pytest.skip("graph capture needs the device")
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16", max_ctx=1024, max_new=6, max_batch=1)
    prompt = [(1000 + i) % cfg.vocab_size for i in range(9)]
    lp = m.prefill(prompt, 0)
    ref = []
    cfg2, w2, m2, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16", max_ctx=1024, max_new=6, max_batch=1)
    for pos, t in enumerate(prompt):
        m2.set_state([t], [pos])
        ld = m2.forward_logits(1)[0].clone()
    assert float((lp - ld).abs().max()) <= 5e-2 * float(ld.abs().max())  # bf16 MFMA prefill vs int8-activation decode: two documented approximations
    m.set_state([int(lp.argmax())], [len(prompt)])
    m.step_counter.zero_()
    m.capture_decode_graph(1)
    for _ in range(6):
        m.replay()
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="past max_new_tokens"):
        m.replay()
    from mistralrs_amd.llama import Llama, LlamaConfig
    with pytest.raises(ValueError, match="max_position_embeddings"):
        Llama(LlamaConfig(hidden_size=256, intermediate_size=512, num_layers=1, num_heads=2, num_kv_heads=1, vocab_size=64, head_dim=128, max_position_embeddings=128,
                          max_context_len=256), dev)


def test_sliding_window_decode_and_prefill(oracle, dev, request):
    """Mistral's sliding window (GGUF <arch>.attention.sliding_window; reference: DecodePlan::GatherSdpa over the last W positions, plan.rs:116-138, mask rule
    paged_attention.rs:551-553).  Window 40 on the tiny model, decoded well past the window:
      * engine == the engine-order restatement with the window, bit for bit, at every position (positions below ctx - W are masked in place);
      * vs the reference's formulation (gather the last W positions, CPU attention over them): f32 noise at the first position past the window, 5e-2 overall;
      * a 70-token prompt through the MFMA prefill with the window mask == the same tokens decoded one by one, within the prefill's bf16 tolerance."""
    from oracle import llama_ref
    emu = request.config.getoption("--host-emulation")
    W = 40
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16", max_batch=1, sliding_window=W)
    mirror = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype="bf16")
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu", kv_dtype="bf16")
    full = llama_ref.LlamaRef(type("C", (), {**cfg.__dict__, "sliding_window": None})(), w, cos, sin, mode="engine", kv_dtype="bf16")
    steps = 44 if emu else 120
    tok, worst, differs_from_full = 1000 % cfg.vocab_size, 0.0, False
    for pos in range(steps):
        exact, want, nowin = mirror.step(tok, pos), ref.step(tok, pos), full.step(tok, pos)
        m.set_state([tok], [pos])
        got = m.forward_logits(1)[0].float().cpu().numpy()
        assert np.array_equal(got, exact), f"position {pos}: engine differs from the windowed engine-order oracle"
        worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
        if pos >= W and not np.array_equal(exact, nowin):
            differs_from_full = True
        if pos < W:
            assert np.array_equal(exact, nowin)  # inside the window nothing is masked
        tok = int(got.argmax())
    assert differs_from_full, "the window never changed the result: the test does not exercise it"
    assert worst <= 5e-2, worst  # one or two moved int8 quants on a 512-wide model over 120 positions (3.7e-2 measured on the MI355X)
    import torch
    if emu and not os.environ.get("MRS_TEST_WINDOW_PREFILL"):
        return  # minutes on the host emulation: opt-in there
    cfg2, w2, m2, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16", max_batch=1, sliding_window=W)
    prompt = [(1000 + 3 * i) % cfg.vocab_size for i in range(70)]
    lp = m2.prefill(prompt, 0)
    cfg3, w3, m3, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16", max_batch=1, sliding_window=W)
    for pos, t in enumerate(prompt):
        m3.set_state([t], [pos])
        ld = m3.forward_logits(1)[0].clone()
    print(f"window prefill vs decode: {float((lp - ld).abs().max()) / float(ld.abs().max()):.3e}")
    if m2.prefill_is_exact:
        assert torch.equal(lp.float(), ld.float()), "prefill (decode engine's arithmetic) != token-by-token decode"
    else:
        assert float((lp - ld).abs().max()) <= 5e-2 * float(ld.abs().max())
    # and the decode step after the prefill reads the pages the prefill wrote through the same window
    nxt = int(ld.argmax())
    m2.set_state([nxt], [len(prompt)]); m3.set_state([nxt], [len(prompt)])
    a, b = m2.forward_logits(1)[0], m3.forward_logits(1)[0]
    print(f"first decode after the prompt: {float((a - b).abs().max()) / float(b.abs().max()):.3e}")
    if m2.prefill_is_exact:
        assert torch.equal(a, b), "decode after the prefill differs from decode after a token-by-token pass: the KV pages differ"
    else:  # KV pages written by the bf16 prompt GEMMs vs pages written by the int8 decode GEMVs, read through the same window: 6.0e-2 measured on the MI355X
        assert float((a - b).abs().max()) <= 1e-1 * float(b.abs().max())


def test_engine_mixtral_moe_vs_cpu_path_oracle(oracle, dev, request):
    steps = 3 if request.config.getoption("--host-emulation") else 24
    types = dict(embd=oracle.Q4_K, q=oracle.Q4_K, k=oracle.Q4_K, v=oracle.Q6_K, o=oracle.Q4_K, gate=oracle.Q4_K, up=oracle.Q4_K, down=oracle.Q6_K, output=oracle.Q6_K)
    from oracle import llama_ref
    cfg, w, m, cos, sin = _mk(oracle, dev, types, "bf16", experts=4)
    _vs_exact(m, cfg, w, cos, sin, "bf16", steps, "moe", check=steps >= 10, factor=1.1)  # 24 positions of a 4-expert toy: 1.046 x on the restatement
