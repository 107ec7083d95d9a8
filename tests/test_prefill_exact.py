"""Prompts in the decode engine's arithmetic (csrc/ext_gemm_qi.hip, Llama::prefill_exact): a prompt token's attention output, hidden states, logits and KV pages
are BIT FOR BIT what a token-by-token decode produces -- and both equal the engine-order restatement of the reference CPU path (oracle/llama_ref.py
mode="engine": Q8_K activation rows, integer block dots, candle rms_norm, single_q.rs softmax, in the engine's documented f32 orders).
Reference: the CPU prompt path runs every linear of a T-token prompt through the same QMatMul f32 fallback as decode (mistralrs-quant/src/gguf/mod.rs:465-478)."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.test_dec_engine import ATTN2
from tests.test_dec_model import Q4KM, Q5, _mk

VP, CI = C.c_void_p, C.c_int
PA = [VP, VP, VP, VP, VP, VP] + [CI] * 8 + [C.c_float, CI, CI, CI, CI, VP]


def check_attention(O, be, heads, kvh, T, max_ctx, kv_dtype=1, window=0, start=0):
    """mrs_prefill_attention_exact for T prompt tokens at positions start .. start + T - 1 == mrs_dec_attention at each position (same pages), bit for bit."""
    hd, bs = 128, 32
    nq = heads * hd
    rng = np.random.default_rng(heads + kvh + T + max_ctx)
    mbs = (max_ctx + bs - 1) // bs
    nblocks = mbs + 1
    to16 = (lambda a: a.astype(np.float16).view(np.uint16)) if kv_dtype == 0 else O.to_bf16_bits
    kc = be.buf(to16((rng.standard_normal((nblocks, kvh, hd // 8, bs, 8)) * 0.7).astype(np.float32)))
    vc = be.buf(to16(rng.standard_normal((nblocks, kvh, hd, bs)).astype(np.float32)))
    bt = be.buf((rng.permutation(nblocks - 1)[:mbs].astype(np.uint32) + 1).reshape(1, mbs))
    q_np = (rng.standard_normal((T, nq)) * 0.5).astype(np.float32)
    q = be.buf(q_np)
    ctx = np.arange(start + 1, start + T + 1, dtype=np.uint32)
    cl = be.buf(ctx)
    scale = np.float32(1.0 / np.sqrt(np.float32(hd)))
    got = be.buf(np.full((T, nq), np.nan, np.float32))
    rc = be.sym("mrs_prefill_attention_exact", PA, CI)(q.ptr, kc.ptr, vc.ptr, bt.ptr, cl.ptr, got.ptr, T, heads, kvh, hd, bs, nq, kvh * hd * bs, hd * bs, scale, max_ctx, kv_dtype,
                                                         window, start + T if (heads + T) % 2 else 0, be.stream)
    assert rc == 0, rc
    res = got.numpy()
    splits = be.sym("mrs_decode_attention_max_splits", [CI], CI)(max_ctx)
    po, pm, pl = be.buf(np.zeros((1, heads, splits, hd), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32))
    ticket = be.buf(np.zeros(kvh, np.uint32))
    dec = be.sym("mrs_dec_attention", ATTN2, CI)
    for t in sorted(set([0, 1, T // 3, T // 2, T - 2, T - 1]) & set(range(T))):
        one, c1, q1 = be.buf(np.full((1, nq), np.nan, np.float32)), be.buf(ctx[t:t + 1].copy()), be.buf(q_np[t:t + 1].copy())
        assert dec(one.ptr, None, ticket.ptr, po.ptr, pm.ptr, pl.ptr, q1.ptr, kc.ptr, vc.ptr, kvh, scale, bt.ptr, c1.ptr, bs, max_ctx, 1, heads, hd, mbs, nq, kvh * hd * bs, hd * bs,
                   kv_dtype, window, be.stream) == 0
        assert np.array_equal(res[t], one.numpy()[0]), (t, float(np.abs(res[t] - one.numpy()[0]).max()))


@pytest.mark.parametrize("heads,kvh,T,max_ctx,kvd,window,start", [(4, 1, 40, 64, 1, 0, 0), (4, 2, 70, 128, 0, 0, 0), (2, 2, 33, 96, 1, 0, 20), (8, 2, 50, 4096, 1, 0, 0), (4, 1, 90, 128, 1, 40, 0), (8, 1, 37, 64, 1, 0, 0), (16, 2, 21, 4096, 0, 0, 5)])
def test_prompt_attention_equals_decode_attention_host_emulation(oracle, heads, kvh, T, max_ctx, kvd, window, start):
    check_attention(oracle, HostBackend(), heads, kvh, T, max_ctx, kvd, window, start)


@pytest.mark.gpu
@pytest.mark.parametrize("heads,kvh,T,max_ctx,kvd,window,start", [(32, 8, 512, 1024, 1, 0, 0), (32, 8, 300, 4096, 1, 0, 100), (8, 4, 129, 160, 0, 0, 0), (32, 8, 700, 832, 1, 256, 0), (32, 32, 64, 512, 1, 0, 0), (64, 8, 200, 2304, 1, 0, 0)])
def test_prompt_attention_equals_decode_attention_gpu(oracle, dev, heads, kvh, T, max_ctx, kvd, window, start):
    check_attention(oracle, GpuBackend(dev), heads, kvh, T, max_ctx, kvd, window, start)


Q80 = lambda O: dict(embd=O.Q8_0, q=O.Q8_0, k=O.Q8_0, v=O.Q8_0, o=O.Q8_0, gate=O.Q8_0, up=O.Q8_0, down=O.Q8_0, output=O.Q8_0)  # BASELINE configs[2]: every linear Q8_0 (ISQ)
MIXED = lambda O: dict(embd=O.Q4_K, q=O.Q8_0, k=O.Q8_0, v=O.Q8_0, o=O.Q4_K, gate=O.Q4_K, up=O.Q4_K, down=O.Q8_0, output=O.Q6_K)  # both activation formats in one model


@pytest.mark.gpu
@pytest.mark.parametrize("kv,mix", [("bf16", "q4km"), ("f16", "q4km"), ("bf16", "q5"), ("bf16", "q8_0"), ("bf16", "mixed"), ("bf16", "moe4"), ("bf16", "moe4-q8_0")])
def test_prefill_equals_token_by_token_decode_and_engine_order_oracle(oracle, dev, request, kv, mix):
    """2-layer Q4_K_M model (q5: Q5_K linears + Q6_K down / output, round 5; q8_0: every linear Q8_0 = configs[2]; moe4: 4 experts, top-2 = configs[4]'s layer shape, round 6):
    prefill(prompt) leaves the SAME KV pages and returns the SAME logits as decoding the prompt token by token, both equal
    LlamaRef(mode="engine") bit for bit, and decoding on from the prefilled pages stays bit-identical to the oracle (greedy)."""
    import torch
    from oracle import llama_ref
    emu = request.config.getoption("--host-emulation")
    n_prompt, n_more = (9, 2) if emu else (70, 16)
    types = {"q4km": Q4KM, "q5": Q5, "q8_0": Q80, "mixed": MIXED, "moe4": Q4KM, "moe4-q8_0": Q80}[mix](oracle)
    experts = 4 if mix.startswith("moe4") else 0
    cfg, w, m1, cos, sin = _mk(oracle, dev, types, kv, experts=experts)
    _, _, m2, _, _ = _mk(oracle, dev, types, kv, experts=experts)
    assert m1.prefill_is_exact
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="engine", kv_dtype=kv)
    # sparse-MoE layers: the oracle's router / expert combination is an f64 restatement (tests/test_dec_model.py holds the engine to it statistically), not the engine's
    # f32 order -- the bit-exact claim of those cases is prefill == token-by-token decode (logits, KV pages, chunked == one pass, decoding on from the pages)
    bit_oracle = experts == 0
    prompt = [(1000 + 37 * i) % cfg.vocab_size for i in range(n_prompt)]
    lp = m1.prefill(prompt, 0).float().cpu().numpy()
    for pos, t in enumerate(prompt):
        want = ref.step(t, pos) if bit_oracle else None
        m2.set_state([t], [pos])
        ld = m2.forward_logits(1)[0].float().cpu().numpy()
        assert not bit_oracle or np.array_equal(ld, want), f"decode position {pos} differs from the engine-order oracle"
    assert np.array_equal(lp, ld), f"prefill logits differ from token-by-token decode: max |d| = {float(np.abs(lp - ld).max()):.3e}"
    for layer, (k1, k2, v1, v2) in enumerate(zip(m1.key_caches, m2.key_caches, m1.value_caches, m2.value_caches)):
        assert torch.equal(k1.view(torch.int16), k2.view(torch.int16)), f"layer {layer}: K pages written by prefill != pages written by decode"
        assert torch.equal(v1.view(torch.int16), v2.view(torch.int16)), f"layer {layer}: V pages written by prefill != pages written by decode"
    # the same prompt in two chunks (second chunk at start_pos > 0 attends the pages of the first): same pages, same logits
    _, _, m3, _, _ = _mk(oracle, dev, types, kv, experts=experts)
    cut = n_prompt // 2 + 1
    m3.prefill(prompt[:cut], 0)
    lc = m3.prefill(prompt[cut:], cut).float().cpu().numpy()
    assert np.array_equal(lc, lp), "two-chunk prefill differs from the one-pass prefill"
    for layer, (k1, k3, v1, v3) in enumerate(zip(m1.key_caches, m3.key_caches, m1.value_caches, m3.value_caches)):
        assert torch.equal(k1.view(torch.int16), k3.view(torch.int16)) and torch.equal(v1.view(torch.int16), v3.view(torch.int16)), f"layer {layer}: chunked prefill pages differ"
    tok = int(lp.argmax())
    for pos in range(n_prompt, n_prompt + n_more):
        m1.set_state([tok], [pos])
        got = m1.forward_logits(1)[0].float().cpu().numpy()
        if bit_oracle:
            assert np.array_equal(got, ref.step(tok, pos)), f"position {pos} after the prefill differs from the engine-order oracle"
        else:  # the runner that decoded the prompt token by token continues from ITS pages: same logits
            m2.set_state([tok], [pos])
            assert np.array_equal(got, m2.forward_logits(1)[0].float().cpu().numpy()), f"position {pos}: decoding on from prefilled pages differs from decoding on from decoded pages"
        tok = int(got.argmax())
