"""GPU parity: ISQ Q8_0 (BASELINE.json configs[2]: in-situ quantization from bf16 weights).

  * the device quantizer reproduces the oracle's quantize_row_q8_0 restatement bit for bit from f32, bf16 and f16 sources;
  * a tiny all-Q8_0 decoder built by ISQ from bf16 weights runs through the runner (decode + MFMA prefill) and matches the
    oracle model that was quantized on the CPU from the same bf16 weights.
"""
import numpy as np
import pytest

from tests.util import round_through

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src", ["f32", "bf16", "f16"])
def test_isq_q8_0_bit_exact(oracle, dev, src):
    import torch
    from mistralrs_amd import isq
    from tests.util import torch_dtype
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((37, 1024)) * rng.uniform(1e-3, 2.0, (37, 1))).astype(np.float32)
    w[5, 64:96] = 0.0  # an all-zero block: d = 0, id = 0
    w = round_through(w, src) if src != "f32" else w
    qt = isq.quantize_q8_0(torch.from_numpy(w).to(dev).to(torch_dtype(src)))
    want = oracle.quantize(oracle.Q8_0, w)
    np.testing.assert_array_equal(qt.data.cpu().numpy().reshape(37, -1), want)
    with pytest.raises(ValueError, match="multiple of the Q8_0 block size"):
        isq.quantize_q8_0(torch.zeros(4, 48, device=dev))


def test_isq_q8_0_model_decode_and_prefill(oracle, dev):
    import torch
    from mistralrs_amd import isq
    from mistralrs_amd.llama import Llama, LlamaConfig, rope_tables
    from oracle import llama_ref
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=512, head_dim=128,
                      rope_theta=10000.0, max_position_embeddings=256, max_batch=2, max_context_len=192)
    rng = np.random.default_rng(11)
    d, ff, nq, nkv = 512, 1024, 512, 256
    shapes = {"token_embd.weight": (512, d), "output.weight": (512, d)}
    for i in range(2):
        for nm, sh in (("attn_q", (nq, d)), ("attn_k", (nkv, d)), ("attn_v", (nkv, d)), ("attn_output", (d, nq)), ("ffn_gate", (ff, d)),
                       ("ffn_up", (ff, d)), ("ffn_down", (d, ff))):
            shapes[f"blk.{i}.{nm}.weight"] = sh
    m = Llama(cfg, dev, max_new_tokens=16)    # weights quantized in situ on the GPU from bf16
    mc = Llama(cfg, dev, max_new_tokens=16)   # same weights quantized by the CPU oracle (the reference's ISQ runs on the CPU)
    from mistralrs_amd.gguf import GgmlDType, QTensor
    w = {}
    for name, sh in shapes.items():
        dense = round_through((rng.standard_normal(sh) * 0.04).astype(np.float32), "bf16")  # "bf16 safetensors"
        m.set_tensor(name, isq.quantize_q8_0(torch.from_numpy(dense).to(dev).to(torch.bfloat16)))
        w[name] = (oracle.Q8_0, oracle.quantize(oracle.Q8_0, dense))
        mc.set_tensor(name, QTensor.from_numpy(GgmlDType.Q8_0, sh, w[name][1], dev))
    for name in [f"blk.{i}.{n}.weight" for i in range(2) for n in ("attn_norm", "ffn_norm")] + ["output_norm.weight"]:
        nw = (1.0 + 0.01 * rng.standard_normal(d)).astype(np.float32)
        m.set_tensor(name, torch.from_numpy(nw))
        mc.set_tensor(name, torch.from_numpy(nw))
        w[name] = nw
    cos, sin = rope_tables(cfg)
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="q8_1", kv_dtype="bf16")
    toks = [(1000 + i) % 512 for i in range(10)]
    for pos, t in enumerate(toks):
        want = ref.step(t, pos)
        m.set_state([t], [pos]); mc.set_state([t], [pos])
        got = m.forward_logits(1)[0]
        assert torch.equal(got, mc.forward_logits(1)[0]), "GPU-ISQ model != CPU-quantized model"
        # vs the oracle: int8 activation-rounding flips move logits by up to ~1e-2 of their range (tests/test_llama_runner.py)
        assert np.abs(got.cpu().numpy() - want).max() <= 3e-2 * np.abs(want).max()
    # MFMA prefill of the same prompt on a fresh cache of sequence slot 1
    last = m.prefill(toks, 0, seq=1).cpu().numpy()
    assert np.abs(last - want).max() <= 3e-2 * np.abs(want).max()
