"""`GgufMatMul` (mistralrs_amd/gguf/matmul.py), the host mirror of the reference's GGUF `QuantMethod` (mistralrs-quant/src/gguf/mod.rs):
dequantize_w bit-exact against the oracle's format restatement, the try_fast_forward dispatch (1..8 -> MMVQ, > 8 -> prompt route), bias,
embedding gather == dequantize + gather (the reference's own test, gguf/mod.rs:815-846, its values and its 1e-6 bound), apply_isq with the
fallback chain, UQFF round trip with a TP shard.  `-m gpu`; also runs on the wave64 host emulation (`pytest --host-emulation -m gpu`)."""
import numpy as np
import pytest

from tests.test_mmvq import _ids, _qt, _weights
from tests.util import round_through, to_np

pytestmark = pytest.mark.gpu
ALL = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"]


@pytest.mark.parametrize("tag", ALL)
def test_dequantize_w_bit_exact(oracle, dev, tag):
    import torch
    from mistralrs_amd.gguf.matmul import GgufMatMul
    n, k = 5, 768 if tag.endswith("_k") else 96
    t, packed = _weights(oracle, tag, n, k, seed=5)
    m = GgufMatMul(_qt(tag, n, k, packed, dev))
    want = oracle.dequantize(t, packed, k).reshape(n, k)
    assert np.array_equal(to_np(m.dequantize_w()), want)
    assert np.array_equal(to_np(m.dequantize_w(torch.bfloat16)), oracle.round_bf16(want))
    assert np.array_equal(to_np(m.dequantize_w(torch.float16)), want.astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("tag", ["q4_k", "q6_k", "q8_0", "q5_0"])
def test_forward_dispatch_and_bias(oracle, dev, tag):
    import torch
    from mistralrs_amd.gguf import fast_gemm, fast_mmq, fast_mmvq
    from mistralrs_amd.gguf.matmul import GgufMatMul
    n, k = 24, 512
    t, packed = _weights(oracle, tag, n, k, seed=8)
    w = _qt(tag, n, k, packed, dev)
    bias = torch.linspace(-1, 1, n, device=dev)
    m, mq = GgufMatMul(w, bias), GgufMatMul(w, None, prompt_route="mmq")
    x3 = torch.from_numpy(oracle.patterned(3 * k, 1, 0.3).reshape(3, k)).to(dev)
    x20 = torch.from_numpy(oracle.patterned(20 * k, 2, 0.3).reshape(20, k)).to(dev)
    assert m.quantized_act_type() is None and m.has_bias() and not mq.has_bias()
    assert torch.equal(m.forward_raw(x3), fast_mmvq.plain(w, x3))            # batch 1..8 -> MMVQ
    assert torch.equal(m.forward(x3), fast_mmvq.plain(w, x3) + bias)        # bias after the matmul
    assert torch.equal(mq.forward_raw(x20), fast_mmq.plain(w, x20))          # batch > 8, reference route
    big = m.forward_raw(x20)
    if fast_gemm.supports(w.dtype):
        assert torch.equal(big, fast_gemm.plain(w, x20))                     # batch > 8, MI355X route (bf16 MFMA)
    ex = oracle.matmul_exact(t, packed, n, k, to_np(x20))
    for got in (big, mq.forward_raw(x20)):
        assert np.abs(to_np(got) - ex).max() <= 5e-2 * np.abs(ex).max()
    xb = x20.to(torch.bfloat16)                                              # 16-bit activations: MMQ keeps the input dtype
    assert m.forward_raw(xb).dtype == torch.bfloat16


def test_embedding_gather_matches_dequantize_gather(oracle, dev):
    """gguf/mod.rs:815-846: quantize -> embedding gather == quantize -> dequantize -> gather, <= 1e-6; values ((i % 37) - 18) / 7."""
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor
    from mistralrs_amd.gguf.matmul import GgufMatMul
    dim, vocab = 256, 8
    vals = ((np.arange(vocab * dim) % 37).astype(np.float32) - 18.0) / 7.0
    for t in (oracle.Q8_0, oracle.Q4_K, oracle.Q6_K):
        packed = oracle.quantize(t, vals.reshape(vocab, dim))
        m = GgufMatMul(QTensor(GgmlDType.from_id(t), (vocab, dim), torch.from_numpy(packed.reshape(-1).copy()).to(dev)))
        ids = torch.tensor([[3, 0, 7], [7, 1, 3]], device=dev)
        got = m.embedding_forward_raw(ids)
        want = m.dequantize_w()[ids.reshape(-1).long()].reshape(2, 3, dim)
        assert got.shape == (2, 3, dim) and float((got - want).abs().max()) <= 1e-6


def test_apply_isq_and_fallback(oracle, dev):
    import torch
    from mistralrs_amd import isq
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.gguf.matmul import GgufMatMul
    n, k = 6, 96  # 96 % 256 != 0: Q4K falls back to Q4_1 (utils/isq.rs:247-287)
    t, packed = _weights(oracle, "q8_0", n, k, seed=2)
    m = GgufMatMul(_qt("q8_0", n, k, packed, dev))
    assert m.plan_isq(GgmlDType.Q4K) == GgmlDType.Q4_1 and m.plan_isq(None) is None and m.apply_isq(None) is m
    m2 = m.apply_isq(GgmlDType.Q4K)
    assert m2.get_qtensor().dtype == GgmlDType.Q4_1
    want = oracle.quantize(oracle.Q4_1, oracle.dequantize(t, packed, k).reshape(n, k))
    assert np.array_equal(m2.get_qtensor().data.cpu().numpy(), want.reshape(-1))
    assert m.apply_isq(GgmlDType.Q8_0) is m


def test_uqff_round_trip_with_shard(oracle, dev, tmp_path):
    import torch
    from mistralrs_amd import uqff
    from mistralrs_amd.distributed import Shard
    from mistralrs_amd.gguf.matmul import GgufMatMul
    n, k = 8, 512
    t, packed = _weights(oracle, "q4_k", n, k, seed=4)
    m = GgufMatMul(_qt("q4_k", n, k, packed, dev), torch.arange(n, dtype=torch.float32, device=dev))
    p = str(tmp_path / "l.uqff")
    uqff.write(p, m.serialize_uqff("blk.0.attn_q"))
    r = uqff.UqffReader(p)
    full = GgufMatMul.from_uqff(r, "blk.0.attn_q", dev)
    assert torch.equal(full.get_qtensor().data, m.get_qtensor().data) and torch.equal(full.b, m.b)
    half = GgufMatMul.from_uqff(r, "blk.0.attn_q", dev, Shard(dim=0, rank=1, world_size=2))
    x = torch.from_numpy(oracle.patterned(2 * k, 4, 0.2).reshape(2, k)).to(dev)
    assert torch.equal(half.forward(x), m.forward(x)[:, n // 2:])  # column-parallel shard == the matching output columns (+ its bias slice)
