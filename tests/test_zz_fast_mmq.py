"""`mistralrs_amd.gguf.fast_mmq` -- the host mirror of mistralrs-quant/src/gguf/fast_mmq.rs (shared_lhs :528, down_from_glu :636, plain :762,
fused_qkv :768, fused_glu :781, fused_ffn :800, grouped :822) over the MMQ C ABI.  The checks read like the reference's own
(fast_mmq.rs:1583-1703: fused == independent projections; MMQ vs the dequantized matmul) plus the oracle's MMQ restatement; error behaviour
(the reference's `bail!`s) is checked without a GPU.  `-m gpu` bodies also run on the wave64 host emulation: `pytest --host-emulation -m gpu`."""
import numpy as np
import pytest

from tests.test_mmvq import _ids, _qt, _weights
from tests.util import assert_close_accum, round_through, to_np, torch_dtype

TYPES = ["q4_0", "q5_0", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"]


def test_layouts_and_support():
    from mistralrs_amd.gguf import GgmlDType, fast_mmq
    assert fast_mmq.ds_layout_for(GgmlDType.Q4K) == "DS4" and fast_mmq.ds_layout_for(GgmlDType.Q6K) == "D4"
    assert fast_mmq.ds_layout_for(GgmlDType.Q2K) == "D2S6" and fast_mmq.ds_layout_for(GgmlDType.Q5_0) == "D4"
    assert fast_mmq.ds_layout_for(GgmlDType.Q5_1) == "DS4" and fast_mmq.supports(GgmlDType.Q8_0)
    assert fast_mmq.BLOCK_Q8_1_MMQ_SIZE == 144


def test_layout_table_matches_oracle(oracle):
    from mistralrs_amd.gguf import GgmlDType, fast_mmq
    names = {0: "D4", 1: "DS4", 2: "D2S6"}
    for t, tag in oracle.TYPE_NAMES.items():
        assert fast_mmq.ds_layout_for(GgmlDType.from_id(t)) == names[oracle.mmq_layout(t)], tag


def test_errors_without_gpu(oracle):
    """Shape / dtype validation happens before any launch (the reference bails the same way)."""
    import torch
    from mistralrs_amd.gguf import fast_mmq
    t, p = _weights(oracle, "q4_k", 4, 256, seed=1)
    w = _qt("q4_k", 4, 256, p, torch.device("cpu"))
    with pytest.raises(ValueError, match="at least one weight"):
        fast_mmq.shared_lhs([], torch.zeros(1, 256))
    with pytest.raises(ValueError, match="GPU|devices"):
        fast_mmq.plain(w, torch.zeros(1, 256))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TYPES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_plain_and_fused_qkv(oracle, dev, tag, dt):
    import torch
    from mistralrs_amd.gguf import fast_mmq
    k, b, ns = 512, 19, (40, 9, 8)
    ws = [_weights(oracle, tag, n, k, seed=3 + n) for n in ns]
    qts = [_qt(tag, n, k, p, dev) for (t, p), n in zip(ws, ns)]
    x = round_through(oracle.patterned(b * k, 5, 0.3).reshape(b, k), dt)
    xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt)).reshape(1, b, k)  # leading dims are kept (output_shape)
    outs = fast_mmq.fused_qkv(qts[0], qts[1], qts[2], xs)
    y = oracle.quantize_q8_1_mmq(x, oracle.mmq_layout(ws[0][0]))
    for o, (t, p), n, qt in zip(outs, ws, ns, qts):
        want, mag = oracle.matmul_q8_1_mmq(t, p, n, k, y)
        assert o.shape == (1, b, n) and o.dtype == torch_dtype(dt)
        assert_close_accum(to_np(o).reshape(b, n), round_through(want, dt) if dt != "f32" else want, mag, dt, k // 16, f"mmq qkv {tag}")
        assert torch.equal(o, fast_mmq.plain(qt, xs))  # fused == independent projections, same launches: bit-identical
        # and the MMQ route stays within activation-quantization noise of the exact dequantized matmul (fast_mmq.rs:1583-1703 style bound)
        ex = oracle.matmul_exact(t, p, n, k, x)
        assert np.abs(to_np(o).reshape(b, n).astype(np.float64) - ex).max() <= 5e-2 * np.abs(ex).max() + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["q4_k", "q6_k", "q8_0"])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_fused_glu_and_ffn(oracle, dev, tag, dt):
    import torch
    from mistralrs_amd.gguf import fast_mmq
    from mistralrs_amd import ops
    k, ff, b = 256, 512, 11
    (tg, pg), (tu, pu), (td, pd) = _weights(oracle, tag, ff, k, seed=1), _weights(oracle, tag, ff, k, seed=2), _weights(oracle, tag, k, ff, seed=3)
    g, u, d = _qt(tag, ff, k, pg, dev), _qt(tag, ff, k, pu, dev), _qt(tag, k, ff, pd, dev)
    x = round_through(oracle.patterned(b * k, 9, 0.5).reshape(b, k), dt)
    xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt))
    glu = fast_mmq.fused_glu(g, u, xs, 0)
    gate, up = fast_mmq.shared_lhs([g, u], xs)
    assert torch.equal(glu, ops.fused_glu(gate, up, 0))
    out = fast_mmq.fused_ffn(g, u, d, xs, 0)
    assert out.shape == (b, k) and out.dtype == torch_dtype(dt)
    # unfused route: materialised GLU product -> plain MMQ.  Same quantizer input up to the activation's libm ulp -> a few int8 steps
    sep = fast_mmq.plain(d, glu)
    scale = float(sep.float().abs().max())
    assert float((out.float() - sep.float()).abs().max()) <= 5e-3 * scale  # the reference's fused-vs-unfused bound (fast_mmq.rs:1583-1703)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["q4_k", "q6_k"])
def test_grouped_moe(oracle, dev, tag):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_mmq
    experts, n, k, tokens, topk = 4, 24, 256, 7, 2
    t = _ids(oracle)[tag]
    packed = oracle.random_blocks(t, experts * n, k, seed=17)
    w = QTensor(GgmlDType.from_id(t), (experts, n, k), torch.from_numpy(packed.reshape(-1)).to(dev))
    rng = np.random.default_rng(2)
    x = rng.standard_normal((tokens, k)).astype(np.float32)
    ids = np.stack([rng.permutation(experts)[:topk] for _ in range(tokens)]).astype(np.int32)  # [tokens, topk]
    flat = ids.reshape(-1)
    order = np.argsort(flat, kind="stable").astype(np.int32)  # compact expert-sorted rows -> flat assignment
    ids_src = (order // topk).astype(np.int32)
    ids_dst = order
    counts = np.bincount(flat, minlength=experts)
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    tot = tokens * topk
    out = fast_mmq.grouped(w, torch.from_numpy(x).to(dev), torch.from_numpy(ids_src).to(dev), torch.from_numpy(ids_dst).to(dev),
                           torch.from_numpy(bounds).to(dev), tot, int(counts.max()), experts)
    assert out.shape == (tot, n) and out.dtype == torch.float32
    rb = oracle.row_bytes(t, k)
    y_all = oracle.quantize_q8_1_mmq(x, oracle.mmq_layout(t))
    got = to_np(out)
    for a in range(tot):
        tok, e = a // topk, int(flat[a])
        want, mag = oracle.matmul_q8_1_mmq(t, packed.reshape(-1)[e * n * rb:(e + 1) * n * rb], n, k, np.ascontiguousarray(y_all[:, tok:tok + 1]))
        assert_close_accum(got[a:a + 1], want, mag, "f32", k // 16, f"grouped {tag} assignment {a}")
