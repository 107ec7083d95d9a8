"""MoE prompt path of the C++ runner (moved out of tests/test_llama_runner.py so that it runs after the suites that were already green on the MI355X:
this test has so far only run on the wave64 host emulation -- `pytest -m gpu --host-emulation tests/test_zz_moe_prefill.py`)."""
import numpy as np
import pytest
import torch  # noqa: F401

from tests.test_llama_runner import Q4KM, _mk, _tokens

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T", [24, 9])
def test_mixtral_moe_prefill_matches_decode_path(oracle, dev, T):
    """Prompt of a Mixtral-style model through mrs_llama_prefill: MFMA GEMMs / flash attention for the attention half, and for the MoE FFN the
    route dispatch + grouped expert GEMMs + weighted aggregation (launch_moe_dispatch / launch_moe_grouped_gemm_<t>; reference route:
    FastExpertsWeights prompt path, moe/experts/backends.rs:969-1100) instead of the token-by-token decode kernels.  Last-token logits within
    3e-2 * max|logit| of the decode path and of the whole-model oracle (a flipped near-tie in a router would show up as a larger gap: the seeds
    here have none); K/V pages within a few bf16 steps; decoding continues from the prefilled pages."""
    from oracle import llama_ref
    kw = dict(hd=128, heads=2, kvh=1, hidden=256, ff=512, vocab=512, experts=4, top_k=2, max_batch=4)
    cfg, w, m, cos, sin = _mk(oracle, dev, True, Q4KM(oracle), **kw)
    _, _, md, _, _ = _mk(oracle, dev, True, Q4KM(oracle), **kw)
    toks = _tokens(T, 5)
    last = m.prefill(toks, 0).cpu().numpy()
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="q8_1", kv_dtype="bf16")
    for pos, t in enumerate(toks):
        md.set_state([t], [pos])
        ld = md.forward_logits(1)[0].clone()
        want = ref.step(t, pos)
    ld = ld.cpu().numpy()
    scale = np.abs(want).max()
    assert np.isfinite(last).all()
    assert np.abs(last - ld).max() <= 3e-2 * scale, np.abs(last - ld).max() / scale
    assert np.abs(last - want).max() <= 3e-2 * scale, np.abs(last - want).max() / scale
    for l in range(cfg.num_layers):
        k0, kd = m.key_caches[l].float(), md.key_caches[l].float()
        # layer 0 sees identical inputs up to the GEMM arithmetic (bf16 MFMA vs int8 Q8_1); deeper layers inherit the FFN differences
        assert float((k0 - kd).abs().max()) <= (2.0 ** -5 if l == 0 else 2.0 ** -4) * float(kd.abs().max()), l
    nxt = int(ld.argmax())
    for mm in (m, md):
        mm.set_state([nxt], [T])
    a, b = m.forward_logits(1)[0].cpu().numpy(), md.forward_logits(1)[0].cpu().numpy()
    assert np.abs(a - b).max() <= 3e-2 * np.abs(b).max()
