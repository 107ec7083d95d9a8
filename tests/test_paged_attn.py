"""GPU parity: paged KV cache ops + paged attention (v1 / v2) vs the oracle.

The reference has NO numeric test for paged_attention v1/v2; the oracle restates
pagedattention.cuh semantics in f64 numpy (oracle.paged_attention_ref).  Bars:
  * reshape_and_cache / gather_kv_cache / copy_blocks: bit-exact (pure data movement);
  * attention: f32 softmax/accumulate vs f64 -> |err| <= 2e-5 * max|V| for f32 I/O, plus one
    storage ulp of the output for f16/bf16 (+ the P-rounding is modelled in the oracle).
"""
import numpy as np
import pytest

from tests.util import round_through, to_np, torch_dtype

pytestmark = pytest.mark.gpu


def _mk_cache(rng, dev, dt, nb, kvh, hd, bs):
    import torch
    from mistralrs_amd import paged_attn
    ks, vs = paged_attn.kv_cache_shapes(nb, kvh, hd, bs, torch_dtype(dt))
    kc = round_through(rng.standard_normal(ks).astype(np.float32), dt)
    vc = round_through(rng.standard_normal(vs).astype(np.float32), dt)
    return kc, vc, torch.from_numpy(kc).to(dev).to(torch_dtype(dt)), torch.from_numpy(vc).to(dev).to(torch_dtype(dt))


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("bs", [16, 32])
def test_reshape_and_cache_gather_roundtrip(oracle, dev, dt, bs):
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(0)
    nb, kvh, hd, T = 9, 4, 128, 50
    kc, vc, kct, vct = _mk_cache(rng, dev, dt, nb, kvh, hd, bs)
    # key/value are slices of a wider fused qkv row: token stride > kvh*hd (backend/mod.rs:39-77)
    wide = round_through(rng.standard_normal((T, 3, kvh, hd)).astype(np.float32), dt)
    wt = torch.from_numpy(wide).to(dev).to(torch_dtype(dt))
    key, value = wt[:, 1], wt[:, 2]
    slots = rng.permutation(nb * bs)[:T].astype(np.int64)
    slots[[3, 17]] = -1  # padding tokens are skipped
    paged_attn.reshape_and_cache(key, value, kct, vct, torch.from_numpy(slots).to(dev))
    oracle.kv_cache_write(kc, vc, wide[:, 1], wide[:, 2], slots)
    np.testing.assert_array_equal(to_np(kct), kc)
    np.testing.assert_array_equal(to_np(vct), vc)
    # gather: 2 sequences laid out through block tables
    bt = np.array([[5, 2, 7, 0], [1, 8, 3, 0]], dtype=np.int32)
    lens = [bs * 2 + 5, bs + 1]
    cu = np.array([0, lens[0], lens[0] + lens[1]], dtype=np.int32)
    ko, vo = paged_attn.gather_kv_cache(kct, vct, torch.from_numpy(bt).to(dev), torch.from_numpy(cu).to(dev), torch_dtype(dt))
    wk = np.concatenate([oracle.kv_cache_gather(kc, vc, bt[s], lens[s])[0] for s in range(2)])
    wv = np.concatenate([oracle.kv_cache_gather(kc, vc, bt[s], lens[s])[1] for s in range(2)])
    np.testing.assert_array_equal(to_np(ko), wk)
    np.testing.assert_array_equal(to_np(vo), wv)


def test_reshape_and_cache_f32_into_bf16(oracle, dev):
    """MI355X-native pair: f32 activations scattered into a bf16 cache (RNE)."""
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(1)
    nb, kvh, hd, bs, T = 4, 2, 64, 32, 20
    kc, vc, kct, vct = _mk_cache(rng, dev, "bf16", nb, kvh, hd, bs)
    k = rng.standard_normal((T, kvh, hd)).astype(np.float32)
    v = rng.standard_normal((T, kvh, hd)).astype(np.float32)
    slots = np.arange(T, dtype=np.int64) + 7
    paged_attn.reshape_and_cache(torch.from_numpy(k).to(dev), torch.from_numpy(v).to(dev), kct, vct, torch.from_numpy(slots).to(dev))
    oracle.kv_cache_write(kc, vc, round_through(k, "bf16"), round_through(v, "bf16"), slots)
    np.testing.assert_array_equal(to_np(kct), kc)
    np.testing.assert_array_equal(to_np(vct), vc)


def test_copy_blocks(dev):
    import torch
    from mistralrs_amd import paged_attn
    g = torch.Generator(device="cpu").manual_seed(0)
    kcs = [torch.randn(6, 2, 8, 16, 8, generator=g).to(dev).to(torch.bfloat16) for _ in range(3)]
    vcs = [torch.randn(6, 2, 64, 16, generator=g).to(dev).to(torch.bfloat16) for _ in range(3)]
    wk, wv = [t.clone() for t in kcs], [t.clone() for t in vcs]
    paged_attn.copy_blocks(kcs, vcs, {0: [3, 5], 2: [1]})
    for a, b in zip(wk + wv, kcs + vcs):
        a[3], a[5], a[1] = a[0].clone(), a[0].clone(), a[2].clone()
        assert torch.equal(a, b)


def _run_attn(oracle, dev, dt, cdt, heads, kvh, hd, bs, ctxs, force, softcap=1.0, alibi=False, sinks=False, seed=0):
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(seed)
    seqs = len(ctxs)
    max_blocks = (max(ctxs) + bs - 1) // bs + 1
    nb = seqs * max_blocks + 3
    kc, vc, kct, vct = _mk_cache(rng, dev, cdt, nb, kvh, hd, bs)
    # poison: slots past each context hold NaN in V (the kernel must zero them, pagedattention.cuh:404-415)
    bt = rng.permutation(nb)[: seqs * max_blocks].reshape(seqs, max_blocks).astype(np.int32)
    for s, c in enumerate(ctxs):
        if c % bs:
            vc[bt[s, c // bs], :, :, c % bs:] = np.nan
    vct = torch.from_numpy(vc).to(dev).to(torch_dtype(cdt))
    q = round_through(rng.standard_normal((seqs, heads, hd)).astype(np.float32) * 2.0, dt)
    qt = torch.from_numpy(q).to(dev).to(torch_dtype(dt))
    # q as a strided view (q_stride > heads*hd), like a slice of a fused qkv output
    qwide = torch.zeros(seqs, heads * hd + 2 * kvh * hd, dtype=qt.dtype, device=dev)
    qwide[:, : heads * hd] = qt.reshape(seqs, -1)
    qv = qwide[:, : heads * hd].view(seqs, heads, hd)
    al = (rng.uniform(0.01, 0.2, heads).astype(np.float32) if alibi else None)
    sk = (rng.standard_normal(heads).astype(np.float32) if sinks else None)
    out = paged_attn.paged_attention(
        qv, kct, vct, torch.from_numpy(bt).to(dev), torch.from_numpy(np.array(ctxs, dtype=np.int32)).to(dev),
        max_context_len=max(ctxs), softmax_scale=1.0 / np.sqrt(hd), softcapping=softcap,
        alibi_slopes=torch.from_numpy(al).to(dev) if alibi else None, sinks=torch.from_numpy(sk).to(dev) if sinks else None,
        force=force)
    vc_clean = np.nan_to_num(vc, nan=0.0)
    want = oracle.paged_attention_ref(q, kc, vc_clean, bt, ctxs, 1.0 / np.sqrt(hd), softcap, al, sk,
                                      round_p=(lambda p: round_through(p, dt)) if dt != "f32" else None)
    got = to_np(out)
    assert np.isfinite(got).all()
    ulp = {"f32": 0.0, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    tol = 3e-5 * np.abs(vc_clean).max() + ulp * np.abs(want) * 1.01
    if force == "v2" and dt != "f32":
        tol = tol + 2 * ulp * np.abs(vc_clean).max() * 0.5  # partials are stored in the query dtype (tmp_out)
    err = np.abs(got - want)
    assert (err <= tol).all(), f"worst {err.max():.3e} tol {tol.flat[err.argmax()] if np.ndim(tol) else tol:.3e}"


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("force", ["v1", "v2"])
def test_paged_attention_llama_shape(oracle, dev, dt, force):
    _run_attn(oracle, dev, dt, dt, heads=32, kvh=8, hd=128, bs=32, ctxs=[1, 33, 640, 1500], force=force)


@pytest.mark.parametrize("heads,kvh", [(8, 8), (8, 4), (8, 1), (6, 2), (12, 4)])
def test_paged_attention_gqa_groupings(oracle, dev, heads, kvh):
    _run_attn(oracle, dev, "bf16", "bf16", heads, kvh, 128, 32, [200, 17], None, seed=heads + kvh)


@pytest.mark.parametrize("hd", [64, 80, 96, 112, 128, 192, 256, 512])
@pytest.mark.parametrize("bs", [8, 16, 32])
def test_paged_attention_head_and_block_sizes(oracle, dev, hd, bs):
    _run_attn(oracle, dev, "f32", "f32", 4, 2, hd, bs, [bs * 3 + 1, 5], "v1", seed=hd)
    _run_attn(oracle, dev, "bf16", "bf16", 4, 2, hd, bs, [530, 5], "v2", seed=hd + 1)


def test_paged_attention_softcap_alibi_sinks(oracle, dev):
    for force in ("v1", "v2"):
        _run_attn(oracle, dev, "f32", "f32", 8, 2, 128, 32, [700, 90], force, softcap=30.0)
        _run_attn(oracle, dev, "f32", "f32", 8, 2, 128, 32, [700, 90], force, alibi=True)
        _run_attn(oracle, dev, "f32", "f32", 8, 2, 128, 32, [700, 90], force, sinks=True)


def test_paged_attention_f32_query_bf16_cache(oracle, dev):
    _run_attn(oracle, dev, "f32", "bf16", 32, 8, 128, 32, [640, 3], "v1")
    _run_attn(oracle, dev, "f32", "bf16", 32, 8, 128, 32, [1200, 3], "v2")


def test_paged_attention_error_behaviour(dev):
    import torch
    from mistralrs_amd import paged_attn
    kc = torch.zeros(2, 2, 16, 32, 8, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros(2, 2, 128, 32, dtype=torch.bfloat16, device=dev)
    bt = torch.zeros(1, 2, dtype=torch.int32, device=dev)
    cl = torch.ones(1, dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="head_size"):
        paged_attn.paged_attention(torch.zeros(1, 4, 100, dtype=torch.bfloat16, device=dev), kc, vc, bt, cl, 32, 1.0)
    with pytest.raises(ValueError, match="block_tables"):
        paged_attn.paged_attention(torch.zeros(2, 4, 128, dtype=torch.bfloat16, device=dev), kc, vc, bt, cl, 32, 1.0)
    with pytest.raises(ValueError, match="context_lens"):
        paged_attn.paged_attention(torch.zeros(1, 4, 128, dtype=torch.bfloat16, device=dev), kc, vc, bt, torch.ones(3, dtype=torch.int32, device=dev), 32, 1.0)


@pytest.mark.parametrize("heads,kvh,hd", [(32, 8, 128), (8, 8, 128), (16, 2, 128), (8, 4, 64)])
def test_decode_attention_q8_1(oracle, dev, heads, kvh, hd):
    """MI355X decode attention (wave-per-KV-chunk split + merge -> Q8_1) vs the oracle attention: the dequantized Q8_1
    output must sit within half a quantisation step (+ the f32-vs-f64 softmax tolerance) of the f64 reference."""
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(heads * 131 + kvh)
    ctxs, bs = [1, 31, 32, 33, 700, 2500], 32
    seqs = len(ctxs)
    max_blocks = (max(ctxs) + bs - 1) // bs + 1
    nb = seqs * max_blocks + 3
    kc, vc, kct, vct = _mk_cache(rng, dev, "bf16", nb, kvh, hd, bs)
    bt = rng.permutation(nb)[: seqs * max_blocks].reshape(seqs, max_blocks).astype(np.int32)
    for s, c in enumerate(ctxs):
        if c % bs:
            vc[bt[s, c // bs], :, :, c % bs:] = np.nan  # stale slots past the context must be ignored
    vct = torch.from_numpy(vc).to(dev).to(torch.bfloat16)
    q = (rng.standard_normal((seqs, heads, hd)) * 2.0).astype(np.float32)
    y, stride = paged_attn.decode_attention_q8_1(torch.from_numpy(q).to(dev), kct, vct, torch.from_numpy(bt).to(dev),
                                                 torch.from_numpy(np.array(ctxs, dtype=np.int32)).to(dev), max(ctxs), 1.0 / np.sqrt(hd))
    want = oracle.paged_attention_ref(q, kc, np.nan_to_num(vc, nan=0.0), bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None,
                                      round_p=lambda p: round_through(p, "bf16")).reshape(seqs, heads * hd)
    blocks = y.cpu().numpy().reshape(seqs, stride, 36)
    d = blocks[:, :, :2].copy().view(np.float16).astype(np.float32)[:, :, 0]
    qs = blocks[:, :, 4:].view(np.int8).astype(np.float32)
    got = (qs * d[:, :, None]).reshape(seqs, -1)[:, : heads * hd]
    dfull = np.repeat(d, 32, axis=1)[:, : heads * hd]
    amax = np.abs(want.reshape(seqs, -1, 32)).max(axis=2)
    np.testing.assert_allclose(d[:, : heads * hd // 32], amax / 127.0, rtol=1e-2, atol=1e-5)  # block scales follow the reference value
    # probabilities are bf16-rounded before P.V (reference semantics) but un-normalised (online softmax), the oracle rounds the
    # normalised ones: each differs from the exact p by <= 2^-9 relative, so the outputs by <= 2^-8 * sum_t p_t |v_t|
    pabs = oracle.paged_attention_ref(q, kc, np.abs(np.nan_to_num(vc, nan=0.0)), bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None).reshape(seqs, heads * hd)
    tol = 0.5 * dfull * 1.01 + 2.0 ** -8 * pabs + 3e-5 * np.abs(np.nan_to_num(vc)).max()
    err = np.abs(got - want)
    bad = err > tol
    assert np.isfinite(got).all() and not bad.any(), (f"{int(bad.sum())} outside tolerance; worst excess {float((err - tol).max()):.3e} at seq "
                                                        f"{np.argwhere(bad)[:4].tolist()} err {err[bad][:4]} tol {tol[bad][:4]} d {dfull[bad][:4]}")


@pytest.mark.parametrize("heads,kvh", [(32, 8), (8, 8), (16, 2), (8, 1)])
@pytest.mark.parametrize("T,start", [(1, 0), (31, 0), (70, 0), (64, 45), (200, 333)])
def test_prefill_attention(oracle, dev, heads, kvh, T, start):
    """MFMA prompt attention over the paged cache vs the oracle (token t = a sequence of context start+t+1 on the same block
    table): q and p rounded to bf16 as the reference's bf16 attention does; stale slots after the sequence end hold NaN."""
    import torch
    from mistralrs_amd import paged_attn
    hd, bs = 128, 32
    rng = np.random.default_rng(heads * 7 + kvh + T * 13 + start)
    total = start + T
    max_blocks = (total + bs - 1) // bs + 1
    nb = max_blocks + 5
    kc, vc, kct, vct = _mk_cache(rng, dev, "bf16", nb, kvh, hd, bs)
    bt1 = rng.permutation(nb)[:max_blocks].astype(np.int32)
    if total % bs:
        vc[bt1[total // bs], :, :, total % bs:] = np.nan
    vc[bt1[total // bs + 1:]] = np.nan
    vct = torch.from_numpy(vc).to(dev).to(torch.bfloat16)
    q = (rng.standard_normal((T, heads, hd)) * 2.0).astype(np.float32)
    got = paged_attn.prefill_attention(torch.from_numpy(q).to(dev), kct, vct, torch.from_numpy(bt1).to(dev), start, 1.0 / np.sqrt(hd)).cpu().numpy()
    ctxs = [start + t + 1 for t in range(T)]
    bt = np.tile(bt1, (T, 1))
    vz = np.nan_to_num(vc, nan=0.0)
    qb = round_through(q, "bf16")
    want = oracle.paged_attention_ref(qb, kc, vz, bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None, round_p=lambda p: round_through(p, "bf16"))
    pabs = oracle.paged_attention_ref(qb, kc, np.abs(vz), bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None)
    # kernel rounds the un-normalised p to bf16, the oracle the normalised p: each within 2^-8 relative (half an ulp of 8 significant
    # bits), so the outputs differ by at most 2^-7 * sum_t p_t |v_t|
    tol = 2.0 ** -7 * pabs + 1e-4 * np.abs(vz).max()
    err = np.abs(got - want.reshape(got.shape))
    tol = tol.reshape(got.shape)
    bad = np.argwhere(err > tol)
    assert np.isfinite(got).all() and len(bad) == 0, (f"{len(bad)} outside tolerance, first {bad[:6].tolist()} err {err[err > tol][:6]} "
                                                       f"tol {tol[err > tol][:6]} got {got[err > tol][:6]}")
    with pytest.raises(ValueError, match="block_table"):
        paged_attn.prefill_attention(torch.from_numpy(q).to(dev), kct, vct, torch.from_numpy(bt1[:1]).to(dev), start + 64, 1.0)
