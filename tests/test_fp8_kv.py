"""FP8 (E4M3) KV cache, cache_dtype 3 of the paged-attention ABI (reshape_and_cache / paged_attention_v1,v2 / gather_kv_cache with
k_scale, v_scale; update_kv_scales_*): reference semantics mistralrs-paged-attn/src/cuda/quantization/fp8/nvidia/quant_utils.cuh
(store fp8_sat(x / scale), load T(float(fp8) * scale)), reshape_and_cache_kernel.cu:42-132, pagedattention.cuh:255-267,396-406,
update_kvscales.cu:46-150.

CPU: the oracle codec (decode table, RNE + saturation, NaN) incl. a cross-check against torch.float8_e4m3fn for in-range values.
GPU: scatter -> codes bit-exact vs the oracle; gather -> values bit-exact; attention v1 / v2 over an fp8 cache vs the oracle attention
run on the dequantized cache (same tolerance as the 16-bit caches); scale update; error paths."""
import numpy as np
import pytest

from tests.util import round_through, torch_dtype


def test_fp8_codec_properties(oracle):
    codes = np.arange(256, dtype=np.uint8)
    vals = oracle.fp8_e4m3_decode(codes)
    assert np.isnan(vals[0x7F]) and np.isnan(vals[0xFF]) and vals[0x7E] == 448.0 and vals[0x08] == 2.0 ** -6 and vals[0x01] == 2.0 ** -9
    finite = ~np.isnan(vals)
    np.testing.assert_array_equal(oracle.fp8_e4m3_encode(vals[finite]), codes[finite])  # every finite code round-trips (incl. -0)
    # ties go to the even code, overflow saturates, NaN keeps its sign
    assert oracle.fp8_e4m3_encode(np.float32(1.0625))[()] == 0x38 and oracle.fp8_e4m3_encode(np.float32(1.1875))[()] == 0x3A
    assert oracle.fp8_e4m3_encode(np.float32(1e9))[()] == 0x7E and oracle.fp8_e4m3_encode(np.float32(-460.0))[()] == 0xFE
    assert oracle.fp8_e4m3_encode(np.float32(np.nan))[()] & 0x7F == 0x7F
    assert oracle.fp8_e4m3_encode(np.float32(2.0 ** -10))[()] == 0x00 and oracle.fp8_e4m3_encode(np.float32(3 * 2.0 ** -10))[()] == 0x02


def test_fp8_codec_matches_torch_in_range(oracle):
    import torch
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch without float8_e4m3fn")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000) * 3.0, rng.standard_normal(2000) * 1e-2, rng.uniform(-440, 440, 2000)]).astype(np.float32)
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    np.testing.assert_array_equal(oracle.fp8_e4m3_encode(x), want)
    np.testing.assert_array_equal(oracle.fp8_e4m3_decode(want), torch.from_numpy(want).view(torch.float8_e4m3fn).float().numpy())


# ------------------------------------------------------------------------------------------------------------------ GPU
def _fp8_cache(rng, nb, kvh, hd, bs):
    """Random fp8 caches (codes without NaN) in the reference layouts: K [nb, kvh, hd/16, bs, 16], V [nb, kvh, hd, bs]."""
    kc = rng.integers(0, 256, size=(nb, kvh, hd // 16, bs, 16)).astype(np.uint8)
    vc = rng.integers(0, 256, size=(nb, kvh, hd, bs)).astype(np.uint8)
    kc[(kc & 0x7F) == 0x7F] = 0x30
    vc[(vc & 0x7F) == 0x7F] = 0x30
    return kc, vc


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("bs", [16, 32])
def test_reshape_and_cache_and_gather_fp8(oracle, dev, dt, bs):
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(bs + len(dt))
    kvh, hd, nb, T = 4, 128, 6, 37
    key = round_through((rng.standard_normal((T, kvh, hd)) * 3.0).astype(np.float32), dt)
    val = round_through((rng.standard_normal((T, kvh, hd)) * 50.0).astype(np.float32), dt)  # some values saturate at 448 * scale
    key[0, 0, :4] = [0.0, -0.0, 1e-9, 1e9]
    ks, vs = np.float32(0.037), np.float32(0.11)
    slots = rng.permutation(nb * bs)[:T].astype(np.int64)
    slots[5] = -1  # padding token: skipped
    td = torch_dtype(dt)
    kct = torch.zeros(nb, kvh, hd // 16, bs, 16, dtype=torch.uint8, device=dev)
    vct = torch.zeros(nb, kvh, hd, bs, dtype=torch.uint8, device=dev)
    kst, vst = torch.tensor([ks], device=dev), torch.tensor([vs], device=dev)
    paged_attn.reshape_and_cache(torch.from_numpy(key).to(dev).to(td), torch.from_numpy(val).to(dev).to(td), kct, vct,
                                 torch.from_numpy(slots).to(dev), kst, vst)
    want_k = np.zeros((nb, kvh, hd // 16, bs, 16), dtype=np.uint8)
    want_v = np.zeros((nb, kvh, hd, bs), dtype=np.uint8)
    kq, vq = oracle.fp8_e4m3_encode(key / ks), oracle.fp8_e4m3_encode(val / vs)
    for t, s in enumerate(slots):
        if s < 0:
            continue
        b, o = divmod(int(s), bs)
        want_k[b, :, :, o, :] = kq[t].reshape(kvh, hd // 16, 16)
        want_v[b, :, :, o] = vq[t]
    np.testing.assert_array_equal(kct.cpu().numpy(), want_k)
    np.testing.assert_array_equal(vct.cpu().numpy(), want_v)
    # gather back: T(float(fp8) * scale), token order of the block table
    # gather walks positions 0..n-1 of a sequence: position p lives in block table[p // bs] at offset p % bs
    seq_tokens = 2 * bs + 3
    kc2, vc2 = _fp8_cache(rng, nb, kvh, hd, bs)
    table = rng.permutation(nb)[:3].astype(np.int32)
    k_out, v_out = paged_attn.gather_kv_cache(torch.from_numpy(kc2).to(dev), torch.from_numpy(vc2).to(dev), torch.from_numpy(table[None]).to(dev),
                                              torch.tensor([0, seq_tokens], dtype=torch.int32, device=dev), td, kst, vst)
    kd = round_through(oracle.fp8_e4m3_decode(kc2) * ks, dt)
    vd = round_through(oracle.fp8_e4m3_decode(vc2) * vs, dt)
    for p in (0, bs - 1, bs, seq_tokens - 1):
        b, o = table[p // bs], p % bs
        np.testing.assert_array_equal(k_out[p].float().cpu().numpy(), kd[b, :, :, o, :].reshape(kvh, hd))
        np.testing.assert_array_equal(v_out[p].float().cpu().numpy(), vd[b, :, :, o])
    with pytest.raises(ValueError, match="k_scale"):
        paged_attn.reshape_and_cache(torch.from_numpy(key).to(dev).to(td), torch.from_numpy(val).to(dev).to(td), kct, vct, torch.from_numpy(slots).to(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("heads,kvh,hd,bs", [(32, 8, 128, 32), (8, 8, 64, 16), (4, 1, 256, 32), (6, 2, 96, 16)])
@pytest.mark.parametrize("force", ["v1", "v2"])
def test_paged_attention_fp8_cache(oracle, dev, dt, heads, kvh, hd, bs, force):
    import torch
    from mistralrs_amd import paged_attn
    rng = np.random.default_rng(heads * 31 + hd + bs)
    ctxs = [1, bs - 1, bs + 1, 700] if force == "v2" else [1, bs, 3 * bs + 5, 130]
    seqs = len(ctxs)
    max_blocks = (max(ctxs) + bs - 1) // bs + 1
    nb = seqs * max_blocks + 2
    kc, vc = _fp8_cache(rng, nb, kvh, hd, bs)
    kc = (kc & 0xBF) | 0x00  # keep |k| < 2 (exponent field < 8) so the logits stay in a sane range
    kc[(kc & 0x7F) == 0x7F] = 0x30
    ks, vs = np.float32(0.5), np.float32(0.02)
    bt = rng.permutation(nb)[: seqs * max_blocks].reshape(seqs, max_blocks).astype(np.int32)
    q = round_through((rng.standard_normal((seqs, heads, hd)) * 1.5).astype(np.float32), dt)
    td = torch_dtype(dt)
    got = paged_attn.paged_attention(torch.from_numpy(q).to(dev).to(td), torch.from_numpy(kc).to(dev), torch.from_numpy(vc).to(dev),
                                     torch.from_numpy(bt).to(dev), torch.from_numpy(np.array(ctxs, dtype=np.int32)).to(dev), max(ctxs),
                                     1.0 / np.sqrt(hd), force=force, k_scale=torch.tensor([ks], device=dev), v_scale=torch.tensor([vs], device=dev))
    kd = round_through(oracle.fp8_e4m3_decode(kc) * ks, dt)   # what the kernel's K / V loads must produce
    vd = round_through(oracle.fp8_e4m3_decode(vc) * vs, dt)
    # oracle attention over an equivalent x = 8 (16-bit) / x = 4 (f32) layout is layout-agnostic: it only indexes [blk, kvh, d, tok]
    kd_std = kd.transpose(0, 1, 2, 4, 3).reshape(nb, kvh, hd, bs)  # [nb, kvh, hd/16, 16, bs] -> [nb, kvh, hd, bs]
    x_std = 16 // {"f32": 4, "f16": 2, "bf16": 2}[dt]
    kd_ref = kd_std.reshape(nb, kvh, hd // x_std, x_std, bs).transpose(0, 1, 2, 4, 3).copy()
    want = oracle.paged_attention_ref(q, kd_ref, vd, bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None,
                                      round_p=(lambda p: round_through(p, dt)) if dt != "f32" else None)
    got = got.float().cpu().numpy()
    pabs = oracle.paged_attention_ref(q, kd_ref, np.abs(vd), bt, ctxs, 1.0 / np.sqrt(hd), 1.0, None, None)
    ulp = {"f32": 2.0 ** -23, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    tol = (2 * ulp + 2e-5) * pabs + ulp * np.abs(want) + 1e-6
    assert np.isfinite(got).all() and (np.abs(got - want) <= tol).all(), float((np.abs(got - want) / tol).max())


@pytest.mark.gpu
def test_update_kv_scales_and_errors(dev):
    import torch
    from mistralrs_amd import paged_attn
    torch.manual_seed(1)
    for td in (torch.float32, torch.float16, torch.bfloat16):
        k = (torch.randn(3, 1000, device=dev) * 5).to(td)
        v = (torch.randn(3, 1000, device=dev) * 0.01).to(td)
        ks, vs = torch.tensor([0.001], device=dev), torch.tensor([1.0], device=dev)
        paged_attn.update_kv_scales(k, v, ks, vs)
        assert np.float32(ks.item()) == np.float32(k.float().abs().max().item()) / np.float32(240.0)   # grows to absmax / 240 (IEEE division)
        assert vs.item() == 1.0                                       # never shrinks
    kc = torch.zeros(2, 1, 8, 8, 16, dtype=torch.uint8, device=dev)
    vc = torch.zeros(2, 1, 128, 8, dtype=torch.uint8, device=dev)
    one = torch.ones(1, device=dev)
    with pytest.raises(ValueError, match="block_size 16 or 32"):
        paged_attn.paged_attention(torch.zeros(1, 1, 128, device=dev), kc, vc, torch.zeros(1, 2, dtype=torch.int32, device=dev),
                                   torch.ones(1, dtype=torch.int32, device=dev), 8, 1.0, k_scale=one, v_scale=one)
