"""CPU: the oracle against independent restatements and the reference tests' own invariants.

No golden vectors exist in the reference tree; what can be pinned on CPU:
  * fp16/bf16 conversion vs numpy for all 65536 bit patterns;
  * block decode vs a second, independent numpy restatement written from the layout table
    (SURVEY appendix A), so a transcription slip in either shows up;
  * quantize->dequantize error bands of the public GGML quantizers;
  * A (exact) / B (candle CPU) / C (GPU Q8_1) matmul oracles agree within the activation-
    quantization error band, and the frozen fixtures under tests/golden/ still reproduce.
"""
import numpy as np
import pytest


def test_fp16_all_bit_patterns(oracle):
    L = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([L.orc_fp16_to_fp32(int(b)) for b in bits], dtype=np.float32)
    ok = ~np.isnan(want)  # numpy quiets signalling NaNs on conversion; payloads are not part of the contract
    np.testing.assert_array_equal(got.view(np.uint32)[ok], want.view(np.uint32)[ok])
    assert np.isnan(got[~ok]).all()
    fin = np.isfinite(want)
    back = np.array([L.orc_fp32_to_fp16(float(v)) for v in want[fin]], dtype=np.uint16)
    np.testing.assert_array_equal(back, bits[fin])
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000) * 10 ** rng.uniform(-8, 5, 20000), [65504, 65520, 1e-8, 6.1e-5]]).astype(np.float32)
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(np.array([L.orc_fp32_to_fp16(float(v)) for v in x], dtype=np.uint16), x.astype(np.float16).view(np.uint16))


def test_bf16_rne(oracle):
    import torch
    x = (np.random.default_rng(1).standard_normal(50000) * 7).astype(np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    np.testing.assert_array_equal(oracle.to_bf16_bits(x), want)


def _np_decode(O, t, raw, k):
    """Independent numpy decode from the layout table."""
    n = raw.shape[0]
    ts, blk = O.type_size(t), O.block_size(t)
    b = raw.reshape(n, k // blk, ts)
    f16 = lambda a: a.copy().view(np.float16).astype(np.float32)[..., 0]
    e = np.arange(blk)
    if t == O.Q8_0:
        return (f16(b[..., 0:2])[..., None] * b[..., 2:34].view(np.int8)).reshape(n, k)
    if t == O.Q4_0:
        qs = b[..., 2:18]
        q = np.where(e < 16, qs[..., e % 16] & 15, qs[..., e % 16] >> 4).astype(np.float32)
        return (f16(b[..., 0:2])[..., None] * (q - 8)).reshape(n, k)
    if t == O.Q4_K or t == O.Q5_K:
        d, dmin, p = f16(b[..., 0:2]), f16(b[..., 2:4]), b[..., 4:16].astype(np.int32)
        g = np.arange(8)
        sc = np.where(g < 4, p[..., g % 4] & 63, (p[..., 8 + g % 4] & 15) | ((p[..., g % 4] >> 6) << 4))
        mn = np.where(g < 4, p[..., 4 + g % 4] & 63, (p[..., 8 + g % 4] >> 4) | ((p[..., 4 + g % 4] >> 6) << 4))
        qs = b[..., (16 if t == O.Q4_K else 48):][..., :128]
        c, pp = e // 64, e % 64
        byte = qs[..., c * 32 + pp % 32]
        q = np.where(pp < 32, byte & 15, byte >> 4).astype(np.int32)
        if t == O.Q5_K:
            qh = b[..., 16:48]
            q = q | (((qh[..., pp % 32] >> (c * 2 + pp // 32)) & 1).astype(np.int32) << 4)
        return (d[..., None] * sc[..., e // 32] * q - dmin[..., None] * mn[..., e // 32]).astype(np.float32).reshape(n, k)
    if t == O.Q6_K:
        ql, qh, sc, d = b[..., :128], b[..., 128:192], b[..., 192:208].view(np.int8), f16(b[..., 208:210])
        h, pos, qt = e // 128, e % 32, (e % 128) // 32
        i = h * 64 + pos + (qt % 2) * 32
        lo = np.where(qt < 2, ql[..., i] & 15, ql[..., i] >> 4).astype(np.int32)
        hi = ((qh[..., h * 32 + pos] >> (qt * 2)) & 3).astype(np.int32)
        return (d[..., None] * sc[..., e // 16] * ((lo | (hi << 4)) - 32)).astype(np.float32).reshape(n, k)
    raise AssertionError


@pytest.mark.parametrize("name", ["Q8_0", "Q4_0", "Q4_K", "Q5_K", "Q6_K"])
def test_decode_vs_independent_numpy(oracle, name):
    t = getattr(oracle, name)
    n, k = 5, 1024
    raw = oracle.random_blocks(t, n, k, seed=3)
    np.testing.assert_allclose(oracle.dequantize(t, raw, k), _np_decode(oracle, t, raw, k), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name,band", [("Q4_0", 0.10), ("Q4_1", 0.09), ("Q5_0", 0.05), ("Q5_1", 0.045), ("Q8_0", 0.007),
                                       ("Q4_K", 0.08), ("Q5_K", 0.042), ("Q6_K", 0.021)])
def test_quantizer_error_bands(oracle, name, band):
    t = getattr(oracle, name)
    w = (np.random.default_rng(2).standard_normal((16, 2048)) * 0.02).astype(np.float32)
    d = oracle.dequantize(t, oracle.quantize(t, w), 2048)
    rel = np.sqrt(((d - w) ** 2).mean()) / 0.02
    assert rel < band, (name, rel)
    # idempotence: re-quantizing the dequantized tensor reproduces it (fixed point of the format grid)
    d2 = oracle.dequantize(t, oracle.quantize(t, d), 2048)
    assert np.sqrt(((d2 - d) ** 2).mean()) / 0.02 < band * 0.35


def test_embedding_rows_equal_dequantized_rows(oracle):
    """Reference invariant (mistralrs-quant/src/gguf/mod.rs:815-846): gathering quantized rows then
    dequantizing == dequantizing then gathering (<= 1e-6), values ((i % 37) - 18) / 7."""
    n, k = 64, 256
    vals = (((np.arange(n * k) % 37) - 18) / 7.0).astype(np.float32).reshape(n, k)
    ids = np.array([3, 0, 63, 17, 17])
    for t in (oracle.Q6_K, oracle.Q8_0):
        q = oracle.quantize(t, vals)
        full = oracle.dequantize(t, q, k)
        np.testing.assert_allclose(oracle.dequantize(t, q[ids], k), full[ids], atol=1e-6)


@pytest.mark.parametrize("name", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"])
def test_matmul_oracles_agree(oracle, name):
    t = getattr(oracle, name)
    n, k, b = 48, 1024, 3
    w = oracle.random_blocks(t, n, k, seed=9, d_scale=0.01)
    x = np.random.default_rng(4).standard_normal((b, k)).astype(np.float32)
    a = oracle.matmul_exact(t, w, n, k, x)
    dq = oracle.dequantize(t, w, k).astype(np.float64)
    np.testing.assert_allclose(a, x.astype(np.float64) @ dq.T, rtol=2e-6, atol=1e-5)
    scale = np.abs(dq).max() * np.abs(x).max() * np.sqrt(k)
    c = oracle.matmul_q8_1(t, w, n, k, oracle.quantize_q8_1(x))
    bb = oracle.matmul_cpu(t, w, n, k, x)
    assert np.abs(c - a).max() < 0.02 * scale, "Q8_1 (GPU semantics) outside the int8 activation error band"
    assert np.abs(bb - a).max() < 0.02 * scale, "candle-CPU semantics outside the int8 activation error band"


def test_q8_1_padding_and_zero_blocks(oracle):
    x = np.zeros((2, 700), dtype=np.float32)
    x[1, :5] = [1, -2, 3, -127, 0.5]
    y = oracle.quantize_q8_1(x)
    assert y.shape == (2, 1024 // 32 * 36)
    assert not y[0].any()
    blk = y[1, :36]
    assert blk[4:9].view(np.int8).tolist() == [1, -2, 3, -127, 1]  # roundf(0.5/1) = 1 (half away from zero)
    assert not y[1, 36 * 22:].any()  # zero padding beyond kx


# ---------------------------------------------------------------- round 3: the restated in-tree CPU path and the device's exact-arithmetic shortcuts
def test_round_trick_exhaustive(oracle):
    """The device rounds half away from zero (candle's .round(), the Q8_K / Q8_0 activation quantizers, fast_exp) as trunc(x + copysign(0.49999997, x)):
    identical to roundf for EVERY float with |x| <= 129 (2.2e9 values; the quantizers round products in [-128, 128], fast_exp |x log2 e| <= 126)."""
    import ctypes as C
    L = oracle.lib()
    L.orc_round_trick_mismatches.restype = C.c_int64
    assert L.orc_round_trick_mismatches(C.c_float(129.0)) == 0


def test_device_quotient_equals_ieee_division(oracle):
    """RmsNorm's x / m on the device = one correctly rounded reciprocal + q0 = x y, r = fma(-m, q0, x), q = fma(r, y, q0) (dec_core2.cuh div_by):
    equal to the IEEE quotient on 2e7 random pairs over six decades and for divisors with an all-ones mantissa."""
    import ctypes as C
    L = oracle.lib()
    L.orc_div_by_mismatches.restype = C.c_int64
    rng = np.random.default_rng(1)
    n = 20_000_000
    x = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    m = (10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    assert L.orc_div_by_mismatches(oracle._p(x), oracle._p(m), C.c_int64(n)) == 0
    m = np.full(1 << 16, np.frombuffer(np.uint32(0x3fffffff).tobytes(), np.float32)[0], np.float32)
    x = rng.standard_normal(m.size).astype(np.float32)
    assert L.orc_div_by_mismatches(oracle._p(x), oracle._p(m), C.c_int64(m.size)) == 0


def test_fast_exp_matches_elem_rs_definition(oracle):
    """orc_fast_exp = attention/backends/cpu/elem.rs:417-433 restated in numpy f32, operation for operation; ~1e-7 relative to exp on [-87, 0]."""
    xs = np.concatenate([np.linspace(-87, 0, 20001, dtype=np.float32), np.float32([-100.0, 0.25, 3.0, 87.0, 100.0])])
    f = np.float32
    want = []
    for x in xs:
        x = f(min(max(x, f(-87.0)), f(87.0)))
        zx = f(x * f(1.44269504088896340736))
        z = f(np.trunc(zx + np.copysign(f(0.5), zx))) if abs(zx - np.trunc(zx)) != 0.5 else f(np.trunc(zx) + np.copysign(f(1.0), zx))
        r = f(f(x - f(z * f(0.6933594))) - f(z * f(-2.1219444e-4)))
        r2 = f(r * r)
        p = f(r + f(r2 * f(f(0.5) + f(r * f(f(0.16666546) + f(r * f(f(0.041665795) + f(r * f(f(0.00833345) + f(r * f(0.0013920345)))))))))))
        e = np.frombuffer(np.uint32((int(z) + 127) << 23).tobytes(), np.float32)[0]
        want.append(f(e * f(f(1.0) + p)))
    got = np.float32([oracle.fast_exp(float(x)) for x in xs])
    assert np.array_equal(got, np.float32(want))
    inside = xs <= 0
    assert np.abs(got[inside] / np.exp(np.clip(xs[inside].astype(np.float64), -87, 87)) - 1).max() < 3e-7


def test_cpu_attention_restatement_vs_f64(oracle):
    """single_q.rs restated (portable elem.rs bodies) and the engine-order attention both agree with the f64 softmax attention to f32 rounding, for one
    and several kv chunks (the reference derives the chunk count from its thread pool)."""
    rng = np.random.default_rng(0)
    H, KVH, hd, S = 8, 2, 128, 700
    q = rng.standard_normal((H, hd)).astype(np.float32)
    k = rng.standard_normal((S, KVH, hd)).astype(np.float32)
    v = rng.standard_normal((S, KVH, hd)).astype(np.float32)
    sc = np.float32(1 / np.sqrt(np.float32(hd)))
    ref = oracle.attention(q[None], k, v, sc)[0]
    for got in (oracle.attention_single_q_cpu(q, k, v, sc, 1), oracle.attention_single_q_cpu(q, k, v, sc, 3), oracle.attention_engine(q, k, v, sc, 1),
                oracle.attention_engine(q, k, v, sc, 4)):
        assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_rms_norm_restatements(oracle):
    """candle's expression (x / sqrt(mean + eps) * w) in element order and in the engine's summation tree vs the f64 form: <= 4 ulp-ish."""
    rng = np.random.default_rng(2)
    for d in (512, 4096, 14336):
        x = rng.standard_normal((2, d)).astype(np.float32)
        w = (1 + 0.05 * rng.standard_normal(d)).astype(np.float32)
        ref = oracle.rms_norm(x, w, 1e-5)
        for got in (oracle.rms_norm_candle(x, w, 1e-5), oracle.rms_norm_engine(x, w, 1e-5)):
            assert np.abs(got - ref).max() <= 4e-6 * np.abs(ref).max()


@pytest.mark.parametrize("name", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
def test_three_cpu_gemv_orders_agree(oracle, name):
    """ggml's generic 8-lane order (orc_matmul_cpu), one term per superblock (orc_gemv_cpu_fast) and the engine's order (orc_gemv_engine): the same
    integers and products, three f32 summation orders -> equal to f32 rounding, ragged row lengths included."""
    t = getattr(oracle, name)
    rng = np.random.default_rng(5)
    for n, k in ((32, 4096), (8, 14336), (8, 256), (8, 2304)):
        w = oracle.quantize(t, (rng.standard_normal((n, k)) * 0.02).astype(np.float32))
        x = rng.standard_normal((1, k)).astype(np.float32)
        a, b, c = oracle.matmul_cpu(t, w, n, k, x), oracle.gemv_cpu_fast(t, w, n, k, x), oracle.gemv_engine(t, w, n, k, x)
        tol = 2e-6 * np.abs(a).max()
        assert np.abs(a - b).max() <= tol and np.abs(a - c).max() <= tol


# ---- the reference's own tests of its CPU attention (mistralrs-core/src/attention/backends/cpu/tests.rs), restated against the restatement -------------
def _naive_attention(q, k, v, scale=1.0):
    """tests.rs:26-58 naive_attention for one query token: softmax(q k^T) v per head, f64."""
    h, d = q.shape
    out = np.empty((h, d))
    for i in range(h):
        logits = (k[:, i, :].astype(np.float64) @ q[i].astype(np.float64)) * scale
        w = np.exp(logits - logits.max())
        out[i] = (w / w.sum()) @ v[:, i, :].astype(np.float64)
    return out


def test_reference_cpu_attention_tests_hold_for_the_restatement(oracle):
    """tests.rs:60-108: `test_flash_attn_cpu_single_q` (all ones, h 2, d 4, 2 keys) and `test_flash_attn_cpu_single_q_multiple_kv_chunks` (h 4, d 8, 1024
    keys, inputs (x % 17) / 17, (x % 19) / 19, (x % 23) / 23, n_kv_groups 1, softmax_scale 1) with the reference's EPS = 1e-4, for one kv chunk and for
    the chunk counts a multi-threaded run takes (single_q.rs splits the keys over threads)."""
    q, k, v = np.ones((2, 4), np.float32), np.ones((2, 2, 4), np.float32), np.ones((2, 2, 4), np.float32)
    assert np.abs(oracle.attention_single_q_cpu(q, k, v, 1.0, 1) - _naive_attention(q, k, v)).max() < 1e-4
    h, d, kv = 4, 8, 1024
    q = (np.arange(h * d) % 17 / 17.0).astype(np.float32).reshape(h, d)
    k = (np.arange(kv * h * d) % 19 / 19.0).astype(np.float32).reshape(kv, h, d)
    v = (np.arange(kv * h * d) % 23 / 23.0).astype(np.float32).reshape(kv, h, d)
    want = _naive_attention(q, k, v)
    for chunks in (1, 2, 8, 64):
        assert np.abs(oracle.attention_single_q_cpu(q, k, v, 1.0, chunks) - want).max() < 1e-4, chunks


def test_reference_softmax_row_check_holds_for_fast_exp(oracle):
    """avx.rs:807-836 (`avx_kernels_match_reference`): the row softmax built on `fast_exp` sums to the libm value within 1e-3 for rows sin(0.13 i) of
    16 / 64 / 128 / 131 elements -- the reference's own accuracy statement for the polynomial the engine uses."""
    for n in (16, 64, 128, 131):
        a = np.sin(np.arange(n, dtype=np.float32) * np.float32(0.13)).astype(np.float32)
        m = a.max()
        got = float(sum(np.float64(oracle.fast_exp(float(x))) for x in (a - m)))
        want = float(np.exp((a - m).astype(np.float64)).sum())
        assert abs(got - want) / want < 1e-3
