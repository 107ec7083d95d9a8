"""GPU parity: Q8_1 activation quantizer + decode GEMV (plain / fused GLU / fused QKV) vs the oracle.

Mirrors the reference's own hot-path tests (mistralrs-quant/src/gguf/fast_mmq.rs:1533-1704:
patterned inputs, shapes 256/512, fused == unfused) and adds an independent oracle comparison,
which the reference lacks.  Bars:
  * Q8_1 bytes: BIT-EXACT vs oracle (integer + fp16 fields);
  * GEMV: integer dots are exact, f32 combination order differs -> bounded by
    8*eps*sqrt(terms)*SUM|terms| (+1 ulp of the storage dtype for f16/bf16 outputs).
"""
import numpy as np
import pytest

from tests.util import assert_close_accum, rel_err_ref, round_through, to_np, torch_dtype

pytestmark = pytest.mark.gpu

ALL_TYPES = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"]
HOT_TYPES = ["q4_k", "q5_k", "q6_k", "q8_0"]


def _ids(O):
    return {v: k for k, v in O.TYPE_NAMES.items()}


def _weights(O, tag, n, k, seed):
    """Real quantizer output where the oracle has one, random valid blocks otherwise."""
    t = _ids(O)[tag]
    if tag in ("q2_k", "q3_k"):
        return t, O.random_blocks(t, n, k, seed=seed, d_scale=0.01)
    rng = np.random.default_rng(seed)
    return t, O.quantize(t, (rng.standard_normal((n, k)) * 0.05).astype(np.float32))


def _qt(tag, n, k, packed, dev):
    from mistralrs_amd.gguf import GgmlDType, QTensor
    dt = {d.tag: d for d in GgmlDType}[tag]
    return QTensor.from_numpy(dt, (n, k), packed, dev)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("k,rows", [(256, 3), (4096, 1), (1000, 2), (14336, 8)])
def test_quantize_q8_1_bit_exact(oracle, dev, dt, k, rows):
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    rng = np.random.default_rng(k + rows)
    x = (rng.standard_normal((rows, k)) * rng.uniform(0.1, 4.0, size=(rows, 1))).astype(np.float32)
    x[0, : min(64, k)] = 0.0  # an all-zero block: amax == 0 -> q = 0, d = 0
    x = round_through(x, dt)
    xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt))
    ws, stride = fast_mmvq.quantize_q8_1(xs, k, rows)
    torch.cuda.synchronize()
    got = ws[: rows * stride * 36].cpu().numpy().reshape(rows, stride * 36)
    want = oracle.quantize_q8_1(x, stride * 32)
    assert stride * 32 == (k + 511) // 512 * 512
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("tag", ALL_TYPES)
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_plain_all_types_all_batches(oracle, dev, tag, dt):
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    n, k = 70, 1024  # 70 rows: ragged against every rows-per-wave/workgroup split
    t, packed = _weights(oracle, tag, n, k, seed=7)
    w = _qt(tag, n, k, packed, dev)
    for b in range(1, 9):
        rng = np.random.default_rng(100 + b)
        x = round_through(rng.standard_normal((b, k)).astype(np.float32), dt)
        xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt))
        out = fast_mmvq.plain(w, xs)
        y = oracle.quantize_q8_1(x)
        want, mag = oracle.matmul_q8_1_mag(t, packed, n, k, y)
        assert out.shape == (b, n) and out.dtype == torch_dtype(dt)
        assert_close_accum(to_np(out), round_through(want, dt) if dt != "f32" else want, mag, dt, k // 16,
                           f"{tag} {dt} b={b}")


@pytest.mark.parametrize("tag", HOT_TYPES)
@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 4096), (4096, 14336), (2, 256), (257, 512), (128, 11008 // 256 * 256)])
def test_plain_model_shapes(oracle, dev, tag, n, k):
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    if n * k > 4096 * 4096 and tag != "q4_k":
        pytest.skip("largest shape only for the headline type (oracle time)")
    t = _ids(oracle)[tag]
    packed = oracle.random_blocks(t, n, k, seed=n + k, d_scale=0.02)
    w = _qt(tag, n, k, packed, dev)
    x = np.random.default_rng(5).standard_normal((1, k)).astype(np.float32)
    out = fast_mmvq.plain(w, torch.from_numpy(x).to(dev))
    want, mag = oracle.matmul_q8_1_mag(t, packed, n, k, oracle.quantize_q8_1(x))
    assert_close_accum(to_np(out), want, mag, "f32", k // 16, f"{tag} {n}x{k}")


@pytest.mark.parametrize("tag", ALL_TYPES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("b", [1, 3, 8])
def test_fused_qkv_matches_oracle_and_plain(oracle, dev, tag, dt, b):
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    k, nq, nk, nv = 512, 96, 33, 32
    ws = [_weights(oracle, tag, n, k, seed=s) for n, s in ((nq, 11), (nk, 29), (nv, 47))]
    qts = [_qt(tag, n, k, p, dev) for (t, p), n in zip(ws, (nq, nk, nv))]
    x = round_through(oracle.patterned(b * k, 3, 0.2).reshape(b, k), dt)
    xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt))
    outs = fast_mmvq.fused_qkv(qts[0], qts[1], qts[2], xs)
    y = oracle.quantize_q8_1(x)
    for o, (t, p), n, qt in zip(outs, ws, (nq, nk, nv), qts):
        want, mag = oracle.matmul_q8_1_mag(t, p, n, k, y)
        assert_close_accum(to_np(o), round_through(want, dt) if dt != "f32" else want, mag, dt, k // 16, f"qkv {tag}")
        # reference test: fused == three independent projections.  Bit-identical whenever both launches take the same arithmetic
        # path; Q4_K / Q5_K use the paired-row (64-weight) path only when every row count of the launch is even (nk = 33 here
        # makes the fused launch fall back to the 32-weight path), then the two agree to f32 summation order
        sep = fast_mmvq.plain(qt, xs)
        if tag in ("q4_k", "q5_k"):
            assert_close_accum(to_np(o), to_np(sep), mag, dt, k // 16, f"qkv vs plain {tag}")
        else:
            assert torch.equal(o, sep)


@pytest.mark.parametrize("tag", ALL_TYPES)
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_fused_glu(oracle, dev, tag, dt, act):
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    if act not in (0, 1) and tag not in HOT_TYPES:
        pytest.skip("activation sweep on the hot types only")
    k, n, b = 512, 130, 2
    (t, pg), (_, pu) = _weights(oracle, tag, n, k, seed=3), _weights(oracle, tag, n, k, seed=4)
    g, u = _qt(tag, n, k, pg, dev), _qt(tag, n, k, pu, dev)
    x = round_through(oracle.patterned(b * k, 5, 0.5).reshape(b, k), dt)
    xs = torch.from_numpy(x).to(dev).to(torch_dtype(dt))
    out = to_np(fast_mmvq.fused_glu(g, u, xs, act))
    y = oracle.quantize_q8_1(x)
    gw, gm = oracle.matmul_q8_1_mag(t, pg, n, k, y)
    uw, um = oracle.matmul_q8_1_mag(t, pu, n, k, y)
    # reference order (mmvq_gguf.cu:858-866): round both to dst_t, act in f32, round, multiply, round
    gr, ur = round_through(gw, dt), round_through(uw, dt)
    want = round_through(round_through(oracle.fused_glu(gr, np.ones_like(gr), act), dt) * ur, dt)
    # tolerance: a few storage ulps (gate/up may each sit on a rounding boundary) + f32 accumulation slack
    ulp = {"f32": 2.0 ** -23, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    slack = 8 * 2.0 ** -23 * np.sqrt(k / 16) * (gm * (np.abs(ur) + 1) + um * (np.abs(gr) + 1))
    tol = 4 * ulp * (np.abs(want) + np.abs(ur) * ulp) + slack * 1.5 + 1e-7
    err = np.abs(out - want)
    assert (err <= tol).all(), f"glu {tag} {dt} act={act}: worst {err.max():.3e} tol {tol.flat[err.argmax()]:.3e}"
    # and the reference's own criterion against the materialised path
    unfused = to_np(fast_mmvq.plain(g, xs)), to_np(fast_mmvq.plain(u, xs))
    mat = round_through(round_through(oracle.fused_glu(unfused[0], np.ones_like(unfused[0]), act), dt) * unfused[1], dt)
    assert rel_err_ref(out, mat) <= 5e-3


def test_error_behaviour(oracle, dev):
    """Same refusals as fast_mmvq.rs:299-330."""
    import torch
    from mistralrs_amd.gguf import fast_mmvq
    t, p = _weights(oracle, "q4_k", 8, 256, 1)
    w = _qt("q4_k", 8, 256, p, dev)
    with pytest.raises(ValueError, match="shape mismatch"):
        fast_mmvq.plain(w, torch.zeros(1, 512, device=dev))
    with pytest.raises(ValueError, match="batch size"):
        fast_mmvq.plain(w, torch.zeros(9, 256, device=dev))
    with pytest.raises(ValueError, match="dtype"):
        fast_mmvq.plain(w, torch.zeros(1, 256, device=dev, dtype=torch.float64))
    with pytest.raises(ValueError, match="different devices"):
        fast_mmvq.plain(w, torch.zeros(1, 256))
