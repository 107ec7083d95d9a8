"""GPU parity: RoPE, fused GLU, RMSNorm family vs the oracle.
Bars: f32 -> <= 2 ulp-ish (rsqrt / exp approximations: 1e-6 relative); f16/bf16 -> the reference
rounds after every arithmetic step, reproduced in the expected value; <= 1 storage ulp allowed."""
import numpy as np
import pytest

from tests.util import round_through, to_np, torch_dtype

pytestmark = pytest.mark.gpu
ULP = {"f32": 2.0 ** -23, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("with_pos", [False, True])
def test_rotary(oracle, dev, dt, neox, with_pos):
    import torch
    from mistralrs_amd import ops
    rng = np.random.default_rng(3)
    T, H, KVH, hd, maxpos = 7, 8, 2, 128, 64
    inv = 1.0 / (500000.0 ** (np.arange(0, hd, 2) / hd))
    ang = np.arange(maxpos)[:, None] * inv[None, :]
    cos, sin = round_through(np.cos(ang).astype(np.float32), dt), round_through(np.sin(ang).astype(np.float32), dt)
    q = round_through(rng.standard_normal((T, H, hd)).astype(np.float32), dt)
    k = round_through(rng.standard_normal((T, KVH, hd)).astype(np.float32), dt)
    pos = rng.integers(0, maxpos, T).astype(np.int32) if with_pos else np.arange(T, dtype=np.int32)
    td = torch_dtype(dt)
    qt, kt = torch.from_numpy(q).to(dev).to(td), torch.from_numpy(k).to(dev).to(td)
    ops.apply_rotary_qk(qt, kt, torch.from_numpy(cos).to(dev).to(td), torch.from_numpy(sin).to(dev).to(td), neox,
                        torch.from_numpy(pos).to(dev) if with_pos else None)
    for got, x in ((to_np(qt), q), (to_np(kt), k)):
        want = oracle.rope(x, cos, sin, pos, neox)
        tol = 3 * ULP[dt] * (np.abs(x).max() * 2)  # each product and the sum round once in dt
        assert np.abs(got - want).max() <= tol + 1e-6
        if dt != "f32":  # exact: the per-operation rounding of the reference's 16-bit instantiation (pinned in tests/test_oracle_ref.py)
            from tests.test_oracle_ref import _rope16
            np.testing.assert_array_equal(got, _rope16(x, cos, sin, pos, neox, dt))


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_fused_glu_op(oracle, dev, dt, act):
    import torch
    from mistralrs_amd import ops
    rng = np.random.default_rng(act)
    wide = round_through(rng.standard_normal((5, 2 * 1001)).astype(np.float32) * 3, dt)
    wt = torch.from_numpy(wide).to(dev).to(torch_dtype(dt))
    a, b = wt[:, :1001], wt[:, 1001:]  # strided rows, odd width -> scalar path
    got = to_np(ops.fused_glu(a, b, act))
    want = round_through(round_through(oracle.fused_glu(wide[:, :1001], np.ones((5, 1001), np.float32), act), dt) * wide[:, 1001:], dt)
    tol = 2 * ULP[dt] * np.abs(want) + 2e-6 * np.abs(wide[:, 1001:]) * (np.abs(wide[:, :1001]) + 1)
    assert (np.abs(got - want) <= tol + 1e-30).all()


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("rows,cols", [(3, 4096), (2, 1000), (1, 14336), (2, 20000)])
def test_rms_norm_family(oracle, dev, dt, rows, cols):
    import torch
    from mistralrs_amd import ops
    rng = np.random.default_rng(cols)
    td = torch_dtype(dt)
    x = round_through(rng.standard_normal((rows, cols)).astype(np.float32), dt)
    r = round_through(rng.standard_normal((rows, cols)).astype(np.float32), dt)
    w = round_through(1 + 0.1 * rng.standard_normal(cols).astype(np.float32), dt)
    xt, rt, wt = (torch.from_numpy(a).to(dev).to(td) for a in (x, r, w))
    eps = 1e-5
    tol = lambda want: 1.01 * ULP[dt] * np.abs(want) + 4e-6 * np.abs(want) + 1e-7
    want = oracle.rms_norm(x, w, eps)
    got = to_np(ops.rms_norm(xt, wt, eps))
    assert (np.abs(got - (round_through(want, dt) if dt != "f32" else want)) <= tol(want)).all()
    # add_rms_norm: residual_out = T(x + r) exactly; norm over the ROUNDED sum (sort.cu:403-428)
    res, nrm = ops.add_rms_norm(xt, rt, wt, eps)
    s = round_through(x + r, dt)
    np.testing.assert_array_equal(to_np(res), s)
    want = oracle.rms_norm(s, w, eps)
    assert (np.abs(to_np(nrm) - (round_through(want, dt) if dt != "f32" else want)) <= tol(want)).all()
    # rms_norm_residual: (r + rms(x)*w) * scale
    sc = round_through(np.array([0.5], np.float32), dt)
    got = to_np(ops.rms_norm_residual(xt, rt, wt, eps, torch.from_numpy(sc).to(dev).to(td)))
    want = (r + oracle.rms_norm(x, w, eps)) * sc[0]
    assert (np.abs(got - (round_through(want, dt) if dt != "f32" else want)) <= tol(want) + 4e-6 * np.abs(r)).all()
