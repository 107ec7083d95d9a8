"""GPU parity: Mixtral-style sparse MoE decode block (router top-2 + indexed expert GEMVs) vs a numpy/oracle restatement of
SparseMoeBlock::forward (models/mixtral.rs:280-304; router ops.rs:259-336; GEMV arithmetic = oracle C, Q8_1 activations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sparse_moe_block_decode(oracle, dev):
    import torch
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.moe import SparseMoeBlock, StackedExperts
    rng = np.random.default_rng(5)
    E, K, ff, top_k, tokens = 8, 512, 1024, 2, 3
    tg, td = oracle.Q4_K, oracle.Q6_K
    gate = np.stack([oracle.random_blocks(tg, ff, K, seed=10 + e, d_scale=0.02) for e in range(E)])
    up = np.stack([oracle.random_blocks(tg, ff, K, seed=30 + e, d_scale=0.02) for e in range(E)])
    down = np.stack([oracle.random_blocks(td, K, ff, seed=50 + e, d_scale=0.02) for e in range(E)])
    gate_w = (rng.standard_normal((E, K)) * 0.3).astype(np.float32)
    norm_w = (1 + 0.01 * rng.standard_normal(K)).astype(np.float32)
    h0 = rng.standard_normal((tokens, K)).astype(np.float32)
    blk = SparseMoeBlock(torch.from_numpy(gate_w).to(dev),
                         StackedExperts(GgmlDType.Q4K, E, ff, K, torch.from_numpy(gate.reshape(-1)).to(dev)),
                         StackedExperts(GgmlDType.Q4K, E, ff, K, torch.from_numpy(up.reshape(-1)).to(dev)),
                         StackedExperts(GgmlDType.Q6K, E, K, ff, torch.from_numpy(down.reshape(-1)).to(dev)), top_k=top_k)
    h, ids, wts = blk.forward(torch.from_numpy(h0).to(dev), torch.from_numpy(norm_w).to(dev))
    h, ids, wts = h.cpu().numpy(), ids.cpu().numpy(), wts.cpu().numpy()
    # ---- reference
    xn = oracle.rms_norm(h0, norm_w, 1e-5)
    logits = xn.astype(np.float64) @ gate_w.astype(np.float64).T
    p = np.exp(logits - logits.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    want_ids = np.argsort(-p, axis=1, kind="stable")[:, :top_k]
    np.testing.assert_array_equal(ids, want_ids)
    ww = np.take_along_axis(p, want_ids, 1); ww /= ww.sum(1, keepdims=True)
    np.testing.assert_allclose(wts, ww, rtol=2e-5)
    y = oracle.quantize_q8_1(xn)
    for t in range(tokens):
        out = h0[t].astype(np.float64).copy()
        mag = np.abs(h0[t]).astype(np.float64)
        for s in range(top_k):
            e = want_ids[t, s]
            g = oracle.matmul_q8_1(tg, gate[e], ff, K, y[t:t + 1])[0]
            u = oracle.matmul_q8_1(tg, up[e], ff, K, y[t:t + 1])[0]
            act = oracle.fused_glu(g[None], u[None], 0)
            dn, dm = oracle.matmul_q8_1_mag(td, down[e], K, ff, oracle.quantize_q8_1(act))
            out += ww[t, s] * dn[0]
            mag += ww[t, s] * dm[0]
        err = np.abs(h[t] - out)
        # f32 accumulation + a possible int8 rounding flip of an activation near a tie (1/127 of its block range)
        assert (err <= 2e-3 * np.abs(out).max() + 1e-5 * mag).all(), float(err.max())


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_moe_router_topk_abi_vs_oracle(oracle, dev, dt):
    """Drop-in `moe_router_topk_{f32,f16,bf16}` (mistralrs-core cuda/ffi.rs:523-579) against the oracle restatement that is pinned to the
    reference kernel: ids identical (ties -> lowest id, NaN never selected), weights to f32 rounding (device expf vs libm)."""
    import torch
    from mistralrs_amd import ops
    from tests.test_oracle_ref import ROUTER_CASES, _router_inputs
    from tests.util import round_through, torch_dtype
    for case in ROUTER_CASES:
        x, bias, esc = _router_inputs(case, rows=11, seed=3)
        x = round_through(np.nan_to_num(x, nan=0.0), dt)
        if case["E"] >= 8:
            x[1, 3] = np.nan
        t = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
        ids, w = ops.moe_router_topk(t(x).to(torch_dtype(dt)), case["k"], case["score"], case["weight"], case["renorm"], t(bias), t(esc), case.get("clamp"),
                                     case.get("norm_min", 0.0), case.get("oscale", 1.0))
        gi, gw = oracle.moe_router_topk(x, case["k"], case["score"], case["weight"], case["renorm"], bias, esc, case.get("clamp"), case.get("norm_min", 0.0),
                                        case.get("oscale", 1.0))
        got_i, got_w = ids.cpu().numpy().astype(np.uint32), w.cpu().numpy()
        # a device exp that differs in the last bit can swap two nearly tied picks: compare ids only where the oracle's selections are separated
        same = got_i == gi
        assert same.mean() >= 0.98, (case, float(same.mean()))
        np.testing.assert_allclose(got_w[same], gw[same], rtol=2e-5, atol=1e-8)
    with pytest.raises(ValueError, match="expert count"):
        ops.moe_router_topk(torch.zeros(2, 12, device=dev), 2)


@pytest.mark.parametrize("tname", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q8_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"])
@pytest.mark.parametrize("input_dim1", [1, 2])
def test_indexed_moe_forward_abi(oracle, dev, tname, input_dim1):
    """Drop-in `launch_indexed_moe_forward_<t>_q8_1` (gguf/ffi.rs:100-260): per task the plain MMVQ result of the selected expert, within the
    f32-accumulation bound of the oracle (itself pinned to the reference kernel, tests/test_oracle_ref.py)."""
    import ctypes as C
    import torch
    from mistralrs_amd import _lib
    from tests.test_oracle_ref import _imoe_case
    t = getattr(oracle, tname)
    E, n, k, batch, topk, w, idx, rng = _imoe_case(oracle, t, seed=4)
    n = 12
    rows_in = batch if input_dim1 == 1 else batch * topk
    x = (rng.standard_normal((rows_in, k)) * rng.uniform(0.3, 4.0, (rows_in, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    kp = oracle.pad512(k)
    wt, yt, it = torch.from_numpy(w).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(idx.astype(np.int32)).to(dev)
    out = torch.zeros(batch * topk, n, device=dev)
    tag = oracle.MOE_TYPE_NAMES[t].replace("_k", "k")
    fn = _lib.sym("quant", f"launch_indexed_moe_forward_{tag}_q8_1", [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p])
    fn(wt.data_ptr(), yt.data_ptr(), it.data_ptr(), out.data_ptr(), n, k, batch, topk, kp, input_dim1, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().numpy().astype(np.float64)
    for task in range(batch * topk):
        e = int(idx[task])
        row = task // topk if input_dim1 == 1 else task
        want, mag = oracle.matmul_q8_1_mag(t, w[e * n:(e + 1) * n], n, k, y[row:row + 1])
        tol = 8 * 2.0 ** -23 * np.sqrt(k / 16) * mag[0].astype(np.float64) + 2.0 ** -23 * np.abs(want[0]) + 1e-30
        assert (np.abs(got[task] - want[0]) <= tol).all(), task

