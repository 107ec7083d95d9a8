"""Parity of the drop-in MoE expert launchers of libmistralrsquant.so (include/mistralrs_quant.h) against the oracle restatements that
tests/test_oracle_ref_moe.py pins to the reference's own kernels:
    launch_moe_gemv_fused_gate_up_<t>_q8_1 / launch_moe_gemv_down_aggregate_<t>_q8_1   (kernels/indexed_moe/indexed_moe.cu:1336-1726)
    launch_moe_dispatch / launch_moe_grouped_gemm_<t> / launch_moe_weighted_reduce_flat*  (kernels/moe_grouped/moe_grouped.cu:630-1235)
Integer tables (dispatch) bit-exact; f32 results within the f32-accumulation bound of the Q8_1 matvec oracle (the HIP core sums the same
int8 block dots in a different lane order); the weighted reduce bit-exact (same f32 operation order, contraction off).

The bodies (`check_*`) take a backend (tests/abi_backends.py): the `-m gpu` tests below run them on the MI355X through the product
library; tests/test_hip_host_emulation.py runs the same bodies on the host emulation of the same kernel sources.
These launchers were written after the round's GPU minutes were spent (host emulation green, first device run = the driver's round-end
`pytest -m gpu`); the file sorts last so that a device-only surprise cannot mask the verified suites under `-x`."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend
from tests.util import round_through

TNAMES = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q8_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
P, I = C.c_void_p, C.c_int


def _f32_tol(k, mag, want):
    return 8 * 2.0 ** -23 * np.sqrt(k / 16) * np.asarray(mag, dtype=np.float64) + 2.0 ** -23 * np.abs(want) + 1e-30


def _tag(oracle, t):
    return oracle.MOE_TYPE_NAMES[t].replace("_k", "k")


def _stack(oracle, t, E, n, k, seed):
    return np.concatenate([oracle.random_blocks(t, n, k, seed=seed + e + t, d_scale=0.02) for e in range(E)], axis=0)  # [E*n, row_bytes]


def check_fused_gate_up(oracle, be, tname, act_type, n=22, k=1024):
    t = getattr(oracle, tname)
    E, batch, topk = 5, 3, 2
    rng = np.random.default_rng(7 + t + act_type)
    gate, up = _stack(oracle, t, E, n, k, 50), _stack(oracle, t, E, n, k, 90)
    idx = rng.integers(0, E, size=batch * topk).astype(np.uint32)
    x = (rng.standard_normal((batch, k)) * rng.uniform(0.3, 3.0, (batch, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    gt, ut, yt, it = be.buf(gate), be.buf(up), be.buf(y), be.buf(idx)
    out = be.buf(np.full((batch * topk, n), np.nan, dtype=np.float32))
    fn = be.sym(f"launch_moe_gemv_fused_gate_up_{_tag(oracle, t)}_q8_1", [P] * 5 + [I] * 6 + [P])
    fn(gt.ptr, ut.ptr, yt.ptr, it.ptr, out.ptr, n, k, batch, topk, oracle.pad512(k), act_type, be.stream)
    got = out.numpy().astype(np.float64)
    want, extra = oracle.moe_gemv_fused_gate_up(t, gate, up, n, k, y, idx, topk, act_type, with_mag=True)
    for task in range(batch * topk):
        g, gm, u, um = extra[task]
        a = np.abs(oracle.moe_act(g, act_type)).astype(np.float64)
        tol = a * _f32_tol(k, um, u) + 1.2 * np.abs(u) * _f32_tol(k, gm, g) + 2.0 ** -20 * np.abs(want[task]) + 1e-30  # device tanhf / expf: few ulp
        assert (np.abs(got[task] - want[task]) <= tol).all(), task


def check_down_aggregate(oracle, be, tname, topk, k=1024):
    t = getattr(oracle, tname)
    E, n, batch = 4, (22 if topk == 2 else 21), 2   # even n: paired-row path for Q4_K / Q5_K; odd n: single rows
    rng = np.random.default_rng(11 + t + topk)
    w = _stack(oracle, t, E, n, k, 130)
    idx = rng.integers(0, E, size=batch * topk).astype(np.uint32)
    tw = rng.uniform(0.1, 0.9, size=batch * topk).astype(np.float32)
    x = (rng.standard_normal((batch * topk, k)) * rng.uniform(0.3, 3.0, (batch * topk, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    wt, yt, twt, it = be.buf(w), be.buf(y), be.buf(tw), be.buf(idx)
    out = be.buf(np.zeros((batch, n), dtype=np.float32))   # the caller zero-fills (gguf/cuda.rs fused decode path)
    fn = be.sym(f"launch_moe_gemv_down_aggregate_{_tag(oracle, t)}_q8_1", [P] * 5 + [I] * 5 + [P])
    fn(wt.ptr, yt.ptr, it.ptr, twt.ptr, out.ptr, n, k, batch, topk, oracle.pad512(k), be.stream)
    got = out.numpy().astype(np.float64)
    want, mag = oracle.moe_gemv_down_aggregate(t, w, n, k, y, idx, tw, topk, with_mag=True)
    tol = _f32_tol(k, mag, want) + topk * 2.0 ** -23 * mag   # + the order of the f32 atomics
    assert (np.abs(got - want) <= tol).all()


def check_dispatch(oracle, be, E, tokens, topk):
    """Bit-exact tables: bounds, stable sorted routes, source tokens, counts, final cursors; twice in a row (scratch is reusable)."""
    rng = np.random.default_rng(E * 131 + tokens)
    ids = np.argsort(rng.random((tokens, E)), axis=1)[:, :topk].astype(np.int32)
    if E >= 8:
        ids[ids == 5] = 4  # an expert that receives nothing
    total = tokens * topk
    it = be.buf(ids)
    i32 = lambda n: be.buf(np.full(n, -7, dtype=np.int32))
    bounds, stok, ssrc, counts, cursors = i32(E + 1), i32(total), i32(total), i32(E), i32(E)
    fn = be.sym("launch_moe_dispatch", [P] * 4 + [I] * 3 + [P] * 3)
    gb, gt, gs, gc, gcur = oracle.moe_dispatch(ids, E, topk)
    for _ in range(2):
        fn(it.ptr, bounds.ptr, stok.ptr, ssrc.ptr, total, E, topk, counts.ptr, cursors.ptr, be.stream)
        np.testing.assert_array_equal(bounds.numpy(), gb)
        np.testing.assert_array_equal(stok.numpy(), gt)
        np.testing.assert_array_equal(ssrc.numpy(), gs)
        np.testing.assert_array_equal(counts.numpy(), gc)
        np.testing.assert_array_equal(cursors.numpy(), gcur)
    stok.fill(-7)   # sorted_source_ids is optional (NULL)
    fn(it.ptr, bounds.ptr, stok.ptr, None, total, E, topk, counts.ptr, cursors.ptr, be.stream)
    np.testing.assert_array_equal(stok.numpy(), gt)


REDUCE_CASES = [("launch_moe_weighted_reduce_flat", "f32", "f32"), ("launch_moe_weighted_reduce_flat_bf16", "f32", "bf16"),
                ("launch_moe_weighted_reduce_flat_f16_input", "f16", "f16"), ("launch_moe_weighted_reduce_flat_bf16_input", "bf16", "bf16")]


def check_weighted_reduce(oracle, be, sym, in_dt, out_dt):
    rng = np.random.default_rng(len(sym))
    tokens, topk, hidden = 7, 3, 1000
    x = round_through((rng.standard_normal((tokens, topk, hidden)) * 3).astype(np.float32), in_dt)
    w = rng.uniform(0.05, 0.9, (tokens, topk)).astype(np.float32)
    xt, wt = be.buf(x, None if in_dt == "f32" else in_dt), be.buf(w)
    out = be.buf(np.zeros((tokens, hidden), dtype=np.float32), None if out_dt == "f32" else out_dt)
    fn = be.sym(sym, [P] * 3 + [I] * 3 + [P], restype=I)
    assert fn(xt.ptr, wt.ptr, out.ptr, tokens, hidden, topk, be.stream) == 0
    np.testing.assert_array_equal(out.numpy().astype(np.float32), oracle.moe_weighted_reduce_flat(x, w, out_dt))


def check_grouped_gemm(oracle, be, tname, mode, k=1024, tokens=90):
    """Dispatch tables feed launch_moe_grouped_gemm_<t>; `tokens` x top-2 over 4 experts with most routes on expert 2 (segments of 1 .. >64
    routes: several 8-route passes with a short last pass), n = 70 rows."""
    t = getattr(oracle, tname)
    E, n, topk = 4, 70, 2
    rng = np.random.default_rng(10 * t + len(mode))
    ids = np.stack([rng.choice(E, size=topk, replace=False) for _ in range(tokens)]).astype(np.int32)
    ids[:, 0] = np.where(rng.random(tokens) < 0.9, 2, ids[:, 0])
    ids[:, 1] = np.where(ids[:, 1] == ids[:, 0], (ids[:, 0] + 1) % E, ids[:, 1])
    total = tokens * topk
    bounds, sorted_tok, _, _, _ = oracle.moe_dispatch(ids, E, topk)
    w = _stack(oracle, t, E, n, k, 300)
    input_dim1 = {"gate_up": 1, "down_weighted": 2, "down_plain": 0}[mode]
    rows_in = tokens if input_dim1 == 1 else total
    x = (rng.standard_normal((rows_in, k)) * rng.uniform(0.3, 3.0, (rows_in, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    tw = rng.uniform(0.1, 0.9, total).astype(np.float32) if mode == "down_weighted" else None
    shape = (tokens, n) if tw is not None else (total, n)
    wt, yt, bt, st = be.buf(w), be.buf(y), be.buf(bounds), be.buf(sorted_tok)
    twt = be.buf(tw) if tw is not None else None
    out = be.buf(np.zeros(shape, dtype=np.float32))
    fn = be.sym(f"launch_moe_grouped_gemm_{_tag(oracle, t)}", [P] * 6 + [I] * 6 + [P])
    fn(wt.ptr, yt.ptr, bt.ptr, st.ptr, twt.ptr if twt is not None else None, out.ptr, n, k, oracle.pad512(k), E, topk, input_dim1, be.stream)
    got = out.numpy().astype(np.float64)
    want, mag = oracle.moe_grouped_gemm(t, w, n, k, y, bounds, sorted_tok, tw, topk, input_dim1, np.zeros(shape, dtype=np.float32))
    tol = _f32_tol(k, mag, want) + topk * 2.0 ** -23 * mag
    assert (np.abs(got - want) <= tol).all()


def check_quantize_q8_1(oracle, be, sym, dt):
    """launch_quantize_q8_1{,_f16,_bf16} (indexed_moe.cu:673-808,1016-1023): bit-exact Q8_1 blocks, rows zero-padded to kx_padded."""
    rng = np.random.default_rng(len(sym))
    rows, k = 3, 700
    x = round_through((rng.standard_normal((rows, k)) * rng.uniform(0.01, 30.0, (rows, 1))).astype(np.float32), dt)
    x[1, 64:96] = 0.0
    kp = oracle.pad512(k)
    xb, y = be.buf(x, None if dt == "f32" else dt), be.buf(np.full((rows, kp // 32 * 36), 0xAA, dtype=np.uint8))
    if dt == "f32":
        be.sym(sym, [P] * 2 + [I] * 4 + [P])(xb.ptr, y.ptr, k, kp, (kp + 255) // 256, rows, be.stream)
    else:
        be.sym(sym, [P] * 2 + [I] * 3 + [P])(xb.ptr, y.ptr, k, kp, rows, be.stream)
    np.testing.assert_array_equal(y.numpy(), oracle.quantize_q8_1(x))


# ------------------------------------------------------------------------------------------------------------------- on the MI355X
@pytest.fixture
def be(dev):
    return GpuBackend(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", TNAMES)
@pytest.mark.parametrize("act_type", [0, 1])
def test_moe_gemv_fused_gate_up_abi(oracle, be, tname, act_type):
    check_fused_gate_up(oracle, be, tname, act_type)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", TNAMES)
@pytest.mark.parametrize("topk", [2, 3])
def test_moe_gemv_down_aggregate_abi(oracle, be, tname, topk):
    check_down_aggregate(oracle, be, tname, topk)


@pytest.mark.gpu
@pytest.mark.parametrize("E,tokens,topk", [(8, 37, 2), (4, 1, 2), (64, 300, 6), (3, 5, 1), (128, 4096, 8)])
def test_moe_dispatch_abi(oracle, be, E, tokens, topk):
    check_dispatch(oracle, be, E, tokens, topk)


@pytest.mark.gpu
@pytest.mark.parametrize("sym,in_dt,out_dt", REDUCE_CASES)
def test_moe_weighted_reduce_flat_abi(oracle, be, sym, in_dt, out_dt):
    check_weighted_reduce(oracle, be, sym, in_dt, out_dt)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", TNAMES)
@pytest.mark.parametrize("mode", ["gate_up", "down_weighted", "down_plain"])
def test_moe_grouped_gemm_abi(oracle, be, tname, mode):
    check_grouped_gemm(oracle, be, tname, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("sym,dt", [("launch_quantize_q8_1", "f32"), ("launch_quantize_q8_1_f16", "f16"), ("launch_quantize_q8_1_bf16", "bf16")])
def test_moe_quantize_q8_1_abi(oracle, be, sym, dt):
    check_quantize_q8_1(oracle, be, sym, dt)
