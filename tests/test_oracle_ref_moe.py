"""CPU: pin the oracle's MoE expert restatements (oracle/oracle.py: moe_gemv_fused_gate_up, moe_gemv_down_aggregate, moe_dispatch,
moe_weighted_reduce_flat, moe_grouped_gemm) against the REFERENCE'S OWN kernels executed on host fibers:
  oracle/_ref/libref_moe_decode.so  <- kernels/indexed_moe/indexed_moe.cu:1157-1615 (fused decode pair)
  oracle/_ref/libref_moe_grouped.so <- kernels/moe_grouped/moe_grouped.cu:1-1102 (dispatch, weighted reduce, tiled grouped GEMM)
built by oracle/build_ref.sh from the sources where they lie (streamed into g++, never copied).  The GPU tests (tests/test_moe.py) then
hold the HIP launchers to the same oracle functions."""
import ctypes as C

import numpy as np
import pytest

from tests.test_oracle_ref import TYPES as _MMVQ_TYPES, _ref

TYPES = _MMVQ_TYPES + [9]  # the MoE kernels are also instantiated for Q8_1 weights
from tests.util import round_through


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32_tol(k, mag, want):
    return 8 * 2.0 ** -23 * np.sqrt(k / 16) * np.asarray(mag, dtype=np.float64) + 2.0 ** -23 * np.abs(want) + 1e-30


def moe_case(oracle, t, seed=0, E=5, n=20, k=512, batch=3, topk=2):
    rng = np.random.default_rng(1000 * seed + t)
    mk = lambda s: np.concatenate([oracle.random_blocks(t, n, k, seed=s + e + t, d_scale=0.02) for e in range(E)], axis=0)  # [E*n, row_bytes]
    idx = rng.integers(0, E, size=batch * topk).astype(np.uint32)
    tw = rng.uniform(0.1, 0.9, size=batch * topk).astype(np.float32)
    return rng, mk, idx, tw


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("act_type", [0, 1])
def test_fused_gate_up_oracle_matches_reference_kernel(oracle, t, act_type):
    lib = _ref("libref_moe_decode.so")
    E, n, k, batch, topk = 5, 20, 512, 3, 2   # n = 20: the last block of 8 rows is ragged (row0 + r < n guard)
    rng, mk, idx, _ = moe_case(oracle, t, seed=1 + act_type, E=E, n=n, k=k, batch=batch, topk=topk)
    gate, up = mk(50), mk(90)
    x = (rng.standard_normal((batch, k)) * rng.uniform(0.3, 3.0, (batch, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    out = np.full((batch * topk, n), np.nan, dtype=np.float32)
    assert lib.ref_moe_gemv_fused_gate_up(t, _vp(gate), _vp(up), _vp(y), _vp(idx), _vp(out), n, k, batch, topk, oracle.pad512(k), act_type) == 0
    got, extra = oracle.moe_gemv_fused_gate_up(t, gate, up, n, k, y, idx, topk, act_type, with_mag=True)
    for task in range(batch * topk):
        g, gm, u, um = extra[task]
        a = oracle.moe_act(g, act_type).astype(np.float64)
        # d(out) <= |act| d(u) + |u| |act'| d(g), |act'| <= 1.13 for both activations; + f32 rounding of act (libm tanhf / expf) and the product
        tol = np.abs(a) * _f32_tol(k, um, u) + 1.2 * np.abs(u) * _f32_tol(k, gm, g) + 2.0 ** -21 * np.abs(got[task]) + 1e-30
        assert (np.abs(out[task].astype(np.float64) - got[task]) <= tol).all(), task


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("topk", [2, 3])
def test_down_aggregate_oracle_matches_reference_kernel(oracle, t, topk):
    lib = _ref("libref_moe_decode.so")
    E, n, k, batch = 4, 21, 512, 2   # n = 21: ragged last block of 16 rows
    rng, mk, idx, tw = moe_case(oracle, t, seed=3 + topk, E=E, n=n, k=k, batch=batch, topk=topk)
    w = mk(130)
    x = (rng.standard_normal((batch * topk, k)) * rng.uniform(0.3, 3.0, (batch * topk, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    out = np.zeros((batch, n), dtype=np.float32)
    assert lib.ref_moe_gemv_down_aggregate(t, _vp(w), _vp(y), _vp(idx), _vp(tw), _vp(out), n, k, batch, topk, oracle.pad512(k)) == 0
    got, mag = oracle.moe_gemv_down_aggregate(t, w, n, k, y, idx, tw, topk, with_mag=True)
    tol = _f32_tol(k, mag, got) + topk * 2.0 ** -23 * mag
    assert (np.abs(out.astype(np.float64) - got) <= tol).all()


@pytest.mark.parametrize("E,tokens,topk", [(8, 37, 2), (4, 1, 2), (64, 300, 6), (3, 5, 1)])
def test_dispatch_oracle_matches_reference_kernels(oracle, E, tokens, topk):
    """moe_dispatch_{count,prefix_sum,scatter}_kernel run in index order == the stable counting sort of the oracle; tables are consistent."""
    lib = _ref("libref_moe_grouped.so")
    rng = np.random.default_rng(E * 131 + tokens)
    ids = np.stack([rng.choice(E, size=topk, replace=False) for _ in range(tokens)]).astype(np.int32)
    if E >= 8:
        ids[ids == 5] = 4  # an expert that receives nothing (empty segment)
    total = tokens * topk
    bounds, sorted_tok, sorted_src = np.zeros(E + 1, np.int32), np.full(total, -1, np.int32), np.full(total, -1, np.int32)
    counts, cursors = np.full(E, 77, np.int32), np.full(E, 77, np.int32)
    assert lib.ref_moe_dispatch(_vp(ids), _vp(bounds), _vp(sorted_tok), _vp(sorted_src), total, E, topk, _vp(counts), _vp(cursors)) == 0
    gb, gt, gs, gc, gcur = oracle.moe_dispatch(ids, E, topk)
    np.testing.assert_array_equal(bounds, gb)
    np.testing.assert_array_equal(sorted_tok, gt)
    np.testing.assert_array_equal(sorted_src, gs)
    np.testing.assert_array_equal(counts, gc)
    np.testing.assert_array_equal(cursors, gcur)
    flat = ids.reshape(-1)
    for e in range(E):
        assert (flat[sorted_tok[bounds[e]:bounds[e + 1]]] == e).all()


@pytest.mark.parametrize("io,in_dt,out_dt", [(0, "f32", "f32"), (1, "f32", "bf16"), (2, "f16", "f16"), (3, "bf16", "bf16")])
def test_weighted_reduce_oracle_matches_reference_kernel(oracle, io, in_dt, out_dt):
    lib = _ref("libref_moe_grouped.so")
    rng = np.random.default_rng(io)
    tokens, topk, hidden = 5, 3, 300   # hidden = 300: two 256-thread blocks, the second one ragged
    x = round_through((rng.standard_normal((tokens, topk, hidden)) * 3).astype(np.float32), in_dt)
    w = rng.uniform(0.05, 0.9, (tokens, topk)).astype(np.float32)
    enc = {"f32": lambda a: a, "f16": lambda a: a.astype(np.float16), "bf16": lambda a: oracle.to_bf16_bits(a)}
    dec = {"f32": lambda a: a, "f16": lambda a: a.astype(np.float32), "bf16": lambda a: oracle.from_bf16_bits(a)}
    xin = np.ascontiguousarray(enc[in_dt](x))
    out = np.ascontiguousarray(enc[out_dt](np.zeros((tokens, hidden), np.float32)))
    assert lib.ref_moe_weighted_reduce_flat(io, _vp(xin), _vp(w), _vp(out), tokens, hidden, topk) == 0
    got = oracle.moe_weighted_reduce_flat(x, w, out_dt)
    np.testing.assert_array_equal(dec[out_dt](out), got)   # same f32 operation order, no contraction: bit-exact


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("mode", ["gate_up", "down_weighted", "down_plain"])
def test_grouped_gemm_oracle_matches_reference_kernel(oracle, t, mode):
    """moe_grouped_gemm_<t> (tiled kernel on host fibers; Q8_0 = the dp4a shared-memory variant) == per sorted route the Q8_1 matvec oracle.
    gate_up: input_dim1 = 1 (token rows), out[ti]; down_weighted: input_dim1 = 2 (route rows), atomicAdd(out[token], acc * w[flat]);
    down_plain: input_dim1 = 0 (rows already sorted), out[ti]."""
    lib = _ref("libref_moe_grouped.so")
    E, n, k, tokens, topk = 4, 70, 512, 90, 2   # n = 70: two 64-row tiles, the second ragged; 180 routes, most of them on expert 2
    rng = np.random.default_rng(10 * t + len(mode))
    ids = np.stack([rng.choice(E, size=topk, replace=False) for _ in range(tokens)]).astype(np.int32)
    ids[:, 0] = np.where(rng.random(tokens) < 0.9, 2, ids[:, 0])   # skew: expert 2 receives more than one 64-route tile
    ids[:, 1] = np.where(ids[:, 1] == ids[:, 0], (ids[:, 0] + 1) % E, ids[:, 1])
    bounds, sorted_tok, _, _, _ = oracle.moe_dispatch(ids, E, topk)
    assert (np.diff(bounds) > 64).any()
    total = tokens * topk
    w = np.concatenate([oracle.random_blocks(t, n, k, seed=300 + e + t, d_scale=0.02) for e in range(E)], axis=0)
    input_dim1 = {"gate_up": 1, "down_weighted": 2, "down_plain": 0}[mode]
    rows_in = tokens if input_dim1 == 1 else total
    x = (rng.standard_normal((rows_in, k)) * rng.uniform(0.3, 3.0, (rows_in, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    tw = rng.uniform(0.1, 0.9, total).astype(np.float32) if mode == "down_weighted" else None
    shape = (tokens, n) if tw is not None else (total, n)
    out = np.zeros(shape, dtype=np.float32)
    assert lib.ref_moe_grouped_gemm(t, _vp(w), _vp(y), _vp(bounds), _vp(sorted_tok), _vp(tw) if tw is not None else None, _vp(out), n, k,
                                    oracle.pad512(k), E, topk, input_dim1) == 0
    got, mag = oracle.moe_grouped_gemm(t, w, n, k, y, bounds, sorted_tok, tw, topk, input_dim1, np.zeros(shape, dtype=np.float32))
    tol = _f32_tol(k, mag, got) + topk * 2.0 ** -23 * mag
    assert (np.abs(out.astype(np.float64) - got) <= tol).all()


def test_imoe_quantize_q8_1_is_the_mmvq_quantizer(oracle):
    """quantize_q8_1 of kernels/indexed_moe/indexed_moe.cu:673-708 (behind launch_quantize_q8_1*) on host fibers == the oracle's Q8_1 quantizer
    (already pinned to mmvq_gguf_quantize_q8_1_f32), byte for byte -- so the HIP quantizer serves both symbol families."""
    lib = _ref("libref_imoe.so")
    rng = np.random.default_rng(8)
    rows, k = 3, 700
    x = (rng.standard_normal((rows, k)) * rng.uniform(0.01, 30.0, (rows, 1))).astype(np.float32)
    x[1, 64:96] = 0.0   # an all-zero block
    kp = oracle.pad512(k)
    y = np.zeros((rows, kp // 32 * 36), dtype=np.uint8)
    assert lib.ref_imoe_quantize_q8_1(_vp(x), _vp(y), k, kp, rows) == 0
    np.testing.assert_array_equal(y, oracle.quantize_q8_1(x))
