"""Importance-matrix collection and importance-weighted ISQ.

Reference: `mistralrs-quant/src/imatrix.rs` (statistics, `.cimatrix`; its unit tests :234-452 are restated below), `gguf/mod.rs:200-262`
(`quantize_expert_stack`), `:633-700` (`apply_isq` with `imatrix_weight`), `:710-739` (`begin/end_track_stats`).  The weighted quantizers are candle's
`QTensor::quantize_imatrix` = GGML's `quantize_row_q{2,3,4,5,6}_K_impl` with quant_weights; candle is a git dependency outside the tree, so the oracle
restates the public GGML algorithm ("parity unpinned") and the device blocks must equal it bit for bit.  C-ABI bodies run on the wave64 host emulation
(CPU) and on the MI355X; the Python surface (`ImatrixLayerStats`, `GgufMatMul`, `isq`) needs device tensors: `-m gpu`."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.util import round_through

SRC_CODE = {"f32": 0, "f16": 1, "bf16": 30}


def _weights(src, n=9, k=1024, seed=0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((n, k)) * rng.uniform(1e-3, 2.0, (n, 1))).astype(np.float32)
    w[1, :256] = 0.0                           # an all-zero superblock
    w[2, 256:288] = 0.37                       # a constant sub-block
    w[3, :] *= 1e-9
    w[4, 512:768] = np.abs(w[4, 512:768])      # all-positive superblock
    w[5, 100] = 60000.0                        # an outlier
    return round_through(w, src) if src != "f32" else w


def _imatrix(k, seed):
    rng = np.random.default_rng(seed)
    im = (rng.standard_normal(k) ** 2 * rng.uniform(0.01, 50.0)).astype(np.float32)
    im[7:40] = 0.0                             # columns no calibration token ever excited
    im[300] = 1e6
    return im


def check_isq_imatrix(oracle, be, tname, src):
    t = getattr(oracle, tname)
    w = _weights(src, seed=t)
    n, k = w.shape
    im = _imatrix(k, t)
    want = oracle.quantize_imatrix(t, w, im)
    wb, imb = be.buf(w, None if src == "f32" else src), be.buf(im)
    out = be.buf(np.full(want.shape, 0xAA, dtype=np.uint8))
    fn = be.sym("mrs_isq_quantize_imatrix", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    assert fn(wb.ptr, SRC_CODE[src], out.ptr, n, k, t, imb.ptr, be.stream) == 0
    np.testing.assert_array_equal(out.numpy(), want)
    assert not np.array_equal(want, oracle.quantize(t, w))                      # the importance vector changes the blocks
    assert fn(wb.ptr, SRC_CODE[src], out.ptr, n, 1000, t, imb.ptr, be.stream) == -1   # k % 256
    assert fn(wb.ptr, SRC_CODE[src], out.ptr, n, k, oracle.Q8_0, imb.ptr, be.stream) == -1   # no weighted quantizer for this target
    # the weighted error goes down (what the importance vector is for)
    d_w, d_p = oracle.dequantize(t, want, k), oracle.dequantize(t, oracle.quantize(t, w), k)
    ew = lambda d: float((im.astype(np.float64) * (d.astype(np.float64) - w) ** 2)[6:].sum())   # rows without the planted degeneracies
    assert ew(d_w) <= ew(d_p)


@pytest.mark.parametrize("tname", ["Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"])
@pytest.mark.parametrize("src", ["f32", "bf16", "f16"])
def test_isq_imatrix_host_emulation(oracle, tname, src):
    check_isq_imatrix(oracle, HostBackend(), tname, src)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"])
@pytest.mark.parametrize("src", ["f32", "bf16"])
def test_isq_imatrix_gpu(oracle, dev, tname, src):
    check_isq_imatrix(oracle, GpuBackend(dev), tname, src)


def check_accumulate(be):
    rng = np.random.default_rng(3)
    x1, x2 = rng.standard_normal((4, 300)).astype(np.float32), rng.standard_normal((3, 300)).astype(np.float32)
    acc = be.buf(np.zeros(300, np.float32))
    fn = be.sym("mrs_imatrix_accumulate", [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p], C.c_int)
    for x in (x1, x2):
        assert fn(be.buf(x).ptr, 0, x.shape[0], 300, acc.ptr, be.stream) == 0
    want = np.zeros(300, np.float32)
    for x in (x1, x2):                       # `row_accum + inp.sqr().sum(0)`: rows in order, then one add
        s = np.zeros(300, np.float32)
        for r in x:
            s = s + r * r
        want = want + s
    np.testing.assert_array_equal(acc.numpy(), want)
    # routed, rank 2 (a token's row goes to all its experts) and rank 3 (per slot): imatrix.rs:292-327
    fr = be.sym("mrs_imatrix_accumulate_routed", [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int)
    x = np.array([[1.0, 2.0], [3.0, 4.0]], np.float32)
    ids = np.array([[0, 2], [2, 1]], np.uint32)
    accum, counts = be.buf(np.zeros((3, 2), np.float32)), be.buf(np.zeros(3, np.float32))
    assert fr(be.buf(x).ptr, 0, be.buf(ids).ptr, 2, 2, 2, 0, 3, accum.ptr, counts.ptr, be.stream) == 0
    np.testing.assert_array_equal(accum.numpy(), np.array([[1, 4], [9, 16], [10, 20]], np.float32))
    np.testing.assert_array_equal(counts.numpy(), np.array([1, 1, 2], np.float32))
    x3 = np.array([[[1.0, 0.0], [0.0, 2.0]]], np.float32)
    accum, counts = be.buf(np.zeros((2, 2), np.float32)), be.buf(np.zeros(2, np.float32))
    assert fr(be.buf(x3).ptr, 0, be.buf(np.array([[0, 1]], np.uint32)).ptr, 1, 2, 2, 1, 2, accum.ptr, counts.ptr, be.stream) == 0
    np.testing.assert_array_equal(accum.numpy(), np.array([[1, 0], [0, 4]], np.float32))


def test_imatrix_accumulate_host_emulation(oracle):
    check_accumulate(HostBackend())


@pytest.mark.gpu
def test_imatrix_accumulate_gpu(oracle, dev):
    check_accumulate(GpuBackend(dev))


def test_cimatrix_file_round_trip(tmp_path):
    """imatrix.rs:175-232: u64 count, then (u64 key length, key, u64 n, n x f32) little endian; only `.cimatrix` names are accepted for writing."""
    import struct
    from mistralrs_amd.imatrix import CollectedImatrixData
    data = CollectedImatrixData({"model.layers.0.mlp.down_proj": np.arange(5, dtype=np.float32), "blk.1.ffn_up": np.array([0.5, -1.25], np.float32)})
    f = tmp_path / "a.cimatrix"
    data.save_imatrix(f)
    raw = f.read_bytes()
    assert struct.unpack("<Q", raw[:8])[0] == 2
    klen = struct.unpack("<Q", raw[8:16])[0]
    assert raw[16:16 + klen].decode() == "model.layers.0.mlp.down_proj"
    assert struct.unpack("<Q", raw[16 + klen:24 + klen])[0] == 5
    back = CollectedImatrixData.load_imatrix(f)
    assert set(back) == set(data) and all(np.array_equal(back[k], data[k]) for k in data)
    with pytest.raises(ValueError, match="Expected a .cimatrix file"):
        data.save_imatrix(tmp_path / "a.imatrix")
    (tmp_path / "bad.cimatrix").write_bytes(raw[:-3])
    with pytest.raises(ValueError):
        CollectedImatrixData.load_imatrix(tmp_path / "bad.cimatrix")


# ------------------------------------------------------------------------------------------------ the reference's own tests, on the device
def _manual_imatrix(rows, ncalls):
    s = np.zeros(rows.shape[1], np.float32)
    for r in rows:
        s += r * r
    return s / np.float32(rows.shape[0]) * np.float32(ncalls)


@pytest.mark.gpu
def test_gguf_layer_collects_imatrix_and_snapshots(oracle, dev):
    """imatrix.rs:248-290 (gguf_layer_collects_imatrix, snapshot_visible_through_layer) and :440-452 (disabled_stats_are_free_of_state)."""
    import torch
    from mistralrs_amd import isq
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.gguf.matmul import GgufMatMul
    g = torch.Generator(device="cpu").manual_seed(0)
    w = torch.randn(64, 32, generator=g).to(dev)
    layer = GgufMatMul(isq.quantize(w, GgmlDType.Q8_0))
    assert layer.stats_snapshot() is None
    x = torch.randn(4, 32, generator=g).to(dev)
    layer.forward_raw(x)
    with pytest.raises(ValueError, match="not tracking stats"):
        layer.end_track_stats()
    layer.begin_track_stats()
    assert layer.stats_snapshot() == (0, 0)
    x1, x2 = torch.randn(4, 32, generator=g).to(dev), torch.randn(3, 32, generator=g).to(dev)
    layer.forward_raw(x1)
    assert layer.stats_snapshot() == (1, 4)
    layer.forward_raw(x2)
    got = layer.end_track_stats().cpu().numpy()
    want = _manual_imatrix(np.concatenate([x1.cpu().numpy(), x2.cpu().numpy()]), 2)
    assert np.abs(got - want).max() < 1e-4
    assert layer.stats_snapshot() is None


@pytest.mark.gpu
def test_routed_stats(oracle, dev):
    """imatrix.rs:292-341: routed_stats_scatter_per_expert, routed_stats_rank3_per_slot, dead_expert_rows_are_zero_not_nan."""
    import torch
    from mistralrs_amd.imatrix import ImatrixLayerStats
    st = ImatrixLayerStats.empty()
    st.enable_routed(3, 2, dev)
    st.process_routed(torch.tensor([[1.0, 2.0], [3.0, 4.0]], device=dev), torch.tensor([[0, 2], [2, 1]], device=dev))
    assert st.snapshot() == (1, 4)
    m = st.compute_imatrix().cpu().numpy()
    np.testing.assert_array_equal(m, np.array([[1, 4], [9, 16], [5, 10]], np.float32))
    st = ImatrixLayerStats.empty()
    st.enable_routed(2, 2, dev)
    st.process_routed(torch.tensor([[[1.0, 0.0], [0.0, 2.0]]], device=dev), torch.tensor([[0, 1]], device=dev))
    np.testing.assert_array_equal(st.compute_imatrix().cpu().numpy(), np.array([[1, 0], [0, 4]], np.float32))
    st = ImatrixLayerStats.empty()
    st.enable_routed(3, 2, dev)
    st.process_routed(torch.tensor([[1.0, 1.0]], device=dev), torch.tensor([[0, 0]], device=dev))
    m = st.compute_imatrix().cpu().numpy()
    assert (m[1] == 0).all() and np.isfinite(m).all()
    st.process(torch.ones(2, 2, device=dev))   # a stray plain forward through a routed layer contributes nothing
    assert st.snapshot() == (1, 2)


@pytest.mark.gpu
def test_expert_stack_per_slab_commutes_and_apply_isq_routes_imatrix(oracle, dev):
    """imatrix.rs:343-438: a shared vector through the per-slab assembly is byte-identical to quantizing the whole stack; per-expert vectors
    differentiate the slabs and leave expert 0 (same vector) bit-identical; apply_isq on a rank-3 weight routes the vector per slab; a vector of
    the wrong length is ignored with a warning (gguf/mod.rs:225-232); an all-zero per-expert vector falls back to the plain quantizer (:248)."""
    import torch
    from mistralrs_amd import isq
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.gguf.matmul import GgufMatMul
    g = torch.Generator(device="cpu").manual_seed(1)
    stack = torch.randn(3, 8, 256, generator=g).to(dev)
    shared = np.array([1.0 + (i % 7) for i in range(256)], np.float32)
    per_slab = isq.quantize_expert_stack(stack, GgmlDType.Q4K, shared)
    whole = isq.quantize_imatrix(stack.reshape(24, 256), shared, GgmlDType.Q4K)
    assert per_slab.shape == (3, 8, 256) and torch.equal(per_slab.data, whole.data)
    want = oracle.quantize_imatrix(oracle.Q4_K, stack.reshape(24, 256).cpu().numpy(), shared)
    np.testing.assert_array_equal(whole.data.cpu().numpy(), want.reshape(-1))
    per_expert = np.concatenate([shared, 100.0 + np.arange(256, dtype=np.float32), np.ones(256, np.float32)])
    routed = isq.quantize_expert_stack(stack, GgmlDType.Q4K, per_expert)
    assert not torch.equal(routed.data, whole.data)
    nb = whole.data.numel() // 3
    assert torch.equal(routed.data[:nb], whole.data[:nb])
    with pytest.warns(UserWarning, match="matches neither in_dim"):
        ignored = isq.quantize_expert_stack(stack, GgmlDType.Q4K, np.ones(100, np.float32))
    plain = isq.quantize(stack.reshape(24, 256), GgmlDType.Q4K)
    assert torch.equal(ignored.data, plain.data)
    dead = np.concatenate([shared, np.zeros(256, np.float32), shared])            # expert 1 saw no traffic
    mixed = isq.quantize_expert_stack(stack, GgmlDType.Q4K, dead)
    assert torch.equal(mixed.data[nb:2 * nb], plain.data[nb:2 * nb]) and torch.equal(mixed.data[:nb], whole.data[:nb])
    # apply_isq on a quantized resident with a rank-3 weight
    skew = np.concatenate([np.ones(256, np.float32), np.where(np.arange(256) < 128, 1000.0, 0.001).astype(np.float32)])
    two = stack[:2].contiguous()
    q8 = GgufMatMul(isq.quantize_expert_stack(two, GgmlDType.Q8_0))
    with_im = q8.apply_isq(GgmlDType.Q4K, skew).dequantize_w()
    without = q8.apply_isq(GgmlDType.Q4K).dequantize_w()
    assert tuple(with_im.shape) == (2, 8, 256) and float((with_im - without).abs().max()) > 0
    # and on a plain [N, K] layer: the weighted quantizer, against the oracle
    lin = GgufMatMul(isq.quantize(stack[0], GgmlDType.Q8_0))
    q6 = lin.apply_isq(GgmlDType.Q6K, shared)
    dense = lin.dequantize_w().cpu().numpy()
    np.testing.assert_array_equal(q6.get_qtensor().data.cpu().numpy(), oracle.quantize_imatrix(oracle.Q6_K, dense, shared).reshape(-1))
