"""HQQ slice (SURVEY 8a `HqqLayer::forward_raw`): unpack + dequantize and bit-packing kernels behind the reference's C ABI
(mistralrs-quant/src/hqq/ffi.rs, bitpack_ffi.rs), the HqqLayer mirror, and their oracle (oracle/hqq_oracle.py).

CPU (-m "not gpu"): the oracle is pinned bit-for-bit to the reference's own kernels run on the host (oracle/_ref/libref_hqq.so, when
built) and to the frozen golden vectors those kernels produced (always); pack / unpack round trip; quantizer round-trip error.
GPU (-m gpu): every dequantize_* symbol (5 bit widths x f32 / f16 / bf16) and every launch_pack_* symbol against the oracle, bit-exact,
vectorised and scalar paths; the golden vectors; size-independent round trip at layer size; HqqLayer.quantize / forward vs the oracle.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import hqq_oracle as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "hqq_*bit.npz")))
BITS = [8, 4, 3, 2, 1]


def _case(bits, h, w, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 2 ** bits, size=(H.PACK[bits] * h, w)).astype(np.uint32 if bits == 3 else np.uint8)
    scale = (rng.uniform(0.001, 0.05, w) * rng.choice([1.0, -1.0], w)).astype(np.float32)
    zero = rng.uniform(0.0, 2 ** bits - 1.0, w).astype(np.float32)
    return q, scale, zero


def test_golden_present():
    assert len(GOLD) == 5, "tests/golden/hqq_*.npz missing: run scripts/gen_golden.py in the build container"


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(f)[:-4] for f in GOLD])
def test_oracle_reproduces_reference_golden(path):
    g = np.load(path)
    bits = int(g["bits"])
    np.testing.assert_array_equal(H.pack(bits, g["q"]), g["packed_ref"])
    np.testing.assert_array_equal(H.dequantize(bits, g["packed_ref"], g["scale"], g["zero"], "f32").view(np.uint32), g["out_ref"].view(np.uint32))


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("h,w", [(1, 1), (5, 33), (16, 64)])
def test_oracle_matches_reference_kernels(bits, h, w):
    p = os.path.join(ROOT, "oracle", "_ref", "libref_hqq.so")
    if not os.path.exists(p):
        pytest.skip("libref_hqq.so not built (needs /root/reference: sh oracle/build_ref.sh /root/reference)")
    lib = C.CDLL(p)
    q, scale, zero = _case(bits, h, w, 10 * bits + h)
    packed = np.zeros((h, w), dtype=np.int32 if bits == 3 else np.uint8)
    assert lib.ref_hqq_pack(bits, q.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p), C.c_size_t(q.shape[0]), C.c_size_t(w)) == 0
    np.testing.assert_array_equal(H.pack(bits, q), packed)
    out = np.empty((q.shape[0], w), dtype=np.float32)
    assert lib.ref_hqq_dequantize_f32(bits, packed.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p), zero.ctypes.data_as(C.c_void_p),
                                      out.ctypes.data_as(C.c_void_p), h, w) == 0
    np.testing.assert_array_equal(H.dequantize(bits, packed, scale, zero, "f32").view(np.uint32), out.view(np.uint32))


@pytest.mark.parametrize("bits", BITS)
def test_oracle_pack_unpack_round_trip_and_ragged_rows(bits):
    q, _, _ = _case(bits, 9, 17, bits)
    np.testing.assert_array_equal(H.unpack(bits, H.pack(bits, q)), q.astype(np.uint8))
    if bits != 8:  # rows beyond P * (rows // P) are ignored, like the reference kernels do
        extra = np.concatenate([q, q[:1]], axis=0)
        assert H.pack(bits, extra).shape == (extra.shape[0] // H.PACK[bits], 17)


@pytest.mark.parametrize("bits,tol", [(8, 0.004), (4, 0.06), (3, 0.13), (2, 0.3)])
def test_oracle_quantizer_round_trip(bits, tol):
    rng = np.random.default_rng(bits)
    gs = 60 if bits == 3 else 64  # 3 bit packs ten rows per i32: a group of 60 rows packs exactly
    w = (rng.standard_normal((gs * 4, 32)) * 0.05).astype(np.float32)
    wq, scale, zero = H.quantize(w, bits, gs)
    back = H.dequantize(bits, wq, scale.reshape(-1), zero.reshape(-1)).reshape(-1)[: w.size].reshape(w.shape)
    span = w.reshape(gs, -1).max(0) - w.reshape(gs, -1).min(0)
    assert np.abs(back - w).mean() <= tol * span.mean(), (np.abs(back - w).mean(), span.mean())


def test_mirror_rejects_cpu_tensors_and_bad_config():
    import torch
    from mistralrs_amd.hqq import HqqConfig, HqqLayer
    with pytest.raises(ValueError, match="bits"):
        HqqConfig(bits=5)
    with pytest.raises(ValueError, match="GPU"):
        HqqLayer.quantize(torch.zeros(64, 8), HqqConfig())


# ------------------------------------------------------------------------------------------------------------------ GPU
def _t(dev, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("h,w", [(5, 33), (7, 52), (48, 1024), (0, 16)])
def test_dequantize_kernels_vs_oracle(dev, bits, dtype, h, w):
    import torch
    from mistralrs_amd import hqq
    from tests.util import round_through, torch_dtype
    q, scale, zero = _case(bits, h, w, 100 * bits + h + w)
    scale, zero = round_through(scale, dtype), round_through(zero, dtype)
    packed = H.pack(bits, q) if h else np.zeros((0, w), dtype=np.int32 if bits == 3 else np.uint8)
    td = torch_dtype(dtype)
    got = hqq.dequantize_packed(bits, _t(dev, packed), _t(dev, scale).to(td), _t(dev, zero).to(td))
    assert got.dtype == td and tuple(got.shape) == (H.PACK[bits] * h, w)
    if h:
        want = H.dequantize(bits, packed, scale, zero, dtype)
        np.testing.assert_array_equal(got.float().cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(f)[:-4] for f in GOLD])
def test_hip_kernels_reproduce_reference_golden(dev, path):
    from mistralrs_amd import hqq
    g = np.load(path)
    bits = int(g["bits"])
    packed = hqq.pack(bits, _t(dev, g["q"].astype(np.int32 if bits == 3 else np.uint8)))
    np.testing.assert_array_equal(packed.cpu().numpy(), g["packed_ref"])
    out = hqq.dequantize_packed(bits, packed, _t(dev, g["scale"]), _t(dev, g["zero"]))
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), g["out_ref"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("bits", BITS)
def test_pack_dequantize_round_trip_at_layer_size(dev, bits):
    """Size-independent property at a real layer size (4096 x 14336 weights): with zero = 0 and scale = 1 the dequantized tensor IS
    the unpacked value tensor, for every chunk position."""
    import torch
    from mistralrs_amd import hqq
    rows = 4090 if bits == 3 else 4096
    w = 14336
    torch.manual_seed(bits)
    q = torch.randint(0, 2 ** bits, (rows, w), device=dev, dtype=torch.int32)
    packed = hqq.pack(bits, q)
    assert tuple(packed.shape) == (rows // H.PACK[bits], w)
    one, zero = torch.ones(w, device=dev), torch.zeros(w, device=dev)
    back = hqq.dequantize_packed(bits, packed, one, zero)
    assert torch.equal(back.to(torch.int32), q)
    with pytest.raises(ValueError, match="same"):
        hqq.dequantize_packed(bits, packed, one, zero.half())
    with pytest.raises(ValueError, match="one per group column"):
        hqq.dequantize_packed(bits, packed, one[:-1], zero[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 4, 3, 2, 1])
def test_hqq_layer_quantize_and_forward(dev, bits):
    import torch
    from mistralrs_amd.hqq import HqqConfig, HqqLayer
    rng = np.random.default_rng(50 + bits)
    n, k = (120, 64) if bits == 3 else (128, 64)  # 3 bit: group of 60 rows packs exactly
    gs = 60 if bits == 3 else 64
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    layer = HqqLayer.quantize(_t(dev, w), HqqConfig(bits=bits, group_size=gs))
    wq_o, scale_o, zero_o = H.quantize(w, bits, gs)
    # same algorithm in f32 on both sides; reduction order / pow differ in the last bit, so a few values may land on the other side of .5
    same = (layer.w_q.cpu().numpy() == wq_o).mean()
    assert same >= 0.995, same
    np.testing.assert_allclose(layer.scales.cpu().numpy(), scale_o, rtol=1e-6)
    np.testing.assert_allclose(layer.zeros.cpu().numpy(), zero_o, rtol=1e-3, atol=1e-3)
    # the layer's own dequantize == oracle dequantize of the layer's own packed data, bit for bit
    want_w = H.dequantize(bits, layer.w_q.cpu().numpy(), layer.scales.cpu().numpy().reshape(-1), layer.zeros.cpu().numpy().reshape(-1))
    want_w = want_w.reshape(-1)[: n * k].reshape(n, k)
    np.testing.assert_array_equal(layer.dequantize().cpu().numpy().view(np.uint32), want_w.view(np.uint32))
    x = rng.standard_normal((5, k)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    got = layer.with_bias(_t(dev, bias)).forward(_t(dev, x)).cpu().numpy()
    want = x.astype(np.float64) @ want_w.astype(np.float64).T + bias
    mag = np.abs(x).astype(np.float64) @ np.abs(want_w).astype(np.float64).T + np.abs(bias)
    assert (np.abs(got - want) <= 2.0 ** -18 * mag).all()
    half = layer.to_dtype(torch.bfloat16)
    assert half.dequantize().dtype == torch.bfloat16


@pytest.mark.gpu
def test_hqq_3bit_default_group_size_round_trip(dev):
    """3 bit with the default group size 64: 64 rows are not a multiple of the 10 values per i32, the reference zero-pads the rows before packing
    (hqq/mod.rs:401-410) and trims after the dequantize; quantize -> dequantize must give back a [n, k] tensor close to the input."""
    import torch
    from mistralrs_amd.hqq import HqqConfig, HqqLayer, pack
    rng = np.random.default_rng(7)
    n, k = 128, 64
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    layer = HqqLayer.quantize(_t(dev, w), HqqConfig(bits=3, group_size=64))
    assert layer.w_q.shape[0] == 7  # ceil(64 / 10) packed rows per group column
    back = layer.dequantize().cpu().numpy()
    assert back.shape == (n, k)
    assert np.abs(back - w).max() <= 0.25 * np.abs(w).max()  # 8 levels per group
    # pack() itself: a [64, w] block packs to 7 rows, the padding rows read back as zeros
    q = torch.from_numpy(rng.integers(0, 8, (64, 16)).astype(np.int32)).to(dev)
    assert pack(3, q).shape == (7, 16)


def _check_fused_gemv(be_kind, dev, bits, dtype, n, k, b):
    """mrs_hqq_gemv vs dequantize_w() @ x: the dequantized values are the reference's bit for bit, so the only freedom is the f32 summation order."""
    import torch
    from mistralrs_amd.hqq import HqqConfig, HqqLayer
    rng = np.random.default_rng(bits * 100 + n)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    layer = HqqLayer.quantize(_t(dev, w), HqqConfig(bits=bits, group_size=64))
    if dtype != torch.float32:
        layer = layer.to_dtype(dtype)
    bias = _t(dev, rng.standard_normal(n).astype(np.float32)).to(dtype)
    x = _t(dev, rng.standard_normal((b, k)).astype(np.float32)).to(dtype)
    got = layer.with_bias(bias).forward(x).float().cpu().numpy()
    wd = layer.dequantize().float().cpu().numpy().astype(np.float64)
    xf = x.float().cpu().numpy().astype(np.float64)
    want = xf @ wd.T + bias.float().cpu().numpy()
    mag = np.abs(xf) @ np.abs(wd).T + np.abs(bias.float().cpu().numpy())
    tol = 2.0 ** -18 * mag if dtype == torch.float32 else 2.0 ** -8 * np.abs(want) + 2.0 ** -18 * mag  # 16-bit outputs: one rounding of the result
    assert (np.abs(got - want) <= tol).all(), float((np.abs(got - want) / mag).max())


@pytest.mark.gpu
@pytest.mark.parametrize("bits,n,k,b", [(4, 128, 64, 1), (4, 256, 1024, 3), (8, 128, 512, 2), (4, 4096, 4096, 1), (8, 1024, 4096, 8), (4, 1024, 14336, 1)])
def test_fused_hqq_gemv_vs_dequantize_matmul(dev, bits, n, k, b):
    import torch
    for dtype in (torch.float32, torch.bfloat16):
        _check_fused_gemv("gpu", dev, bits, dtype, n, k, b)


@pytest.mark.parametrize("bits,n,k,b", [(4, 128, 64, 1), (4, 64, 1040, 2), (8, 128, 96, 3), (4, 64, 2096, 1)])
def test_fused_hqq_gemv_host_emulation(bits, n, k, b):
    """The fused kernel on the wave64 host emulation (f32) vs the oracle's dequantize @ x."""
    from tests.abi_backends import HostBackend
    be = HostBackend()
    rng = np.random.default_rng(bits + n + k)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    wq, scale, zero = H.quantize(w, bits, 64)
    inv = scale.reshape(-1).astype(np.float32)  # H.quantize already returns 1 / scale, the dequantizer's multiplier
    wd = H.dequantize(bits, wq, inv, zero.reshape(-1).astype(np.float32)).reshape(-1)[: n * k].reshape(n, k)
    x = rng.standard_normal((b, k)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    fn = be.sym("mrs_hqq_gemv", [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int)
    bw, bs, bz, bb, bx, bo = be.buf(wq), be.buf(inv), be.buf(zero.reshape(-1).astype(np.float32)), be.buf(bias), be.buf(x), be.buf(np.zeros((b, n), dtype=np.float32))
    assert fn(bits, 0, bw.ptr, bs.ptr, bz.ptr, bb.ptr, bx.ptr, k, bo.ptr, n, n, k, b, None) == 0
    want = x.astype(np.float64) @ wd.astype(np.float64).T + bias
    mag = np.abs(x).astype(np.float64) @ np.abs(wd).astype(np.float64).T + np.abs(bias)
    assert (np.abs(bo.numpy() - want) <= 2.0 ** -18 * mag).all()
    assert fn(3, 0, bw.ptr, bs.ptr, bz.ptr, None, bx.ptr, k, bo.ptr, n, n, k, b, None) == -1  # other bit widths: dequantize + dense matmul
    assert fn(bits, 0, bw.ptr, bs.ptr, bz.ptr, None, bx.ptr, k, bo.ptr, n, n, k, 9, None) == -1


def test_fused_hqq_gemv_refuses_rows_that_are_not_16_byte_multiples():
    """K % 16 != 0: the 16-byte weight loads would be misaligned -> -1, HqqLayer.forward keeps dequantize + matmul for such layers."""
    from tests.abi_backends import HostBackend
    be = HostBackend()
    d = be.buf(np.zeros(1 << 16, np.uint8))
    fn = be.sym("mrs_hqq_gemv", [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int)
    assert fn(4, 0, d.ptr, d.ptr, d.ptr, None, d.ptr, 1028, d.ptr, 64, 64, 1028, 1, be.stream) == -1
