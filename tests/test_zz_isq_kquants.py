"""ISQ on the device for every GGML target the GGUF kernels read (`mrs_isq_quantize`: Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K; reference:
`generate_isq!` -> candle `QTensor::quantize` on the host cores, mistralrs-quant/src/utils/isq.rs:323-361).  The device blocks must be
BIT-IDENTICAL to the oracle's restatement of GGML's reference quantizers (make_qkx2_quants / make_qx_quants searches included), from f32,
f16 and bf16 sources, including degenerate blocks.  Same body on the wave64 host emulation (CPU) and on the MI355X (`-m gpu`);
plus the reference's dtype fallback chain (utils/isq.rs:247-287) on the host."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.util import round_through

TARGETS = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
SRC_CODE = {"f32": 0, "f16": 1, "bf16": 30}


def _weights(src, n=19, k=1024, seed=0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((n, k)) * rng.uniform(1e-3, 2.0, (n, 1))).astype(np.float32)
    w[3, :256] = 0.0                           # an all-zero superblock
    w[5, 256:288] = 0.37                       # a constant sub-block (max == min)
    w[7, :] *= 1e-9                            # tiny magnitudes
    w[9, 512:768] = np.abs(w[9, 512:768])      # all-positive superblock (min clamps to 0)
    w[11, 100] = 60000.0                       # an outlier that dominates its block
    return round_through(w, src) if src != "f32" else w


def check_isq(oracle, be, tname, src):
    t = getattr(oracle, tname)
    w = _weights(src, seed=t)
    n, k = w.shape
    want = oracle.quantize(t, w)
    wb = be.buf(w, None if src == "f32" else src)
    out = be.buf(np.full(want.shape, 0xAA, dtype=np.uint8))
    fn = be.sym("mrs_isq_quantize", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p], C.c_int)
    assert fn(wb.ptr, SRC_CODE[src], out.ptr, n * k, t, be.stream) == 0
    np.testing.assert_array_equal(out.numpy(), want)
    assert fn(wb.ptr, SRC_CODE[src], out.ptr, 48, t, be.stream) == -1   # not a multiple of the block size
    assert fn(wb.ptr, 5, out.ptr, n * k, t, be.stream) == -1             # unknown source dtype


@pytest.mark.parametrize("tname", TARGETS)
@pytest.mark.parametrize("src", ["f32", "bf16", "f16"])
def test_isq_host_emulation(oracle, tname, src):
    check_isq(oracle, HostBackend(), tname, src)


def test_isq_partial_workgroups_host_emulation(oracle):
    """Superblock counts that leave lane groups / half-waves of the last workgroup idle (dead groups must not disturb live ones)."""
    be = HostBackend()
    fn = be.sym("mrs_isq_quantize", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p], C.c_int)
    rng = np.random.default_rng(1)
    for t, nblk in ((oracle.Q4_K, 33), (oracle.Q5_K, 1), (oracle.Q6_K, 17), (oracle.Q2_K, 17), (oracle.Q3_K, 5), (oracle.Q8_0, 9), (oracle.Q4_0, 257)):
        k = oracle.block_size(t) * nblk
        w = rng.standard_normal((1, k)).astype(np.float32)
        want = oracle.quantize(t, w)
        wb, out = be.buf(w), be.buf(np.zeros_like(want))
        assert fn(wb.ptr, 0, out.ptr, k, t, None) == 0
        np.testing.assert_array_equal(out.numpy(), want)


def test_quantization_behaviour_fallback_chain():
    """get_quantization_behaviour (utils/isq.rs:263-287): K-quants fall back to a 32-wide format when K % 256 != 0, skip when K % 32 != 0."""
    from mistralrs_amd.gguf.qtensor import GgmlDType as G
    from mistralrs_amd.isq import get_quantization_behaviour as qb
    assert qb((4096, 4096), G.Q4K) == G.Q4K
    assert qb((4096, 4000), G.Q4K) == G.Q4_1      # 4000 = 125 * 32
    assert qb((4096, 4000), G.Q5K) == G.Q5_0
    assert qb((4096, 4000), G.Q6K) == G.Q5_1
    assert qb((4096, 4000), G.Q2K) == G.Q4_0 and qb((4096, 4000), G.Q3K) == G.Q4_0
    assert qb((4096, 4001), G.Q4K) is None        # no block size divides it: skip
    assert qb((4096, 4001), G.Q8_0) is None
    assert qb((4096, 4096), G.F32) is None
    assert qb((), G.Q4K) is None


@pytest.mark.gpu
@pytest.mark.parametrize("tname", TARGETS)
@pytest.mark.parametrize("src", ["f32", "bf16", "f16"])
def test_isq_abi_gpu(oracle, dev, tname, src):
    check_isq(oracle, GpuBackend(dev), tname, src)


@pytest.mark.gpu
def test_isq_wrapper_gpu(oracle, dev):
    """mistralrs_amd.isq.quantize: QTensor with the oracle's bytes; the quantized tensor feeds the decode GEMV."""
    import torch
    from mistralrs_amd import isq
    from mistralrs_amd.gguf import GgmlDType, fast_mmvq
    w = _weights("bf16", n=32, k=512, seed=5)
    qt = isq.quantize(torch.from_numpy(w).to(dev).to(torch.bfloat16), GgmlDType.Q4K)
    want = oracle.quantize(oracle.Q4_K, w)
    np.testing.assert_array_equal(qt.data.cpu().numpy().reshape(32, -1), want)
    x = np.random.default_rng(2).standard_normal((1, 512)).astype(np.float32)
    got = fast_mmvq.plain(qt, torch.from_numpy(x).to(dev)).cpu().numpy()
    ref, mag = oracle.matmul_q8_1_mag(oracle.Q4_K, want, 32, 512, oracle.quantize_q8_1(x))
    assert (np.abs(got - ref) <= 8 * 2.0 ** -23 * np.sqrt(512 / 16) * mag + 1e-30).all()
    with pytest.raises(ValueError, match="multiple of the Q4K block size"):
        isq.quantize(torch.zeros(4, 480, device=dev), GgmlDType.Q4K)


def test_isq_host_tables():
    """`IsqType::supports_imatrix` (lib.rs:1038-1043) and `block_pack_factor` (lib.rs:946-957) for the GGML targets."""
    from mistralrs_amd import isq
    from mistralrs_amd.gguf.qtensor import GgmlDType as G
    assert [t.name for t in G if isq.supports_imatrix(t)] == ["Q2K", "Q3K", "Q4K", "Q5K", "Q6K"]
    assert all(isq.supports_imatrix(t) == isq.imatrix_capable(t) for t in G)
    # bf16 source: Q4_K 256 * 2 / 144 = 3.55 -> 3 (256 / 3 * 2 = 170 >= 144); Q8_0 32 * 2 / 34 -> 1; Q4_0 64 / 18 -> 3 (32 / 3 * 2 = 20 >= 18); Q6_K 512 / 210 -> 2
    assert isq.pack_factor(G.Q4K) == 3 and isq.pack_factor(G.Q8_0) == 1 and isq.pack_factor(G.Q4_0) == 3 and isq.pack_factor(G.Q6K) == 2
    assert isq.pack_factor(G.Q2K) == 6 and isq.pack_factor(G.Q4K, 4) == 7   # 512 / 84 = 6; f32 source: 1024 / 144 = 7 (256 / 7 * 4 = 144 >= 144)
