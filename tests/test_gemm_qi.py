"""Prompt GEMM in the reference CPU path's arithmetic on the matrix cores (csrc/ext_gemm_qi.hip): Q8_K activation rows (engine-order RmsNorm + candle's
quantizer) x Q4_K / Q5_K / Q6_K weights through v_mfma_f32_32x32x16_f16 on exact small integers, combined in the decode engine's f32 order (ORD-U).
Every output row equals the decode engine's GEMV of that token (oracle/cpu_path_oracle.c orc_gemv_engine) BIT FOR BIT -- so does the GPU result against the
batch-1 decode kernel itself.  Also: the matrix core's f32 accumulator is exact on the integer operands the kernel feeds it (adversarial magnitudes)."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend

VP, CI = C.c_void_p, C.c_int


def run_gemm(O, be, tname, n, k, T, norm, glu=False, acc=False, seed=0, packed=None, x_override=None):
    t = getattr(O, tname)
    rng = np.random.default_rng(seed)
    if packed is None:
        packed = O.quantize(t, (rng.standard_normal((n, k)) * 0.05).astype(np.float32)).reshape(n, -1)
    nb = be.sym("mrs_gemm_qi_repack_bytes", [CI, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, k)
    assert nb > 0
    src, wq = be.buf(np.ascontiguousarray(packed).reshape(-1)), be.buf(np.full(nb, 0xA5, np.uint8))
    assert be.sym("mrs_gemm_qi_repack", [VP, CI, C.c_longlong, C.c_longlong, VP, VP], CI)(src.ptr, t, n, k, wq.ptr, be.stream) == 0
    x = rng.standard_normal((T, k)).astype(np.float32)
    x2 = rng.standard_normal((T, k)).astype(np.float32) if glu else None
    if T > 2:
        x[1, : min(256, k)] = 0.0  # an all-zero activation block
    if x_override is not None:
        x = np.ascontiguousarray(x_override, dtype=np.float32)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32) if norm else None
    ab = be.sym("mrs_qi_act_bytes", [CI, CI], C.c_size_t)(T, k)
    act = be.buf(np.zeros(ab, np.uint8))
    xb = be.buf(x)
    x2b = be.buf(x2) if glu else None
    nwb = be.buf(nw) if norm else None
    tmp = be.buf(np.zeros((T, k), np.float32)) if glu else None
    rc = be.sym("mrs_qi_quantize_for", [CI, VP, VP, CI, VP, C.c_float, CI, CI, VP, VP, VP], CI)(t, xb.ptr, x2b.ptr if glu else None, k, nwb.ptr if norm else None, 1e-5, T, k, act.ptr,
                                                                                                   tmp.ptr if glu else None, be.stream)
    assert rc == 0, rc
    base = rng.standard_normal((T, n)).astype(np.float32) if acc else np.full((T, n), np.nan, np.float32)
    ob = be.buf(base.copy())
    rc = be.sym("mrs_gemm_qi", [VP, CI, CI, CI, VP, CI, VP, CI, CI, VP], CI)(wq.ptr, t, n, k, act.ptr, T, ob.ptr, n, 1 if acc else 0, be.stream)
    assert rc == 0, rc
    got = ob.numpy()
    # the split launch (one workgroup per run of superblocks + the reduce kernel): the same additions in the same order
    ws = be.buf(np.zeros(4 * T * n, np.float32))
    ob2 = be.buf(base.copy())
    rc = be.sym("mrs_gemm_qi_ws", [VP, CI, CI, CI, VP, CI, VP, CI, CI, VP, C.c_size_t, VP], CI)(wq.ptr, t, n, k, act.ptr, T, ob2.ptr, n, 1 if acc else 0, ws.ptr, 4 * T * n * 4, be.stream)
    assert rc == 0, rc
    assert np.array_equal(ob2.numpy(), got, equal_nan=True), "split launch differs from the single-pass launch"
    xin = O.fused_glu_engine(x, x2) if glu else x
    if norm:
        xin = O.rms_norm_engine(xin, nw, 1e-5)
    eng = np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in xin], axis=0)
    want = base + eng if acc else eng
    assert np.array_equal(got, want), (tname, n, k, T, int((got != want).sum()), float(np.abs(got - want).max()))
    return packed, x, got


CASES = [("Q8_0", 40, 512, 5, 0), ("Q8_0", 66, 1280, 34, 1), ("Q8_0", 130, 1024, 133, 1), ("Q8_0", 32, 2816, 3, 0), ("Q5_K", 40, 512, 5, 0), ("Q5_K", 66, 1280, 34, 1), ("Q5_K", 32, 2816, 3, 0), ("Q4_K", 40, 512, 5, 0), ("Q4_K", 36, 1280, 4, 1), ("Q4_K", 130, 1024, 33, 1), ("Q4_K", 64, 4096, 7, 1), ("Q4_K", 33, 2816, 3, 0), ("Q6_K", 40, 512, 5, 0), ("Q6_K", 70, 1024, 130, 1),
         ("Q6_K", 32, 3584, 4, 0)]


@pytest.mark.parametrize("tname,n,k,T,norm", CASES)
def test_gemm_qi_host_emulation(oracle, tname, n, k, T, norm):
    run_gemm(oracle, HostBackend(), tname, n, k, T, norm, seed=n + k + T)


def test_gemm_qi_glu_and_accumulate_host_emulation(oracle):
    run_gemm(oracle, HostBackend(), "Q4_K", 48, 768, 6, 0, glu=True, seed=1)
    run_gemm(oracle, HostBackend(), "Q6_K", 36, 512, 9, 0, acc=True, seed=2)
    run_gemm(oracle, HostBackend(), "Q5_K", 40, 768, 6, 0, glu=True, acc=True, seed=3)
    run_gemm(oracle, HostBackend(), "Q8_0", 40, 768, 6, 0, glu=True, acc=True, seed=4)


def q5k_extreme_blocks(n, k):
    """block_q5_K bytes with every 6-bit scale 63 and every 5-bit value 31 (d = 1, dmin = 0): with activation rows whose Q8_K quants are all -128 / all 127 a superblock's
    integer sum is -64.0 M / +63.5 M -- the four f32 accumulators of the Q5_K branch each hold 16.0 M < 2^24, the largest they can be asked to"""
    blk = np.zeros(176, np.uint8)
    blk[0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
    blk[4:8] = 0xFF    # sc 0..3 = 63, top bits: the high bits of sc 4..7
    blk[8:12] = 0x00   # mn 0..3 = 0, high bits of mn 4..7 = 0
    blk[12:16] = 0x0F  # low nibbles of sc 4..7 = 15, of mn 4..7 = 0
    blk[16:176] = 0xFF  # qh, qs: every value 31
    out = np.tile(blk, (n, k // 256)).reshape(n, k // 256, 176)
    out[1::2, :, 16:176] = 0x00  # odd rows: every value 0 -- the kernel's signed operand sc (q - 16) = -1008 at its largest, isum = XA + XB + 16 S cancels to 0 exactly
    out[2::4, ::2, 16:48] = 0x00  # and rows mixing 15 / 31 by superblock
    return out.reshape(n, -1)


def run_q5k_extremes(O, be):
    n, k, T = 33, 1024, 4
    packed = q5k_extreme_blocks(n, k)
    x = np.ones((T, k), np.float32)          # all equal and positive: max = +x, iscale = -128 / x, every quant -128
    x[1] = -3.0                              # all equal and negative: every quant min(127, 128) = 127
    x[2, ::2] = -1.0                         # alternating signs: the first maximum is +1 -> quants -128 / +127
    x[3, :256] = 0.0
    run_gemm(O, be, "Q5_K", n, k, T, 0, seed=5, packed=packed, x_override=x)


def test_gemm_qi_q5k_extreme_magnitudes_host_emulation(oracle):
    run_q5k_extremes(oracle, HostBackend())


@pytest.mark.gpu
def test_gemm_qi_q5k_extreme_magnitudes_gpu(oracle, dev):
    run_q5k_extremes(oracle, GpuBackend(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,T,norm", CASES + [("Q4_K", 4096, 4096, 512, 1), ("Q5_K", 2048, 4096, 300, 1), ("Q5_K", 512, 14336, 129, 0), ("Q4_K", 1024, 14336, 300, 0), ("Q6_K", 1024, 4096, 512, 1), ("Q6_K", 512, 14336, 129, 0),
                                                     ("Q4_K", 28672, 4096, 64, 1), ("Q8_0", 4096, 4096, 512, 1), ("Q8_0", 1024, 14336, 300, 0)])
def test_gemm_qi_gpu(oracle, dev, tname, n, k, T, norm):
    run_gemm(oracle, GpuBackend(dev), tname, n, k, T, norm, seed=n + k + T)


@pytest.mark.gpu
def test_gemm_qi_glu_and_accumulate_gpu(oracle, dev):
    run_gemm(oracle, GpuBackend(dev), "Q4_K", 512, 14336, 70, 0, glu=True, seed=1)
    run_gemm(oracle, GpuBackend(dev), "Q6_K", 4096, 4096, 200, 0, acc=True, seed=2)
    run_gemm(oracle, GpuBackend(dev), "Q8_0", 512, 14336, 70, 0, glu=True, acc=True, seed=3)


@pytest.mark.gpu
def test_mfma_f16_accumulates_small_integers_exactly(dev):
    """The premise of the kernel: every partial sum of a superblock is an integer below 2^24, and v_mfma_f32_32x32x16_f16 keeps it exact."""
    import torch
    from mistralrs_amd import _lib
    L = _lib.load("ext")
    L.mrs_mfma_f16_int_probe.argtypes = [VP, VP, VP, CI, VP]

    def run(A, B):
        ks = A.shape[0]
        lanes = np.arange(64)
        a_op = np.zeros((ks, 64, 8), np.float16); b_op = np.zeros((ks, 64, 8), np.float16)
        for j in range(8):
            a_op[:, :, j] = A[:, lanes % 32, 8 * (lanes // 32) + j]
            b_op[:, :, j] = B[:, lanes % 32, 8 * (lanes // 32) + j]
        ta, tb = torch.from_numpy(a_op).to(dev), torch.from_numpy(b_op).to(dev)
        out = torch.empty(64, 16, device=dev)
        assert L.mrs_mfma_f16_int_probe(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), ks, torch.cuda.current_stream().cuda_stream) == 0
        o = out.cpu().numpy().astype(np.float64)
        got = np.zeros((32, 32))
        for l in range(64):
            for i in range(16):
                got[(i // 4) * 8 + (l // 32) * 4 + (i % 4), l % 32] = o[l, i]
        return got, np.einsum("srk,sck->rc", A.astype(np.int64), B.astype(np.int64))

    rng = np.random.default_rng(0)
    for amax, ks in ((504, 16), (480, 16), (256, 16), (2048, 1)):
        for _ in range(8):
            A = rng.integers(-amax, amax + 1, (ks, 32, 16)); B = rng.integers(-128, 128, (ks, 32, 16)) if amax != 2048 else rng.integers(0, 64, (ks, 32, 16))
            got, ref = run(A, B)
            assert np.array_equal(got, ref)
    got, ref = run(np.full((16, 32, 16), 504), np.full((16, 32, 16), -128))  # |sum| = 16.5 M, the largest a Q4_K superblock can reach
    assert np.array_equal(got, ref) and ref[0, 0] == -16515072
