"""Prompt GEMM in the reference CPU path's arithmetic on the matrix cores (csrc/ext_gemm_qi.hip): Q8_K activation rows (engine-order RmsNorm + candle's
quantizer) x Q4_K / Q6_K weights through v_mfma_f32_32x32x16_f16 on exact small integers, combined in the decode engine's f32 order (ORD-U).
Every output row equals the decode engine's GEMV of that token (oracle/cpu_path_oracle.c orc_gemv_engine) BIT FOR BIT -- so does the GPU result against the
batch-1 decode kernel itself.  Also: the matrix core's f32 accumulator is exact on the integer operands the kernel feeds it (adversarial magnitudes)."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend

VP, CI = C.c_void_p, C.c_int


def run_gemm(O, be, tname, n, k, T, norm, glu=False, acc=False, seed=0):
    t = getattr(O, tname)
    rng = np.random.default_rng(seed)
    packed = O.quantize(t, (rng.standard_normal((n, k)) * 0.05).astype(np.float32)).reshape(n, -1)
    nb = be.sym("mrs_gemm_qi_repack_bytes", [CI, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, k)
    assert nb > 0
    src, wq = be.buf(np.ascontiguousarray(packed).reshape(-1)), be.buf(np.full(nb, 0xA5, np.uint8))
    assert be.sym("mrs_gemm_qi_repack", [VP, CI, C.c_longlong, C.c_longlong, VP, VP], CI)(src.ptr, t, n, k, wq.ptr, be.stream) == 0
    x = rng.standard_normal((T, k)).astype(np.float32)
    x2 = rng.standard_normal((T, k)).astype(np.float32) if glu else None
    if T > 2:
        x[1, : min(256, k)] = 0.0  # an all-zero activation block
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32) if norm else None
    ab = be.sym("mrs_qi_act_bytes", [CI, CI], C.c_size_t)(T, k)
    act = be.buf(np.zeros(ab, np.uint8))
    xb = be.buf(x)
    x2b = be.buf(x2) if glu else None
    nwb = be.buf(nw) if norm else None
    tmp = be.buf(np.zeros((T, k), np.float32)) if glu else None
    rc = be.sym("mrs_qi_quantize", [VP, VP, CI, VP, C.c_float, CI, CI, VP, VP, VP], CI)(xb.ptr, x2b.ptr if glu else None, k, nwb.ptr if norm else None, 1e-5, T, k, act.ptr,
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


CASES = [("Q4_K", 40, 512, 5, 0), ("Q4_K", 36, 1280, 4, 1), ("Q4_K", 130, 1024, 33, 1), ("Q4_K", 64, 4096, 7, 1), ("Q4_K", 33, 2816, 3, 0), ("Q6_K", 40, 512, 5, 0), ("Q6_K", 70, 1024, 130, 1),
         ("Q6_K", 32, 3584, 4, 0)]


@pytest.mark.parametrize("tname,n,k,T,norm", CASES)
def test_gemm_qi_host_emulation(oracle, tname, n, k, T, norm):
    run_gemm(oracle, HostBackend(), tname, n, k, T, norm, seed=n + k + T)


def test_gemm_qi_glu_and_accumulate_host_emulation(oracle):
    run_gemm(oracle, HostBackend(), "Q4_K", 48, 768, 6, 0, glu=True, seed=1)
    run_gemm(oracle, HostBackend(), "Q6_K", 36, 512, 9, 0, acc=True, seed=2)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,T,norm", CASES + [("Q4_K", 4096, 4096, 512, 1), ("Q4_K", 1024, 14336, 300, 0), ("Q6_K", 1024, 4096, 512, 1), ("Q6_K", 512, 14336, 129, 0),
                                                     ("Q4_K", 28672, 4096, 64, 1)])
def test_gemm_qi_gpu(oracle, dev, tname, n, k, T, norm):
    run_gemm(oracle, GpuBackend(dev), tname, n, k, T, norm, seed=n + k + T)


@pytest.mark.gpu
def test_gemm_qi_glu_and_accumulate_gpu(oracle, dev):
    run_gemm(oracle, GpuBackend(dev), "Q4_K", 512, 14336, 70, 0, glu=True, seed=1)
    run_gemm(oracle, GpuBackend(dev), "Q6_K", 4096, 4096, 200, 0, acc=True, seed=2)


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
