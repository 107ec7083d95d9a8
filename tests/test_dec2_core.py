"""Round-4 GEMV core (csrc/dec_core2.cuh): one lane = one superblock, records of up to 64 superblocks, ORD-U combination.
The plain launcher (mrs_dec2_gemv) against the engine-order restatement (oracle/cpu_path_oracle.c orc_gemv_engine) BIT FOR BIT and against the
generic ggml order (orc_matmul_cpu) to f32-order tolerance, over every geometry class of dec2::geo_for (rows per record 16 / 8 / 4 / 2 / 1, one and two
tile steps per chunk, chunks with dead lanes, ragged last record).  Same bodies on the wave64 host emulation (CPU suite) and on the MI355X."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend


class Mat(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]


def repack2(be, t, packed, n, k):
    nbytes = be.sym("mrs_dec2_repack_bytes", [C.c_int, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, k)
    assert nbytes > 0
    src = be.buf(np.ascontiguousarray(packed).reshape(-1))
    dst = be.buf(np.full(nbytes, 0xA5, dtype=np.uint8))  # the repack must write every byte a kernel can read
    assert be.sym("mrs_dec2_repack", [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p], C.c_int)(src.ptr, t, n, k, dst.ptr, be.stream) == 0
    return dst, Mat(dst.ptr, t, n, k)


GEMV = [C.POINTER(Mat), C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]


def check(O, be, tname, n, k, b, norm, seed=0):
    t = getattr(O, tname)
    rng = np.random.default_rng(seed)
    packed = O.quantize(t, (rng.standard_normal((n, k)) * 0.05).astype(np.float32)).reshape(n, -1)
    keep, m = repack2(be, t, packed, n, k)
    x = rng.standard_normal((b, k)).astype(np.float32)
    if seed % 2:
        x[0, : min(256, k)] = 0.0  # an all-zero activation block now and then
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32) if norm else None
    xb, ob = be.buf(x), be.buf(np.full((b, n), np.nan, dtype=np.float32))
    nb = be.buf(nw) if norm else None
    rc = be.sym("mrs_dec2_gemv", GEMV, C.c_int)(C.byref(m), xb.ptr, k, nb.ptr if norm else None, 1e-5, ob.ptr, n, b, be.stream)
    assert rc == 0, rc
    got = ob.numpy()
    xin = O.rms_norm_engine(x, nw, 1e-5) if norm else x
    eng = np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in xin], axis=0)
    assert np.array_equal(got, eng), (tname, n, k, b, "engine-order oracle", float(np.abs(got - eng).max()), int((got != eng).sum()))
    want = O.matmul_cpu(t, packed, n, k, xin)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


# (type, rows, K, columns, norm): K / 256 = 1, 2, 3 (LPC 1), 5 (LPC 2), 11 (LPC 4, ragged chunk), 16 (LPC 4), 20 (LPC 8, dead lanes), 32 (LPC 8), 56 (LPC 16, 14 of 16),
# 64 (LPC 16), 112 (two tile steps)
CASES = [("Q4_K", 37, 256, 1, 0), ("Q4_K", 70, 512, 1, 1), ("Q4_K", 19, 768, 2, 0), ("Q4_K", 21, 1280, 1, 0), ("Q4_K", 13, 2816, 1, 1), ("Q4_K", 18, 4096, 1, 1),
         ("Q4_K", 11, 5120, 2, 0), ("Q4_K", 7, 8192, 1, 0), ("Q4_K", 5, 14336, 1, 0), ("Q4_K", 3, 16384, 1, 0), ("Q4_K", 3, 28672, 1, 0),
         ("Q6_K", 50, 512, 1, 0), ("Q6_K", 22, 4096, 1, 1), ("Q6_K", 6, 14336, 2, 0), ("Q6_K", 3, 28672, 1, 0),
         ("Q5_K", 40, 512, 1, 0), ("Q5_K", 9, 4096, 2, 1), ("Q5_K", 4, 14336, 1, 0),
         ("Q8_0", 40, 512, 1, 0), ("Q8_0", 10, 4096, 1, 1), ("Q8_0", 4, 14336, 2, 0)]


@pytest.mark.parametrize("tname,n,k,b,norm", CASES)
def test_gemv_host_emulation(oracle, tname, n, k, b, norm):
    check(oracle, HostBackend(), tname, n, k, b, norm, seed=n + k)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b,norm", CASES + [("Q4_K", 4096, 4096, 1, 1), ("Q4_K", 28672, 4096, 1, 1), ("Q4_K", 4096, 14336, 1, 0), ("Q6_K", 4096, 14336, 1, 0),
                                                     ("Q6_K", 32064, 4096, 1, 1), ("Q8_0", 4096, 4096, 2, 1), ("Q5_K", 2048, 4096, 1, 0), ("Q4_K", 1024, 8192, 4, 1)])
def test_gemv_gpu(oracle, dev, tname, n, k, b, norm):
    check(oracle, GpuBackend(dev), tname, n, k, b, norm, seed=n + k)


def test_ring2_variant_host_emulation():
    """The 2-tile ring (dec_core2.cuh stream() RING2: chosen at launch when a workgroup has >= 8 units, i.e. never at the row counts a CPU test can afford) forced through
    MRS_DEC_RING2=1 in a child process (the launcher reads the variable once per process): a slice of the cases above, bit for bit."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MRS_DEC_RING2="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_dec2_core.py"), "-m", "not gpu",
                        "-k", "test_gemv_host_emulation and (4096 or 14336 or 512)"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_ring_variants_gpu(oracle, dev):
    """Both ring depths on the device at shapes on either side of the launcher's rule (8 units per workgroup): gate / up sized (ring 2 by rule), down sized (ring 4)."""
    check(oracle, GpuBackend(dev), "Q4_K", 14336, 4096, 1, 1, seed=5)   # 3584 units: 14 per workgroup -> ring 2
    check(oracle, GpuBackend(dev), "Q4_K", 4096, 14336, 1, 0, seed=6)   # 1024 units: 4 per workgroup -> ring 4
    check(oracle, GpuBackend(dev), "Q6_K", 16384, 4096, 2, 1, seed=7)   # 16 per workgroup, two columns
