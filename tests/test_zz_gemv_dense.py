"""Dense decode GEMV `launch_gemv_{bf16,f16,f32}` (mistralrs-quant/src/gemv/ffi.rs:12-56, kernels/gemv/gemv.cu):
  * CPU: the oracle (f64 dot + error budget) against the reference kernel executed on host fibers (oracle/_ref/libref_gemv.so);
  * CPU: the product kernel on the wave64 host emulation; GPU (`-m gpu`): the product kernel on the MI355X -- same body.
Tolerance: f32 accumulation over K terms in any order (2^-23 * (log2-ish factor) * sum |terms|) + one rounding to T."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.test_oracle_ref import _ref
from tests.util import round_through

P, I = C.c_void_p, C.c_int
EPS_T = {"f32": 2.0 ** -24, "f16": 2.0 ** -11, "bf16": 2.0 ** -8}
CODE = {"f16": 0, "bf16": 1, "f32": 2}
CASES = [(37, 512, 1, True), (37, 512, 8, False), (5, 4096, 3, True), (9, 100, 2, True), (4, 63, 1, False), (130, 24, 5, True)]


def _inputs(dt, M, K, B, seed):
    rng = np.random.default_rng(seed)
    a = round_through((rng.standard_normal((M, K)) * 0.05).astype(np.float32), dt)
    x = round_through(rng.standard_normal((B, K)).astype(np.float32), dt)
    bias = round_through(rng.standard_normal(M).astype(np.float32), dt)
    return a, x, bias


def _tol(K, mag, want, dt):
    return 4 * 2.0 ** -23 * np.sqrt(K) * mag + EPS_T[dt] * 1.001 * np.abs(want) + 1e-30


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("M,K,B,has_bias", CASES)
def test_gemv_oracle_matches_reference_kernel(oracle, dt, M, K, B, has_bias):
    lib = _ref("libref_gemv.so")
    a, x, bias = _inputs(dt, M, K, B, seed=M + K)
    enc = {"f32": lambda v: v, "f16": lambda v: v.astype(np.float16), "bf16": lambda v: oracle.to_bf16_bits(v)}[dt]
    dec = {"f32": lambda v: v, "f16": lambda v: v.astype(np.float32), "bf16": lambda v: oracle.from_bf16_bits(v)}[dt]
    A, X, Bi = (np.ascontiguousarray(enc(v)) for v in (a, x, bias))
    Y = np.ascontiguousarray(enc(np.zeros((B, M), np.float32)))
    vp = lambda v: v.ctypes.data_as(P)
    assert lib.ref_gemv(CODE[dt], vp(A), vp(X), vp(Bi), vp(Y), M, K, B, int(has_bias)) == 0
    want, mag = oracle.gemv_dense(a, x, bias if has_bias else None)
    assert (np.abs(dec(Y).astype(np.float64) - want) <= _tol(K, mag, want, dt)).all()


def check_gemv(oracle, be, dt, M, K, B, has_bias):
    a, x, bias = _inputs(dt, M, K, B, seed=M + K + 1)
    d = None if dt == "f32" else dt
    A, X, Bi = be.buf(a, d), be.buf(x, d), be.buf(bias, d)
    Y = be.buf(np.full((B, M), 7.0, np.float32), d)
    fn = be.sym(f"launch_gemv_{dt}", [P] * 4 + [I] * 3 + [C.c_bool, P])
    fn(A.ptr, X.ptr, Bi.ptr if has_bias else None, Y.ptr, M, K, B, has_bias, be.stream)
    want, mag = oracle.gemv_dense(a, x, bias if has_bias else None)
    assert (np.abs(Y.numpy().astype(np.float64) - want) <= _tol(K, mag, want, dt)).all()


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("M,K,B,has_bias", CASES)
def test_gemv_host_emulation(oracle, dt, M, K, B, has_bias):
    check_gemv(oracle, HostBackend(), dt, M, K, B, has_bias)


def test_gemv_batch_outside_1_8_runs_batch_1(oracle):
    """The reference's dispatch falls back to the batch-1 kernel for any other batch_size (gemv.cu:219-221): only row 0 of Y is written."""
    be = HostBackend()
    a, x, _ = _inputs("f32", 6, 64, 9, seed=3)
    A, X, Y = be.buf(a), be.buf(x), be.buf(np.full((9, 6), 7.0, np.float32))
    be.sym("launch_gemv_f32", [P] * 4 + [I] * 3 + [C.c_bool, P])(A.ptr, X.ptr, None, Y.ptr, 6, 64, 9, False, None)
    want, mag = oracle.gemv_dense(a, x[:1])
    assert (np.abs(Y.numpy()[0] - want[0]) <= _tol(64, mag[0], want[0], "f32")).all()
    assert (Y.numpy()[1:] == 7.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("M,K,B,has_bias", CASES + [(4096, 4096, 1, False), (8192, 4096, 4, True)])
def test_gemv_abi_gpu(oracle, dev, dt, M, K, B, has_bias):
    check_gemv(oracle, GpuBackend(dev), dt, M, K, B, has_bias)
