"""MMQ C ABI (`launch_mmq_quantize_q8_1_{D4,DS4,D2S6}`, `launch_mmq_quantize_glu_q8_1_*`, `launch_mmq_gguf_<t>`, `launch_mmq_gguf_<t>_moe`;
reference: mistralrs-quant/src/gguf/ffi.rs:1313-1452, fast_mmq.rs:388-447, kernels/mmq_gguf/):
  * CPU: the oracle's block_q8_1_mmq quantizer BIT-EXACT against the reference's own kernel executed on host fibers
    (oracle/_ref/libref_mmq_quantize.so, f32 / f16 / bf16 inputs, all three layouts, row gather, zero blocks, ragged K);
  * CPU: the product kernels (mistral.rs_amd/csrc/mmq.hip) on the wave64 host emulation; GPU (`-m gpu`): the same bodies on the MI355X.
The quantizer is held to bit-exact.  The matmul is integer dots combined in f32 in a different order than the reference's MMA tiles (no order
is "the" reference order: its own stream-k fix-up changes it with the SM count), so it is held to the f32-accumulation budget
4 * 2^-23 * sqrt(K/16) * SUM|terms| + one rounding to the output dtype against the oracle's f64 combination of the same integer dots --
the same budget the MMVQ tests use (tests/test_mmvq.py)."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.test_oracle_ref import _ref
from tests.util import round_through

P, I, L = C.c_void_p, C.c_int, C.c_int64
LAYOUTS = {0: "D4", 1: "DS4", 2: "D2S6"}
CODE = {"f32": 0, "f16": 1, "bf16": 30}
EPS_T = {"f32": 2.0 ** -24, "f16": 2.0 ** -11, "bf16": 2.0 ** -8}


def _x(rows, k, dt, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, k)) * rng.uniform(0.05, 3.0, (rows, 1))).astype(np.float32)
    if k >= 64:
        x[0, 32:64] = 0.0  # an all-zero scale group (the reference reaches 0 * inf there)
    return round_through(x, dt)


def _enc(oracle, v, dt):
    return np.ascontiguousarray({"f32": lambda a: a, "f16": lambda a: a.astype(np.float16), "bf16": oracle.to_bf16_bits}[dt](v))


# ------------------------------------------------------------------------------------------------ oracle <-> reference kernel
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("rows,k", [(3, 512), (5, 1000), (2, 96), (1, 4096), (4, 36)])
def test_mmq_quantize_oracle_matches_reference_kernel(oracle, layout, dt, rows, k):
    lib = _ref("libref_mmq_quantize.so")
    lib.ref_mmq_quantize.argtypes = [I, I, P, P, P, L, L, L, L]
    x = _x(rows, k, dt, seed=rows * 7 + k + layout)
    ids = np.random.default_rng(k).permutation(rows).astype(np.int32) if rows > 2 else None
    want = oracle.quantize_q8_1_mmq(x, layout, ids=ids)
    X = _enc(oracle, x, dt)
    got = np.zeros_like(want)
    assert lib.ref_mmq_quantize(CODE[dt], layout, X.ctypes.data, None if ids is None else ids.ctypes.data, got.ctypes.data, k, k, oracle.pad512(k),
                                rows) == 0
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------------ product kernels
def check_quantize(oracle, be, layout, dt, rows, k, gather):
    x = _x(rows, k, dt, seed=rows * 11 + k + layout)
    ids = np.random.default_rng(k + 1).integers(0, rows, rows + 2).astype(np.int32) if gather else None
    want = oracle.quantize_q8_1_mmq(x, layout, ids=ids)
    d = None if dt == "f32" else dt
    X, Y = be.buf(x, d), be.buf(np.zeros_like(want))
    IDS = be.buf(ids) if gather else None
    fn = be.sym(f"launch_mmq_quantize_q8_1_{LAYOUTS[layout]}", [P, P, P, I] + [L] * 8 + [P])
    fn(X.ptr, IDS.ptr if gather else None, Y.ptr, CODE[dt], k, k, 0, 0, oracle.pad512(k), want.shape[1], 1, 1, be.stream)
    assert np.array_equal(Y.numpy(), want)


def _dequant_mmq(oracle, y, layout):
    """block_q8_1_mmq [kb, n, 144] -> f32 [n, kb * 128] (d * q)."""
    kb, n, _ = y.shape
    q = y[:, :, 16:].view(np.int8).astype(np.float32).reshape(kb, n, 128)
    if layout == 0:
        d = np.repeat(y[:, :, :16].copy().view(np.float32).reshape(kb, n, 4), 32, axis=2)
    elif layout == 1:
        d = np.repeat(y[:, :, :16].copy().view(np.float16).reshape(kb, n, 8)[:, :, 0::2].astype(np.float32), 32, axis=2)
    else:
        d = np.repeat(y[:, :, :4].copy().view(np.float16).reshape(kb, n, 2).astype(np.float32), 64, axis=2)
    return (d * q).transpose(1, 0, 2).reshape(n, kb * 128), d.transpose(1, 0, 2).reshape(n, kb * 128)


def check_quantize_glu(oracle, be, layout, dt, act, f32_entry):
    """activation(gate) * up formed in the input dtype, then the same quantizer.  libm's expf / tanhf differ by an ulp between host and device,
    so the check is on the decoded values: |d * q - product| <= 0.5 d (+ the half rounding of d for the 16-bit headers)."""
    rows, k = 3, 640
    rng = np.random.default_rng(act + layout)
    g = round_through(rng.standard_normal((rows, k)).astype(np.float32) * 2, dt)
    u = round_through(rng.standard_normal((rows, k)).astype(np.float32), dt)
    a = round_through(oracle.fused_glu(g, np.ones_like(g), act), dt) if dt != "f32" else oracle.fused_glu(g, np.ones_like(g), act)
    prod = round_through(a * u, dt)
    kp = oracle.pad512(k)
    d = None if dt == "f32" else dt
    G, U, Y = be.buf(g, d), be.buf(u, d), be.buf(np.zeros((kp // 128, rows, 144), np.uint8))
    if f32_entry:
        be.sym(f"launch_mmq_quantize_glu_q8_1_{LAYOUTS[layout]}_f32", [P, P, P, P] + [L] * 4 + [I, P])(G.ptr, U.ptr, None, Y.ptr, k, k, kp, rows, act, be.stream)
    else:
        be.sym(f"launch_mmq_quantize_glu_q8_1_{LAYOUTS[layout]}", [P, P, P, P, I] + [L] * 4 + [I, P])(G.ptr, U.ptr, None, Y.ptr, CODE[dt], k, k, kp, rows, act,
                                                                                                   be.stream)
    got, dd = _dequant_mmq(oracle, Y.numpy(), layout)
    full = np.zeros((rows, kp), np.float32)
    full[:, :k] = prod
    slack = 2e-3 * np.abs(full) + 1e-6  # libm ulp + the half-rounded d
    assert (np.abs(got - full) <= 0.5 * dd * 1.002 + slack).all()


def _mmq_inputs(oracle, t, n, k, cols, seed):
    w = oracle.random_blocks(t, n, k, seed=seed)
    x = _x(cols, k, "f32", seed + 1)
    y = oracle.quantize_q8_1_mmq(x, oracle.mmq_layout(t))
    return w, y


def _tol(k, mag, want, dt):
    return 4 * 2.0 ** -23 * np.sqrt(k / 16) * mag + EPS_T[dt] * 1.001 * np.abs(want) + 1e-30


def check_mmq(oracle, be, t, dt, n, k, cols):
    w, y = _mmq_inputs(oracle, t, n, k, cols, seed=t * 13 + n + cols)
    want, mag = oracle.matmul_q8_1_mmq(t, w, n, k, y)
    W, Y = be.buf(w), be.buf(y)
    D = be.buf(np.full((cols, n), 7.0, np.float32), None if dt == "f32" else dt)
    fn = be.sym(f"launch_mmq_gguf_{oracle.TYPE_NAMES[t]}", [P, P, P, P] + [L] * 5 + [I, I, L, I, I, P])
    fn(None, W.ptr, Y.ptr, D.ptr, k, n, cols, k // oracle.block_size(t), n, 0, 256, 160 << 10, 64, CODE[dt], be.stream)
    got = D.numpy().astype(np.float64)
    assert (np.abs(got - want) <= _tol(k, mag, want, dt)).all(), float((np.abs(got - want) / (mag + 1e-30)).max())


def check_mmq_moe(oracle, be, t, n, k, experts, counts):
    """Routes in expert-sorted order: expert e owns y columns [bounds[e], bounds[e+1]), results scatter to dst column ids_dst[j]."""
    routes = int(sum(counts))
    rng = np.random.default_rng(t + routes)
    w = oracle.random_blocks(t, experts * n, k, seed=t + 5)
    x = _x(routes, k, "f32", t + 9)
    y = oracle.quantize_q8_1_mmq(x, oracle.mmq_layout(t))
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ids_dst = rng.permutation(routes).astype(np.int32)
    rb = oracle.row_bytes(t, k)
    want = np.zeros((routes, n), np.float64)
    mag = np.zeros((routes, n), np.float64)
    for e in range(experts):
        if counts[e] == 0:
            continue
        sl = slice(bounds[e], bounds[e + 1])
        o, m = oracle.matmul_q8_1_mmq(t, w.reshape(-1)[e * n * rb:(e + 1) * n * rb], n, k, np.ascontiguousarray(y[:, sl]))
        want[ids_dst[sl]], mag[ids_dst[sl]] = o, m
    W, Y, B, IDS = be.buf(w), be.buf(y), be.buf(bounds), be.buf(ids_dst)
    D = be.buf(np.full((routes, n), 7.0, np.float32))
    fn = be.sym(f"launch_mmq_gguf_{oracle.TYPE_NAMES[t]}_moe", [P] * 6 + [L] * 7 + [I, I, L, I, P])
    fn(None, W.ptr, Y.ptr, IDS.ptr, B.ptr, D.ptr, k, n, routes, k // oracle.block_size(t), n, experts, int(max(counts)), 0, 256, 160 << 10, 64, be.stream)
    got = D.numpy().astype(np.float64)
    assert (np.abs(got - want) <= _tol(k, mag, want, "f32")).all()


def test_mmq_oracle_close_to_exact_matmul(oracle):
    """Sanity of the MMQ restatement itself: against the f64 matmul of the dequantized weights with the dequantized activations the only
    differences are the stored (half-rounded, pre-quantization) partial sums."""
    for t in oracle.MMVQ_TYPES:
        n, k, cols = 6, 1024, 3
        w, y = _mmq_inputs(oracle, t, n, k, cols, seed=t)
        out, mag = oracle.matmul_q8_1_mmq(t, w, n, k, y)
        xa, _ = _dequant_mmq(oracle, y, oracle.mmq_layout(t))
        ex = oracle.matmul_exact(t, w, n, k, xa[:, :k])
        assert (np.abs(out - ex) <= 2e-2 * mag + 1e-6).all(), oracle.TYPE_NAMES[t]


HOST_Q = [(0, "f32", 3, 512, False), (1, "f16", 5, 1000, True), (2, "bf16", 2, 96, False), (1, "f32", 4, 36, True), (2, "f32", 1, 2048, False),
          (0, "bf16", 2, 700, True)]


@pytest.mark.parametrize("layout,dt,rows,k,gather", HOST_Q)
def test_mmq_quantize_host_emulation(oracle, layout, dt, rows, k, gather):
    check_quantize(oracle, HostBackend(), layout, dt, rows, k, gather)


@pytest.mark.parametrize("layout,dt,act,f32_entry", [(1, "f32", 0, True), (0, "f16", 1, False), (2, "bf16", 2, False), (1, "bf16", 0, False), (0, "f32", 4, False)])
def test_mmq_quantize_glu_host_emulation(oracle, layout, dt, act, f32_entry):
    check_quantize_glu(oracle, HostBackend(), layout, dt, act, f32_entry)


@pytest.mark.parametrize("tname", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"])
def test_mmq_host_emulation(oracle, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq(oracle, HostBackend(), t, "f32", 7, 512, 11)
    check_mmq(oracle, HostBackend(), t, "bf16", 5, 256, 3)


def check_mmq_row_paths(oracle, be, t, n, k, cols):
    """Launches with >= 512 (16-row x 8-column) tiles take the 4-rows-per-wave kernel, smaller ones the 1-row kernel (csrc/mmq.hip launch_mmq_t);
    the per-output arithmetic does not depend on the path: the big launch (ragged against 16 rows) is within budget of the oracle AND its
    first rows are bit-identical to a small launch over the same weights."""
    assert ((n + 15) // 16) * ((cols + 7) // 8) >= 512 and n % 16
    w, y = _mmq_inputs(oracle, t, n, k, cols, seed=t + 1)
    want, mag = oracle.matmul_q8_1_mmq(t, w, n, k, y)
    W, Y = be.buf(w), be.buf(y)
    D, D7 = be.buf(np.full((cols, n), 7.0, np.float32)), be.buf(np.full((cols, 7), 7.0, np.float32))
    fn = be.sym(f"launch_mmq_gguf_{oracle.TYPE_NAMES[t]}", [P, P, P, P] + [L] * 5 + [I, I, L, I, I, P])
    fn(None, W.ptr, Y.ptr, D.ptr, k, n, cols, k // oracle.block_size(t), n, 0, 256, 160 << 10, 64, 0, be.stream)
    fn(None, W.ptr, Y.ptr, D7.ptr, k, 7, cols, k // oracle.block_size(t), 7, 0, 256, 160 << 10, 64, 0, be.stream)
    got = D.numpy().astype(np.float64)
    assert (np.abs(got - want) <= _tol(k, mag, want, "f32")).all()
    if cols >= 48 and k % 256 == 0:
        return  # prompt-sized launches of the DS4 K-quants run on the matrix cores (mmq_mfma_kernel): same integers, another f32 summation order
    assert np.array_equal(D.numpy()[:, :7], D7.numpy())


@pytest.mark.parametrize("tname", ["q4_k", "q6_k"])
def test_mmq_row_paths_host_emulation(oracle, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_row_paths(oracle, HostBackend(), t, 1030, 256, 64)


@pytest.mark.parametrize("tname", ["q4_k", "q6_k", "q8_0", "q2_k"])
def test_mmq_moe_host_emulation(oracle, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_moe(oracle, HostBackend(), t, 6, 256, 4, [3, 0, 9, 1])


MFMA_CASES = [("q4_k", "f32", 200, 512, 70), ("q5_k", "bf16", 33, 256, 130), ("q4_k", "f16", 129, 768, 48), ("q6_k", "f32", 200, 512, 70), ("q6_k", "bf16", 40, 256, 129),
              ("q8_0", "f32", 200, 384, 70), ("q4_0", "f16", 45, 128, 130), ("q4_1", "f32", 33, 640, 48), ("q5_0", "f32", 100, 256, 64), ("q5_1", "bf16", 64, 384, 50),
              ("q3_k", "f32", 200, 512, 70), ("q3_k", "f16", 33, 256, 129), ("q2_k", "f32", 200, 512, 70), ("q2_k", "bf16", 40, 768, 49)]


@pytest.mark.parametrize("tname,dt,n,k,cols", MFMA_CASES)
def test_mmq_matrix_core_route_host_emulation(oracle, tname, dt, n, k, cols):
    """launch_mmq_gguf_<t> with >= 48 columns and >= 32 rows: one v_mfma_i32_32x32x32_i8 per 32-value sub-block (csrc/mmq.hip mmq_mfma_*_kernel);
    ragged row / column tiles in both tile sizes (64 x 64 for small launches, 128 x 128 otherwise); the same budget against the f64 oracle as the
    v_dot4 kernel."""
    t = {v: k_ for k_, v in oracle.TYPE_NAMES.items()}[tname]
    be = HostBackend()
    policy = be.sym("mrs_mmq_set_small_tiles_below", [I], None)
    try:
        for below in (1 << 30, 0):
            policy(below)
            check_mmq(oracle, be, t, dt, n, k, cols)
    finally:
        policy(384)


@pytest.mark.parametrize("tname", ["q4_k", "q5_k", "q6_k", "q8_0", "q5_0", "q4_1", "q3_k", "q2_k"])
def test_mmq_matrix_core_route_moe_host_emulation(oracle, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_moe(oracle, HostBackend(), t, 40, 256, 3, [50, 0, 70])


# ------------------------------------------------------------------------------------------------ MI355X
@pytest.mark.gpu
@pytest.mark.parametrize("tname,dt,n,k,cols", MFMA_CASES + [("q4_k", "f32", 1030, 4096, 512), ("q5_k", "f32", 4100, 1024, 257), ("q6_k", "f32", 1030, 4096, 512), ("q8_0", "f32", 1030, 4096, 512), ("q4_0", "f32", 515, 2048, 300), ("q3_k", "f32", 1030, 4096, 512), ("q2_k", "f32", 515, 2048, 300)])
def test_mmq_matrix_core_route_gpu(oracle, dev, tname, dt, n, k, cols):
    t = {v: k_ for k_, v in oracle.TYPE_NAMES.items()}[tname]
    be = GpuBackend(dev)
    policy = be.sym("mrs_mmq_set_small_tiles_below", [I], None)
    try:
        for below in (1 << 30, 0):
            policy(below)
            check_mmq(oracle, be, t, dt, n, k, cols)
    finally:
        policy(384)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["q4_k", "q5_k", "q6_k", "q8_0", "q5_0", "q4_1", "q3_k", "q2_k"])
def test_mmq_matrix_core_route_moe_gpu(oracle, dev, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_moe(oracle, GpuBackend(dev), t, 40, 256, 3, [50, 0, 70])
    check_mmq_moe(oracle, GpuBackend(dev), t, 200, 1024, 8, [64, 100, 0, 2, 130, 8, 1, 300])


@pytest.mark.gpu
@pytest.mark.parametrize("layout,dt,rows,k,gather", HOST_Q + [(1, "bf16", 512, 4096, False), (0, "f16", 64, 14336, True)])
def test_mmq_quantize_abi_gpu(oracle, dev, layout, dt, rows, k, gather):
    check_quantize(oracle, GpuBackend(dev), layout, dt, rows, k, gather)


@pytest.mark.gpu
@pytest.mark.parametrize("layout,dt,act,f32_entry", [(1, "f32", 0, True), (0, "f16", 1, False), (2, "bf16", 2, False), (1, "bf16", 0, False), (0, "f32", 3, False)])
def test_mmq_quantize_glu_abi_gpu(oracle, dev, layout, dt, act, f32_entry):
    check_quantize_glu(oracle, GpuBackend(dev), layout, dt, act, f32_entry)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("tname", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"])
def test_mmq_abi_gpu(oracle, dev, tname, dt):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    be = GpuBackend(dev)
    check_mmq(oracle, be, t, dt, 7, 512, 11)
    check_mmq(oracle, be, t, dt, 130, 4096, 33)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k"])
def test_mmq_row_paths_abi_gpu(oracle, dev, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_row_paths(oracle, GpuBackend(dev), t, 1030, 512, 64)
    check_mmq_row_paths(oracle, GpuBackend(dev), t, 4100, 1024, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["q4_k", "q5_k", "q6_k", "q8_0", "q2_k", "q4_0"])
def test_mmq_moe_abi_gpu(oracle, dev, tname):
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    check_mmq_moe(oracle, GpuBackend(dev), t, 6, 256, 4, [3, 0, 9, 1])
    check_mmq_moe(oracle, GpuBackend(dev), t, 64, 1024, 8, [5, 17, 0, 2, 8, 8, 1, 30])
