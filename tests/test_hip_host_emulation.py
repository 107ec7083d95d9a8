"""CPU: the PRODUCT kernel sources of libmistralrsquant.so compiled for the host on wave64 fibers (oracle/hip_host/hip/hip_runtime.h,
oracle/build_hip_host.sh -> oracle/_hiphost/libhiphost.so; test infrastructure) and called through the same C-ABI symbols.
  * calibration: launchers that are parity-green on the MI355X (Q8_1 quantizer, plain / fused-GLU / fused-QKV MMVQ, indexed MoE forward)
    give the same answers here -- bit-exact where the GPU test is bit-exact -- so the emulation models DPP / readlane / ballot / barriers
    the way the device executes them;
  * the MoE expert launchers added after the round's GPU minutes were spent run the bodies of tests/test_zz_moe_expert_abi.py here.
  * a curated slice of the EXISTING `-m gpu` test files (MFMA prefill GEMM and flash attention, decode attention, RoPE / RMSNorm / GLU, paged attention)
    is re-run unchanged in a child process with `--host-emulation` (tests/conftest.py), which keeps the matrix-core lane model and the
    lockstep-LDS syncs of the emulation honest in every CPU run.
Nothing in the product path uses this library."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.abi_backends import HostBackend
from tests import test_zz_moe_expert_abi as M

P, I = C.c_void_p, C.c_int
TN = M.TNAMES


@pytest.fixture(scope="module")
def be():
    return HostBackend()


@pytest.mark.parametrize("tname", [t for t in TN if t != "Q8_1"])  # Q8_1 weights: MoE launchers only
def test_calibration_quantizer_and_plain_mmvq(oracle, be, tname):
    t = getattr(oracle, tname)
    n, k, b = 13, 1024, 3
    rng = np.random.default_rng(t)
    w = oracle.random_blocks(t, n, k, seed=t, d_scale=0.02)
    x = rng.standard_normal((b, k)).astype(np.float32)
    kp = oracle.pad512(k)
    y = be.buf(np.zeros((b, kp // 32 * 36), np.uint8))
    xb, wb = be.buf(x), be.buf(w)   # keep the buffers alive across the calls
    be.sym("launch_mmvq_gguf_quantize_q8_1_f32", [P] * 2 + [I] * 3 + [P])(xb.ptr, y.ptr, k, kp, b, None)
    yo = oracle.quantize_q8_1(x)
    np.testing.assert_array_equal(y.numpy(), yo)   # bit-exact, as on the device (tests/test_mmvq.py)
    out = be.buf(np.full((b, n), np.nan, np.float32))
    be.sym(f"launch_mmvq_gguf_{oracle.TYPE_NAMES[t]}_f32_plain", [P] * 3 + [I] * 5 + [P])(wb.ptr, y.ptr, out.ptr, k, n, kp // 32, n, b, None)
    want, mag = oracle.matmul_q8_1_mag(t, w, n, k, yo)
    assert (np.abs(out.numpy().astype(np.float64) - want) <= M._f32_tol(k, mag, want)).all()


@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K", "Q8_0", "Q5_0"])
def test_calibration_fused_glu_and_qkv(oracle, be, tname):
    t = getattr(oracle, tname)
    k, b = 512, 2
    rng = np.random.default_rng(t + 1)
    x = rng.standard_normal((b, k)).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    kp = oracle.pad512(k)
    yb = be.buf(y)
    # fused_glu (silu), f32: act(gate) * up of the two Q8_1 matvecs
    n = 24
    gate, up = oracle.random_blocks(t, n, k, seed=3 + t, d_scale=0.02), oracle.random_blocks(t, n, k, seed=9 + t, d_scale=0.02)
    out, gb, ub = be.buf(np.full((b, n), np.nan, np.float32)), be.buf(gate), be.buf(up)
    be.sym(f"launch_mmvq_gguf_{oracle.TYPE_NAMES[t]}_f32_fused_glu", [P] * 4 + [I] * 6 + [P])(gb.ptr, ub.ptr, yb.ptr, out.ptr, k, n, kp // 32, n, b, 0, None)
    g, gm = oracle.matmul_q8_1_mag(t, gate, n, k, y)
    u, um = oracle.matmul_q8_1_mag(t, up, n, k, y)
    want = oracle.fused_glu(g, u, 0).astype(np.float64)
    a = np.abs(oracle.fused_glu(g, np.ones_like(u), 0)).astype(np.float64)
    tol = a * M._f32_tol(k, um, u) + 1.2 * np.abs(u) * M._f32_tol(k, gm, g) + 2.0 ** -21 * np.abs(want) + 1e-30
    assert (np.abs(out.numpy() - want) <= tol).all()
    # fused_qkv: three matrices, one launch
    nq, nk, nv = 16, 8, 8
    ws = [oracle.random_blocks(t, r, k, seed=20 + i + t, d_scale=0.02) for i, r in enumerate((nq, nk, nv))]
    outs = [be.buf(np.full((b, r), np.nan, np.float32)) for r in (nq, nk, nv)]
    wbs = [be.buf(w) for w in ws]
    be.sym(f"launch_mmvq_gguf_{oracle.TYPE_NAMES[t]}_f32_fused_qkv", [P] * 7 + [I] * 6 + [P])(*(w.ptr for w in wbs), yb.ptr, *(o.ptr for o in outs), k, nq, nk, nv,
                                                                                         kp // 32, b, None)
    for w, o, r in zip(ws, outs, (nq, nk, nv)):
        want, mag = oracle.matmul_q8_1_mag(t, w, r, k, y)
        assert (np.abs(o.numpy().astype(np.float64) - want) <= M._f32_tol(k, mag, want)).all()


@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K", "Q4_0", "Q8_1"])
@pytest.mark.parametrize("input_dim1", [1, 2])
def test_calibration_indexed_moe_forward(oracle, be, tname, input_dim1):
    t = getattr(oracle, tname)
    E, n, k, batch, topk = 5, 12, 512, 3, 2
    rng = np.random.default_rng(4 + t)
    w = M._stack(oracle, t, E, n, k, 50)
    idx = rng.integers(0, E, size=batch * topk).astype(np.uint32)
    rows_in = batch if input_dim1 == 1 else batch * topk
    y = oracle.quantize_q8_1((rng.standard_normal((rows_in, k)) * rng.uniform(0.3, 4.0, (rows_in, 1))).astype(np.float32))
    out = be.buf(np.full((batch * topk, n), np.nan, np.float32))
    wb, yb, ib = be.buf(w), be.buf(y), be.buf(idx)
    be.sym(f"launch_indexed_moe_forward_{M._tag(oracle, t)}_q8_1", [P] * 4 + [I] * 6 + [P])(wb.ptr, yb.ptr, ib.ptr, out.ptr, n, k, batch, topk, oracle.pad512(k),
                                                                                        input_dim1, None)
    got = out.numpy().astype(np.float64)
    for task in range(batch * topk):
        e, row = int(idx[task]), (task // topk if input_dim1 == 1 else task)
        want, mag = oracle.matmul_q8_1_mag(t, w[e * n:(e + 1) * n], n, k, y[row:row + 1])
        assert (np.abs(got[task] - want[0]) <= M._f32_tol(k, mag[0], want[0])).all(), task


# ---- the new launchers
@pytest.mark.parametrize("tname", TN)
@pytest.mark.parametrize("act_type", [0, 1])
def test_emulated_fused_gate_up(oracle, be, tname, act_type):
    M.check_fused_gate_up(oracle, be, tname, act_type, k=512)


@pytest.mark.parametrize("tname", TN)
@pytest.mark.parametrize("topk", [2, 3])
def test_emulated_down_aggregate(oracle, be, tname, topk):
    M.check_down_aggregate(oracle, be, tname, topk, k=512)


@pytest.mark.parametrize("E,tokens,topk", [(8, 37, 2), (4, 1, 2), (64, 300, 6), (3, 5, 1)])
def test_emulated_dispatch(oracle, be, E, tokens, topk):
    M.check_dispatch(oracle, be, E, tokens, topk)


@pytest.mark.parametrize("sym,in_dt,out_dt", M.REDUCE_CASES)
def test_emulated_weighted_reduce(oracle, be, sym, in_dt, out_dt):
    M.check_weighted_reduce(oracle, be, sym, in_dt, out_dt)


@pytest.mark.parametrize("tname", TN)
@pytest.mark.parametrize("mode", ["gate_up", "down_weighted", "down_plain"])
def test_emulated_grouped_gemm(oracle, be, tname, mode):
    M.check_grouped_gemm(oracle, be, tname, mode, k=512, tokens=40)


@pytest.mark.parametrize("sym,dt", [("launch_quantize_q8_1", "f32"), ("launch_quantize_q8_1_f16", "f16"), ("launch_quantize_q8_1_bf16", "bf16")])
def test_emulated_moe_quantize_q8_1(oracle, be, sym, dt):
    M.check_quantize_q8_1(oracle, be, sym, dt)


GPU_SUITE_SLICES = [
    ("tests/test_gemm.py", "130-384-512-Q4_K or 1-128-256-Q6_K or (large_m and 256-128-512-False-Q8_0) or slab_producers"),   # MFMA GEMM, both kernels
    ("tests/test_paged_attn.py", "(prefill_attention and 31-0-8-8) or (prefill_attention and 64-45-16-2) or (decode_attention and 8-4-64) "
                                 "or (llama_shape and v2 and bf16) or reshape_and_cache_f32_into_bf16"),
    ("tests/test_glue_ops.py", ""),
]


@pytest.mark.parametrize("path,expr", GPU_SUITE_SLICES, ids=[p.split("/")[-1] for p, _ in GPU_SUITE_SLICES])
def test_gpu_suite_slice_on_host_emulation(path, expr):
    """The unmodified GPU tests, CPU tensors, emulated kernels.  A child process: --host-emulation repoints the package's loader and patches torch."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", path, "-m", "gpu", "--host-emulation", "-q", "-x", "-p", "no:cacheprovider"] + (["-k", expr] if expr else [])
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
