"""CPU: pin the oracle against the REFERENCE'S OWN arithmetic (oracle/_ref, built by oracle/build_ref.sh from the
device functions of /root/reference/mistralrs-quant/kernels -- compiled for the host, never copied):

  * block decode of all 10 MMVQ formats == kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu
    get_quant / get_affine_params (w = scale*q - offset), bit for bit;
  * the Q8_1 matvec oracle == SUM of the reference's vec_dot_<type>_q8_1 terms (mmvq_gguf.cu:388-685), to f32
    rounding of the individual terms (the reference adds f32 terms; both sides combine them in f64);
  * GLU activations == apply_glu_activation (mmvq_gguf.cu:44-85).
These libraries travel to the GPU box as prebuilt files; when absent (fresh clone without /root/reference) the tests skip.
"""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def _ref(name):
    p = os.path.join(REF_DIR, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not built (needs /root/reference: make -C oracle ref)")
    return C.CDLL(p)


TYPES = [2, 3, 6, 7, 8, 10, 11, 12, 13, 14]


@pytest.mark.parametrize("t", TYPES)
def test_block_decode_matches_reference_format_spec(oracle, t):
    lib = _ref("libref_affine.so")
    n, k = 5, 768
    blocks = oracle.random_blocks(t, n, k, seed=100 + t, d_scale=0.02)
    want = np.empty((n, k), dtype=np.float32)
    assert lib.ref_dequantize(t, blocks.ctypes.data_as(C.c_void_p), n, k, want.ctypes.data_as(C.c_void_p)) == 0
    got = oracle.dequantize(t, blocks, k)
    # reference computes scale*q - offset in f32; the oracle's decode must agree to the last bit (only the sign of an
    # exact zero may differ: the affine form yields +0 where d*0 yields -0 for a negative scale)
    np.testing.assert_array_equal(got, want)
    nz = want != 0
    np.testing.assert_array_equal(got[nz].view(np.uint32), want[nz].view(np.uint32))


@pytest.mark.parametrize("t", TYPES)
def test_q8_1_matvec_matches_reference_vec_dot(oracle, t):
    lib = _ref("libref_mmvq.so")
    rng = np.random.default_rng(t)
    n, k, b = 7, 1024, 3
    w = oracle.random_blocks(t, n, k, seed=200 + t, d_scale=0.02)
    x = (rng.standard_normal((b, k)) * np.array([[0.5], [2.0], [30.0]])).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    want = np.empty((b, n), dtype=np.float64)
    assert lib.ref_mmvq(t, w.ctypes.data_as(C.c_void_p), n, k, y.ctypes.data_as(C.c_void_p), y.shape[1] // 36, b,
                        want.ctypes.data_as(C.c_void_p)) == 0
    got, mag = oracle.matmul_q8_1_mag(t, w, n, k, y)
    # each reference term is an f32 expression of a few products: relative 2^-22 of the term magnitudes (+ f32 output rounding)
    tol = 2.0 ** -21 * mag.astype(np.float64) + 2.0 ** -23 * np.abs(want) + 1e-30
    assert (np.abs(got - want) <= tol).all(), float(np.max(np.abs(got - want) / tol))


def test_glu_activations_match_reference(oracle):
    lib = _ref("libref_mmvq.so")
    lib.ref_glu_activation.restype = C.c_float
    lib.ref_glu_activation.argtypes = [C.c_float, C.c_int]
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-12, 12, 481), [0.0, -0.0, 1e-6, 30.0, -30.0]]).astype(np.float32)
    for act in range(5):
        for v in xs:
            a, r = L.orc_glu_act(float(v), act), lib.ref_glu_activation(float(v), act)
            assert abs(a - r) <= 2e-6 * max(1.0, abs(r)), (act, float(v), a, r)
