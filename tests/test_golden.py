"""Frozen golden vectors (tests/golden/*.npz, produced by scripts/gen_golden.py from the reference's own device functions
compiled for the host): they pin
  * the oracle (CPU, always run):  block decode bit-exact, Q8_1 quantizer bit-exact, Q8_1 matvec to f32 term rounding;
  * the HIP kernels (-m gpu):      launch_mmvq_gguf_quantize_q8_1_f32 bit-exact, launch_mmvq_gguf_<t>_f32_plain within the
                                   f32-accumulation bound of the reference value, for all 10 MMVQ formats.
"""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "mmvq_*.npz")))


def test_golden_present():
    assert len(FILES) == 10, "tests/golden is incomplete: run scripts/gen_golden.py in the build container"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    t, n, k = int(g["type"]), int(g["n"]), int(g["k"])
    np.testing.assert_array_equal(oracle.quantize_q8_1(g["x"]), g["y_q8_1"])
    deq = oracle.dequantize(t, g["w"], k)
    np.testing.assert_array_equal(deq, g["dequant_ref"])
    got, mag = oracle.matmul_q8_1_mag(t, g["w"], n, k, g["y_q8_1"])
    tol = 2.0 ** -21 * mag.astype(np.float64) + 2.0 ** -23 * np.abs(g["out_ref"]) + 1e-30
    assert (np.abs(got - g["out_ref"]) <= tol).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_kernels_reproduce_golden(dev, path):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_mmvq
    g = np.load(path)
    t, n, k = int(g["type"]), int(g["n"]), int(g["k"])
    x = torch.from_numpy(g["x"]).to(dev)
    ws, stride = fast_mmvq.quantize_q8_1(x, k, x.shape[0])
    torch.cuda.synchronize()
    got_y = ws[: x.shape[0] * stride * 36].cpu().numpy().reshape(x.shape[0], -1)
    np.testing.assert_array_equal(got_y, g["y_q8_1"])  # activation quantizer: bit-exact
    w = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), g["w"], dev)
    got = fast_mmvq.plain(w, x).cpu().numpy().astype(np.float64)
    # f32 accumulation of ~k/16 terms in an arbitrary order: 8 eps sqrt(terms) * sum|terms|
    tol = 8 * 2.0 ** -23 * np.sqrt(k / 16) * g["mag"].astype(np.float64) + 2.0 ** -23 * np.abs(g["out_ref"]) + 1e-30
    err = np.abs(got - g["out_ref"])
    assert (err <= tol).all(), float((err / tol).max())
