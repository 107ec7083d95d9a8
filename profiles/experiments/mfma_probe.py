#!/usr/bin/env python
"""Is v_mfma_f32_32x32x16_f16 exact on small-integer operands (all partial sums < 2^24)?  Random and adversarial cases vs int64 numpy."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import mistralrs_amd  # noqa: F401
from mistralrs_amd import _lib

L = _lib.load("ext")
L.mrs_mfma_f16_int_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def run(A, B):  # A, B: [ksteps][32][16] integer arrays (rows x k)
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
    ref = np.einsum("srk,sck->rc", A.astype(np.int64), B.astype(np.int64))
    got = np.zeros((32, 32))
    for l in range(64):
        for i in range(16):
            got[(i // 4) * 8 + (l // 32) * 4 + (i % 4), l % 32] = o[l, i]
    return got, ref


rng = np.random.default_rng(0)
res = {}
for name, amax, bmax, ks in (("q4k_like", 504, 128, 16), ("q6k_lo", 480, 128, 16), ("small", 15, 127, 16), ("long", 504, 128, 64)):
    bad = 0; worst = 0
    for t in range(20):
        A = rng.integers(-amax, amax + 1, (ks, 32, 16)); B = rng.integers(-bmax, bmax, (ks, 32, 16))
        got, ref = run(A, B)
        bad += int((got != ref).sum()); worst = max(worst, int(np.abs(ref).max()))
    res[name] = {"mismatches": bad, "max_abs_ref": worst}
# adversarial: everything at the maximum with one sign: |sum| = 256 * 504 * 128 = 16.5 M < 2^24
A = np.full((16, 32, 16), 504); B = np.full((16, 32, 16), -128)
got, ref = run(A, B); res["all_max"] = {"mismatches": int((got != ref).sum()), "ref": int(ref[0, 0]), "got": float(got[0, 0])}
# alternating large cancellations
A = np.where(np.arange(16)[None, None, :] % 2 == 0, 504, -503) * np.ones((16, 32, 1), int); B = np.full((16, 32, 16), 127)
got, ref = run(A, B); res["cancel"] = {"mismatches": int((got != ref).sum()), "ref": int(ref[0, 0])}
# beyond 2^24 (expected inexact: documents where exactness ends)
A = np.full((32, 32, 16), 504); B = np.full((32, 32, 16), 127); A[:, :, ::3] = 503
got, ref = run(A, B); res["beyond_2p24"] = {"mismatches": int((got != ref).sum()), "ref": int(ref[0, 0]), "got": float(got[0, 0])}
print(json.dumps(res))
