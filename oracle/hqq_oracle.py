"""oracle/hqq_oracle.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and scripts/gen_golden.py, never by mistral.rs_amd/).

CPU restatement (numpy) of the reference's HQQ path:
  * bit packing            mistralrs-quant/kernels/hqq/hqq_bitpack.cu:7-135 (device) == HqqBits::bitpack_type CPU arms (hqq/mod.rs:150-400)
  * unpack + dequantize    mistralrs-quant/kernels/hqq/hqq.cu:26-35 (8 bit), :94-110 (4), :171-186 (2), :278-301 (1), :399-427 (3 bit)
  * quantizer              mistralrs-quant/src/hqq/quantize.rs:9-84 + optimize.rs:29-95 (proximal solver, lp 0.7, beta 10, kappa 1.01)
  * layer forward          hqq/mod.rs:1092-1100 (dequantize, then the dense linear incl. bias)

Pinning: pack / dequantize are checked bit-for-bit against the reference's own kernels compiled for the host by oracle/build_ref.sh
(oracle/_ref/libref_hqq.so, f32 instantiation; tests/test_oracle_ref.py) and frozen in tests/golden/hqq_*.npz.  The f16 / bf16
instantiations evaluate `(T(q) - zero) * scale` in T: restated here as f32 operations rounded to T after each step (exact for one
add / one multiply of two T operands); they are NOT executable here (no CUDA half types on the host) -> "restated" only.
The quantizer follows candle tensor semantics (f32 throughout, round = half away from zero, mean over the group axis): parity
unpinned (candle is not buildable here), exercised through the round trip  dequantize(quantize(W)) ~ W.
"""
from __future__ import annotations

import numpy as np

PACK = {8: 1, 4: 2, 3: 10, 2: 4, 1: 8}


def _round_t(x: np.ndarray, dtype: str) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    if dtype == "f32":
        return x
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        b = x.view(np.uint32).astype(np.uint64)
        r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint32) << 16
        out = r.astype(np.uint32).view(np.float32).copy()
        nan = np.isnan(x)
        out[nan] = np.nan
        return out
    raise ValueError(dtype)


def pack(bits: int, q: np.ndarray) -> np.ndarray:
    """q [rows][w] unpacked values -> packed [rows / P][w]; chunk i (rows i*step ..) most significant first."""
    p = PACK[bits]
    q = np.asarray(q)
    rows, w = q.shape
    step = rows // p
    if bits == 8:
        return q.astype(np.uint8).copy()
    out = np.zeros((step, w), dtype=np.uint32)
    mask = (1 << bits) - 1
    for i in range(p):
        v = q[i * step:(i + 1) * step].astype(np.uint32) & mask
        out |= v << ((27 - 3 * i) if bits == 3 else (8 - bits * (i + 1)))
    return out.astype(np.uint32).view(np.int32) if bits == 3 else out.astype(np.uint8)


def unpack(bits: int, wq: np.ndarray) -> np.ndarray:
    p = PACK[bits]
    v = np.asarray(wq).astype(np.int64) & (0xFFFFFFFF if bits == 3 else 0xFF)
    h, w = v.shape
    out = np.empty((p * h, w), dtype=np.uint8)
    for c in range(p):
        sh = (27 - 3 * c) if bits == 3 else (8 - bits * (c + 1))
        out[c * h:(c + 1) * h] = (v >> sh) & ((1 << bits) - 1)
    return out


def dequantize(bits: int, wq: np.ndarray, scale: np.ndarray, zero: np.ndarray, dtype: str = "f32") -> np.ndarray:
    """out[(c*h + r)][j] = (T(q) - zero[j]) * scale[j] in T (scale / zero: [w] already representable in T) -> f32 array of T values."""
    q = unpack(bits, wq).astype(np.float32)
    z = np.asarray(zero, dtype=np.float32).reshape(1, -1)
    s = np.asarray(scale, dtype=np.float32).reshape(1, -1)
    return _round_t(_round_t(q - z, dtype) * s, dtype)


def _shrink_lp(x: np.ndarray, beta: float, p: float) -> np.ndarray:
    ax = np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ax - np.float32(1.0 / beta) * np.power(ax, np.float32(p - 1.0), dtype=np.float32)
    return (np.sign(x) * np.maximum(t, 0)).astype(np.float32)


def _round_half_away(x: np.ndarray) -> np.ndarray:
    return (np.sign(x) * np.floor(np.abs(x) + np.float32(0.5))).astype(np.float32)


def quantize(w: np.ndarray, bits: int, group_size: int, steps: int = 20, round_zeros: bool = False):
    """HqqLayer::quantize with axis 0, channel_wise: returns (packed Wq, scale [1][G], zero [1][G]) with W.reshape(group_size, -1)."""
    wf = np.asarray(w, dtype=np.float32).reshape(group_size, -1)
    mn, mx = wf.min(axis=0, keepdims=True), wf.max(axis=0, keepdims=True)
    max_v = np.float32(round(2.0 ** bits - 1.0))
    with np.errstate(divide="ignore"):
        scale = np.clip(max_v / (mx - mn), 0.0, 2e4).astype(np.float32)
    zero = (-mn * scale).astype(np.float32)
    if round_zeros:
        zero = _round_half_away(zero)
    beta, best = 10.0, 1e4
    for _ in range(steps):
        wq = np.clip(_round_half_away(wf * scale + zero), 0.0, max_v)
        wr = (wq - zero) / scale
        we = _shrink_lp(wf - wr, beta, 0.7)
        zero = np.mean(wq - (wf - we) * scale, axis=0, keepdims=True, dtype=np.float32)
        beta *= 1.01
        err = float(np.mean(np.abs(wf - wr), dtype=np.float32))
        if err < best:
            best = err
        else:
            break
    wq = np.clip(_round_half_away(wf * scale + zero), 0.0, max_v)
    return pack(bits, wq.astype(np.uint32)), (np.float32(1.0) / scale).astype(np.float32), zero.astype(np.float32)


def forward(x: np.ndarray, bits: int, wq, scale, zero, w_shape, bias=None, dtype: str = "f32") -> np.ndarray:
    """x [.., K] @ dequantize(W).reshape(w_shape)^T (+ bias), accumulated in f64 (the checker, not a rounding model of the GEMM)."""
    w = dequantize(bits, wq, scale.reshape(-1), zero.reshape(-1), dtype).reshape(w_shape).astype(np.float64)
    y = np.asarray(x, dtype=np.float64) @ w.T
    return y + np.asarray(bias, dtype=np.float64) if bias is not None else y
