"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np

ULP = {"f32": 2.0 ** -23, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}


def torch_dtype(name):
    import torch
    return {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[name]


def round_through(x: np.ndarray, name: str) -> np.ndarray:
    """Round an f32 array through the given storage dtype (RNE), back to f32."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch_dtype(name)).float().numpy()


def to_np(t) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def assert_close_accum(got: np.ndarray, want: np.ndarray, mag: np.ndarray, dtype: str, k_terms: int, what: str = ""):
    """|got - want| <= accumulation bound + 1 storage ulp.

    `mag` is sum_i |w_i x_i| per output (the quantity f32 rounding errors scale with); a sum of
    k_terms f32 products accumulated in any order is within ~sqrt(k)*eps*mag; we allow 8*eps*sqrt(k).
    For f16/bf16 outputs one extra ulp of the *result* is allowed (round-to-nearest boundary flips).
    """
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    bound = 8 * ULP["f32"] * np.sqrt(max(k_terms, 1)) * np.asarray(mag, dtype=np.float64)
    if dtype != "f32":
        bound = bound + ULP[dtype] * np.maximum(np.abs(want), np.abs(got)) * 1.01
    bound = bound + 1e-30
    err = np.abs(got - want)
    bad = err > bound
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.size} outside bound; worst err {err.max():.3e} "
                           f"(bound there {bound.flat[err.argmax()]:.3e}, want {want.flat[err.argmax()]:.6g})")


def rel_err_ref(actual: np.ndarray, expected: np.ndarray) -> float:
    """The reference tests' metric: max |a-e| / (1+|e|)  (fast_mmq.rs:1556-1582)."""
    return float(np.max(np.abs(actual - expected) / (1.0 + np.abs(expected))))
