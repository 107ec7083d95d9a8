"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of the CPU oracle (oracle/ggml_oracle.c, oracle/llama_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under mistral.rs_amd/ does (tests/test_no_oracle_in_product.py enforces it).

Parity status (details: DESIGN.md section 2, tests/test_oracle_ref.py, tests/test_golden.py): PINNED against the reference's own code
compiled for / executed on the host by oracle/build_ref.sh (oracle/_ref/*.so) -- block decode of all 10 formats, the Q8_1 quantizer,
the MMVQ dot products and complete kernels, GLU activations, RMSNorm family, RoPE, paged-cache scatter / gather / copy, paged
attention v1 / v2 (f32 path), HQQ pack / dequantize; "parity unpinned" for what lives in candle (CPU QMatMul arithmetic, oracle B),
the FP8-E4M3 codec (CUDA intrinsics) and the 16-bit instantiations of attention.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libggml_oracle.so")

# ggml type ids (reference: mistralrs-quant/src/gguf/archive.rs:73-160)
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K, BF16 = 10, 11, 12, 13, 14, 15, 30
TYPE_NAMES = {Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0",
              Q2_K: "q2_k", Q3_K: "q3_k", Q4_K: "q4_k", Q5_K: "q5_k", Q6_K: "q6_k"}
MMVQ_TYPES = tuple(TYPE_NAMES)
# the MoE launchers also take Q8_1 as a WEIGHT format (gguf/ffi.rs:268,424,601,800)
MOE_TYPE_NAMES = {**TYPE_NAMES, Q8_1: "q8_1"}


def build(force: bool = False) -> str:
    """Compile the oracle (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("ggml_oracle.c", "llama_oracle.c", "cpu_path_oracle.c", "ggml_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libggml_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fp16_to_fp32.restype = C.c_float
        _lib.orc_fp16_to_fp32.argtypes = [C.c_uint16]
        _lib.orc_fp32_to_fp16.restype = C.c_uint16
        _lib.orc_fp32_to_fp16.argtypes = [C.c_float]
        _lib.orc_glu_act.restype = C.c_float
        _lib.orc_glu_act.argtypes = [C.c_float, C.c_int]
        _lib.orc_fast_exp.restype = C.c_float
        _lib.orc_fast_exp.argtypes = [C.c_float]
        _lib.orc_silu_engine.restype = C.c_float
        _lib.orc_silu_engine.argtypes = [C.c_float]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def block_size(t: int) -> int:
    return lib().orc_block_size(t)


def type_size(t: int) -> int:
    return lib().orc_type_size(t)


def row_bytes(t: int, k: int) -> int:
    assert k % block_size(t) == 0, (t, k)
    return k // block_size(t) * type_size(t)


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def get_threads() -> int:
    return lib().orc_get_threads()


# ---------------------------------------------------------------- weights
def quantize(t: int, w: np.ndarray) -> np.ndarray:
    """w [N, K] f32 -> packed uint8 [N, row_bytes]."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, k = w.shape
    out = np.zeros((n, row_bytes(t, k)), dtype=np.uint8)
    rc = lib().orc_quantize_row(t, _p(w), _p(out), C.c_int64(n * k))
    if rc != 0:
        raise ValueError(f"oracle has no quantizer for ggml type {t}")
    return out


def quantize_imatrix(t: int, w: np.ndarray, imatrix: np.ndarray) -> np.ndarray:
    """w [N, K] f32 with the importance vector imatrix [K] (shared by every row) -> packed uint8 [N, row_bytes]: GGML's quantize_row_q{4,5,6}_K_impl
    with quant_weights (what candle's QTensor::quantize_imatrix runs; call sites gguf/mod.rs:238-252)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    qw = np.ascontiguousarray(imatrix, dtype=np.float32)
    n, k = w.shape
    if qw.shape != (k,):
        raise ValueError("imatrix must have one entry per input column")
    out = np.zeros((n, row_bytes(t, k)), dtype=np.uint8)
    for r in range(n):
        if lib().orc_quantize_row_imatrix(t, _p(w[r]), _p(out[r]), C.c_int64(k), _p(qw)) != 0:
            raise ValueError(f"oracle has no importance-weighted quantizer for ggml type {t}")
    return out


def dequantize(t: int, blocks: np.ndarray, k: int) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    n = blocks.size // row_bytes(t, k)
    out = np.empty((n, k), dtype=np.float32)
    lib().orc_dequantize_row(t, _p(blocks), _p(out), C.c_int64(n * k))
    return out


def random_blocks(t: int, n: int, k: int, seed: int = 0, d_scale: float = 0.01) -> np.ndarray:
    out = np.zeros((n, row_bytes(t, k)), dtype=np.uint8)
    lib().orc_random_blocks(t, _p(out), C.c_int64(n * (k // block_size(t))), C.c_uint64(seed),
                            C.c_float(d_scale))
    return out


# ---------------------------------------------------------------- activations
def pad512(k: int) -> int:
    return (k + 511) // 512 * 512


def quantize_q8_1(x: np.ndarray, k_padded: int | None = None) -> np.ndarray:
    """x [B, K] f32 -> uint8 [B, k_padded/32*36] (GPU semantics)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b, k = x.shape
    kp = pad512(k) if k_padded is None else k_padded
    out = np.zeros((b, kp // 32 * 36), dtype=np.uint8)
    lib().orc_quantize_q8_1(_p(x), _p(out), k, kp, b)
    return out


MMQ_D4, MMQ_DS4, MMQ_D2S6 = 0, 1, 2  # mmq_q8_1_ds_layout (mmq_gguf.cuh:64-68)


def mmq_layout(t: int) -> int:
    """The block_q8_1_mmq layout the reference pairs with weight type t (mmq_gguf.cuh:100-135, fast_mmq.rs ds_layout_for)."""
    return int(lib().orc_mmq_layout(t))


def quantize_q8_1_mmq(x: np.ndarray, layout: int, ne0: int | None = None, ids=None, ne1: int | None = None) -> np.ndarray:
    """x [rows, K] f32 -> uint8 [ne0/128, ne1, 144] block_q8_1_mmq (mmq_quantize.cu:104-198); ne0 = padded K (multiple of 512 as the callers
    pass it), ids = optional row gather (token ne1 reads row ids[i1])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, k = x.shape
    kp = pad512(k) if ne0 is None else ne0
    n1 = (len(ids) if ids is not None else rows) if ne1 is None else ne1
    out = np.zeros(((kp + 127) // 128, n1, 144), dtype=np.uint8)
    idp = None if ids is None else _p(np.ascontiguousarray(ids, dtype=np.int32))
    lib().orc_quantize_q8_1_mmq(_p(x), idp, _p(out), layout, C.c_int64(k), C.c_int64(k), C.c_int64(kp), C.c_int64(n1))
    return out


def matmul_q8_1_mmq(t: int, w: np.ndarray, n: int, k: int, y: np.ndarray, stride_row_x: int | None = None):
    """(out, mag) [ncols_y, n]: MMQ maths of weight type t against block_q8_1_mmq y [k_padded/128, ncols_y, 144] in the layout mmq_layout(t)."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    ncols = y.shape[1]
    out = np.empty((ncols, n), dtype=np.float32)
    mag = np.empty((ncols, n), dtype=np.float32)
    srx = k // block_size(t) if stride_row_x is None else stride_row_x
    lib().orc_matmul_q8_1_mmq(t, _p(w), n, k, C.c_int64(srx), _p(y), C.c_int64(ncols), _p(out), _p(mag))
    return out, mag


def quantize_q8_K(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    b, k = x.shape
    out = np.zeros((b, k // 256 * 292), dtype=np.uint8)
    lib().orc_quantize_q8_K(_p(x), _p(out), C.c_int64(b * k))
    return out


# ---------------------------------------------------------------- matmul oracles
def matmul_exact(t: int, w: np.ndarray, n: int, k: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, k)
    out = np.empty((x.shape[0], n), dtype=np.float32)
    lib().orc_matmul_exact(t, _p(w), n, k, _p(x), x.shape[0], _p(out))
    return out


def matmul_q8_1(t: int, w: np.ndarray, n: int, k: int, y: np.ndarray) -> np.ndarray:
    """y: Q8_1 bytes [B, stride*36] as produced by quantize_q8_1."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    b = y.shape[0]
    out = np.empty((b, n), dtype=np.float32)
    lib().orc_matmul_q8_1(t, _p(w), n, k, _p(y), y.shape[1] // 36, b, _p(out))
    return out


def matmul_q8_1_mag(t: int, w: np.ndarray, n: int, k: int, y: np.ndarray):
    """(out, mag): mag = SUM |terms| per output, the scale f32 accumulation error grows with."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    b = y.shape[0]
    out = np.empty((b, n), dtype=np.float32)
    mag = np.empty((b, n), dtype=np.float32)
    lib().orc_matmul_q8_1_ex(t, _p(w), n, k, _p(y), y.shape[1] // 36, b, _p(out), _p(mag))
    return out, mag


def matmul_cpu(t: int, w: np.ndarray, n: int, k: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, k)
    out = np.empty((x.shape[0], n), dtype=np.float32)
    lib().orc_matmul_cpu(t, _p(w), n, k, _p(x), x.shape[0], _p(out))
    return out


def gemv_cpu_fast(t: int, w: np.ndarray, n: int, k: int, x: np.ndarray) -> np.ndarray:
    """b = 1 reference-CPU-path matvec, vectorisable + OpenMP (llama_oracle.c): the cpu_baseline kernel."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty((1, n), dtype=np.float32)
    if lib().orc_gemv_cpu_fast(t, _p(w), n, k, _p(x), _p(out)) != 0:
        return matmul_cpu(t, w, n, k, x)
    return out


def gemv_engine(t: int, w: np.ndarray, n: int, k: int, x: np.ndarray) -> np.ndarray:
    """b = 1 matvec in the decode engine's f32 summation order (cpu_path_oracle.c orc_gemv_engine): the HIP engine equals it bit for bit."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty((1, n), dtype=np.float32)
    if lib().orc_gemv_engine(t, _p(w), n, k, _p(x), _p(out)) != 0:
        raise ValueError(f"the decode engine does not take ggml type {t}")
    return out


# ---------------------------------------------------------------- glue
def rms_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    d = x.shape[-1]
    lib().orc_rms_norm(_p(x), _p(w), _p(out), x.size // d, d, C.c_float(eps))
    return out


def rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray, positions: np.ndarray, neox: bool) -> np.ndarray:
    """x [tokens, heads, head_dim] -> rotated copy."""
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    cos = np.ascontiguousarray(cos, dtype=np.float32)
    sin = np.ascontiguousarray(sin, dtype=np.float32)
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    t, h, hd = x.shape
    lib().orc_rope(_p(x), _p(cos), _p(sin), _p(pos), t, h, hd, cos.shape[1] * 2, int(neox))
    return x


def fused_glu(a: np.ndarray, b: np.ndarray, act: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_fused_glu(_p(a), _p(b), _p(out), C.c_int64(a.size), act)
    return out


def attention(q, k, v, scale: float, softcap: float = 1.0) -> np.ndarray:
    """q [T,H,hd], k/v [S,KVH,hd] f32; causal with the T queries at the END of the S keys."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    t, h, hd = q.shape
    s, kvh, _ = k.shape
    out = np.empty_like(q)
    lib().orc_attention(_p(q), _p(k), _p(v), _p(out), t, s, h, kvh, hd, C.c_float(scale), C.c_float(softcap))
    return out


# ---------------------------------------------------------------- the in-tree CPU decode path and the engine's orders (cpu_path_oracle.c)
def fast_exp(x: float) -> float:
    return float(lib().orc_fast_exp(C.c_float(x)))


def rms_norm_candle(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """candle_nn::ops::rms_norm on the CPU: f32 sum of squares in element order, x / sqrt(mean + eps) * w."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    d = x.shape[-1]
    lib().orc_rms_norm_candle(_p(x), _p(w), _p(out), x.size // d, d, C.c_float(eps))
    return out


def rms_norm_engine(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """Same expression with the decode engine's summation tree (csrc/dec_core2.cuh act_finish)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    d = x.shape[-1]
    lib().orc_rms_norm_engine(_p(x), _p(w), _p(out), x.size // d, d, C.c_float(eps))
    return out


def fused_glu_engine(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_fused_glu_engine(_p(a), _p(b), _p(out), C.c_int64(a.size))
    return out


def attention_single_q_cpu(q, k, v, scale: float, n_kv_chunks: int = 1) -> np.ndarray:
    """attention/backends/cpu/single_q.rs for ONE query token: q [H, hd], k / v [S, KVH, hd] f32 -> [H, hd]."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    h, hd = q.shape
    s, kvh, _ = k.shape
    out = np.empty_like(q)
    lib().orc_attention_single_q_cpu(_p(q), _p(k), _p(v), _p(out), s, h, kvh, hd, C.c_float(scale), int(n_kv_chunks))
    return out


def attention_full_cpu(q, k, v, scale: float, window: int = 0) -> np.ndarray:
    """attention/backends/cpu/full.rs (tiled q-block path, causal / sliding-window binary mask rows) for the prompt of ONE sequence:
    q [T, H, hd], k / v [S, KVH, hd] f32 (S >= T: the last T positions are the queries) -> [T, H, hd]."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    t, h, hd = q.shape
    s, kvh, _ = k.shape
    out = np.empty_like(q)
    lib().orc_attention_full_cpu(_p(q), _p(k), _p(v), _p(out), t, s, h, kvh, hd, C.c_float(scale), int(window or 0))
    return out


def attention_engine(q, k, v, scale: float, bpw: int = 1, window: int = 0) -> np.ndarray:
    """The decode engine's attention order (csrc/dec_attn.cuh + ext_dec.hip dec_attn2_kernel), head size 128: q [H, 128], k / v [S, KVH, 128] -> [H, 128].
    window > 0: only the last `window` positions are attended (masked in place: the 32-token blocks stay anchored at position 0, as in the paged cache)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    h, hd = q.shape
    assert hd == 128
    s, kvh, _ = k.shape
    out = np.empty_like(q)
    lib().orc_attention_engine_w(_p(q), _p(k), _p(v), _p(out), s, h, kvh, C.c_float(scale), int(bpw), int(window or 0))
    return out


# ---------------------------------------------------------------- dtype helpers
def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """f32 -> bf16 bit pattern (uint16), round-to-nearest-even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def round_bf16(x: np.ndarray) -> np.ndarray:
    return from_bf16_bits(to_bf16_bits(x))


def patterned(n: int, salt: int, scale: float) -> np.ndarray:
    """The reference tests' deterministic input pattern (fast_mmq.rs:1545-1554):
    ((i*37 + salt*19) % 211 / 105 - 1) * scale."""
    i = np.arange(n, dtype=np.int64)
    return (((i * 37 + salt * 19) % 211).astype(np.float32) / 105.0 - 1.0) * np.float32(scale)


# ---------------------------------------------------------------- paged KV cache (numpy, small cases)
def kv_cache_write(key_cache: np.ndarray, value_cache: np.ndarray, key: np.ndarray, value: np.ndarray,
                   slot_mapping: np.ndarray) -> None:
    """reshape_and_cache semantics (reshape_and_cache_kernel.cu): K [nb,kvh,hd/x,bs,x], V [nb,kvh,hd,bs];
    negative slots are skipped."""
    nb, kvh, hdx, bs, x = key_cache.shape
    for t, slot in enumerate(slot_mapping):
        if slot < 0:
            continue
        b, o = divmod(int(slot), bs)
        key_cache[b, :, :, o, :] = key[t].reshape(kvh, hdx, x)
        value_cache[b, :, :, o] = value[t]


def kv_cache_gather(key_cache: np.ndarray, value_cache: np.ndarray, block_table: np.ndarray, ctx: int):
    """-> dense K, V [ctx, kvh, hd] for one sequence."""
    nb, kvh, hdx, bs, x = key_cache.shape
    ks, vs = [], []
    for pos in range(ctx):
        b, o = int(block_table[pos // bs]), pos % bs
        ks.append(key_cache[b, :, :, o, :].reshape(kvh, hdx * x))
        vs.append(value_cache[b, :, :, o])
    return np.stack(ks), np.stack(vs)


def paged_attention_ref(q, key_cache, value_cache, block_tables, context_lens, scale, softcap=1.0, alibi=None,
                        sinks=None, round_p=None):
    """Decode attention semantics of pagedattention.cuh:110-486: one query per sequence, softmax over the
    context with inv = 1/(sum + 1e-6), optional sink logit, probabilities optionally rounded through
    `round_p(array)->array` (the reference rounds them to the query dtype) before P.V.  f64 arithmetic.
    Pinned to the reference kernels themselves (v1, v2 + reduce, f32) by tests/test_oracle_ref.py via oracle/_ref/libref_pa.so."""
    q = np.asarray(q, dtype=np.float64)
    seqs, heads, hd = q.shape
    kvh = key_cache.shape[1]
    out = np.zeros((seqs, heads, hd))
    for s in range(seqs):
        ctx = int(context_lens[s])
        k, v = kv_cache_gather(key_cache.astype(np.float64), value_cache.astype(np.float64), block_tables[s], ctx)
        for h in range(heads):
            g = h // (heads // kvh)
            logit = scale * (k[:, g, :] @ q[s, h])
            if softcap != 1.0:
                logit = np.tanh(logit / softcap) * softcap
            if alibi is not None and alibi[h] != 0:
                # REFERENCE QUIRK, mirrored for parity: context_len is a uint32_t in the kernel (pagedattention.cuh:138), so the ALiBi
                # distance `token_idx - context_len + 1` (:283) is evaluated in unsigned arithmetic: 0 for the last token, 2^32 - k for
                # the k-th token before it; the f32 product and the f32 addition to qk are part of the observable result (the bias
                # swamps qk).  Upstream vLLM has `int context_len` and gets the intended -k.
                u = ((np.arange(ctx, dtype=np.int64) - ctx + 1) % (1 << 32)).astype(np.uint32)
                bias = np.float32(alibi[h]) * u.astype(np.float32)
                logit = (logit.astype(np.float32) + bias).astype(np.float32).astype(np.float64)
            m = logit.max() if sinks is None else max(logit.max(), float(sinks[h]))
            e = np.exp(logit - m)
            den = e.sum() + (np.exp(float(sinks[h]) - m) if sinks is not None else 0.0)
            p = e * (1.0 / (den + 1e-6))
            if round_p is not None:
                p = round_p(p.astype(np.float32)).astype(np.float64)
            out[s, h] = p @ v[:, g, :]
    return out.astype(np.float32)


# ---------------------------------------------------------------------------------------------------- FP8 (E4M3) KV cache
# OCP E4M3FN (bias 7, no inf, S.1111.111 = NaN, max 448): the storage format of cache_dtype 3.  The reference converts with the CUDA
# intrinsics __nv_cvt_float_to_fp8(x / scale, __NV_SATFINITE, __NV_E4M3) and __nv_cvt_fp8_to_halfraw (mistralrs-paged-attn/src/cuda/
# quantization/fp8/nvidia/quant_utils.cuh:24-29,79-88,135-145,187-217); their source is not in the tree, so this restates the format
# definition ("parity unpinned" vs the intrinsic; cross-checked against torch.float8_e4m3fn for in-range values in tests/test_fp8_kv.py).
def fp8_e4m3_decode(codes: np.ndarray) -> np.ndarray:
    c = np.asarray(codes, dtype=np.uint8).astype(np.int32)
    s, e, m = c >> 7, (c >> 3) & 15, c & 7
    val = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2((e - 7).astype(np.float64)))
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return (np.where(s == 1, -val, val)).astype(np.float32)


_FP8_GRID = None


def fp8_e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float -> E4M3 code: round to nearest, ties to the even code, saturate to +-448 (SATFINITE), NaN -> 0x7f | sign."""
    global _FP8_GRID
    if _FP8_GRID is None:
        _FP8_GRID = fp8_e4m3_decode(np.arange(0, 0x7F, dtype=np.uint8)).astype(np.float64)  # 127 non-negative finite values, ascending
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x.astype(np.float64))
    hi = np.clip(np.searchsorted(_FP8_GRID, a, side="left"), 0, len(_FP8_GRID) - 1)
    lo = np.clip(hi - 1, 0, None)
    dlo, dhi = a - _FP8_GRID[lo], _FP8_GRID[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo)
    code = np.where(a >= _FP8_GRID[-1], len(_FP8_GRID) - 1, code)
    code = np.where(np.isnan(a), 0x7F, code).astype(np.uint8)
    return (code | (np.signbit(x).astype(np.uint8) << 7)).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------- MoE router
def moe_router_topk(logits: np.ndarray, top_k: int, score_mode: int = 1, weight_mode: int = 0, renormalize: bool = True, bias=None,
                    expert_scale=None, clamp=None, norm_min: float = 0.0, output_scale: float = 1.0):
    """moe_router_topk_kernel (mistralrs-core/src/cuda/sort.cu:1186-1357; caller ops.rs:259-336) per row of logits [rows, E], in f32 with
    the kernel's summation structure (expert e in lane e % 32: strided partial sums, then an xor butterfly 16..1):
    clamp, NaN -> -inf; score = raw (0) / softmax (1) / sigmoid (2); selection = score + bias, NaN -> -inf; top_k arg-max rounds, ties to
    the lowest expert id; weight = score (0) / softmax over the picked raw logits (1) / sigmoid(raw) (2); renormalise by max(sum, norm_min);
    * output_scale (* expert_scale[id]).  Returns (ids uint32 [rows, k], weights f32 [rows, k]).  Pinned to the reference kernel run on
    the host (tests/test_oracle_ref.py)."""
    f = np.float32
    x = np.asarray(logits, dtype=np.float32)
    rows, E = x.shape
    slots = E // 32 if E > 32 else 1

    def lane_view(v, fill):  # [slots, 32] with expert e at [e // 32, e % 32]
        out = np.full(slots * 32, fill, dtype=np.float32)
        out[:E] = v
        return out.reshape(slots, 32)

    def butterfly(v, op):  # 32 lanes, masks 16..1, every lane ends with the same value (the kernel's order)
        v = v.astype(np.float32).copy()
        for m in (16, 8, 4, 2, 1):
            v = op(v, v[np.arange(32) ^ m]).astype(np.float32)
        return v[0]

    def softmax_lanes(v, limit):  # v [slots, 32]; entries >= limit (entry = lane + 32 i) are excluded when limit is not None
        idx = np.arange(32)[None, :] + 32 * np.arange(v.shape[0])[:, None]
        live = np.ones_like(v, dtype=bool) if limit is None else idx < limit
        mx = np.full(32, -np.inf, dtype=np.float32)
        for i in range(v.shape[0]):
            mx = np.where(live[i], np.maximum(mx, v[i]), mx)
        mx = butterfly(mx, np.maximum)
        with np.errstate(invalid="ignore"):
            ex = np.where(live, np.exp((v - mx).astype(np.float32)).astype(np.float32), f(0))
        s = np.zeros(32, dtype=np.float32)
        for i in range(v.shape[0]):
            s = (s + np.where(live[i], ex[i], f(0))).astype(np.float32)
        inv = f(1) / butterfly(s, np.add)
        return np.where(live, (ex * inv).astype(np.float32), ex)

    ids = np.zeros((rows, top_k), dtype=np.uint32)
    wts = np.zeros((rows, top_k), dtype=np.float32)
    for r in range(rows):
        v = x[r].copy()
        if clamp is not None:
            v = np.minimum(np.maximum(v, f(clamp[0])), f(clamp[1])).astype(np.float32)
        v = np.where(np.isnan(v), f(-np.inf), v)
        raw = lane_view(v, -np.inf)
        if score_mode == 1:
            score = softmax_lanes(raw, None)
        elif score_mode == 2:
            with np.errstate(over="ignore"):
                score = (f(1) / (f(1) + np.exp(-raw).astype(np.float32))).astype(np.float32)
        else:
            score = raw.copy()
        sel = score.copy()
        if bias is not None:
            sel.reshape(-1)[:E] = (sel.reshape(-1)[:E] + np.asarray(bias, dtype=np.float32)).astype(np.float32)
        sel = np.where(np.isnan(sel), f(-np.inf), sel).reshape(-1)
        if E < 32:
            sel[E:] = -np.inf  # lanes beyond the experts hold -inf logits in the kernel (sigmoid(-inf) = 0 is never selected before real ones unless all tie)
        out = np.zeros(top_k, dtype=np.float32)
        flat_score, flat_raw = score.reshape(-1), raw.reshape(-1)
        for k in range(top_k):
            best = int(np.flatnonzero(sel == sel.max())[0])  # ties: lowest expert id
            ids[r, k] = best
            if weight_mode == 1:
                out[k] = flat_raw[best]
            elif weight_mode == 2:
                out[k] = f(1) / (f(1) + np.exp(-flat_raw[best], dtype=np.float32))
            else:
                out[k] = flat_score[best]
            sel[best] = -np.inf
        oslots = max(slots, (top_k + 31) // 32)
        ow = np.zeros(oslots * 32, dtype=np.float32)
        ow[:top_k] = out
        ow = ow.reshape(oslots, 32)
        if weight_mode == 1:
            ow = softmax_lanes(ow, top_k)
        if renormalize:
            s = np.zeros(32, dtype=np.float32)
            idx = np.arange(32)[None, :] + 32 * np.arange(oslots)[:, None]
            for i in range(oslots):
                s = (s + np.where(idx[i] < top_k, ow[i], f(0))).astype(np.float32)
            tot = np.maximum(butterfly(s, np.add), f(norm_min))
            ow = (ow * (f(1) / tot)).astype(np.float32)
        w = ow.reshape(-1)[:top_k]
        sc = np.full(top_k, f(output_scale), dtype=np.float32)
        if expert_scale is not None:
            sc = (sc * np.asarray(expert_scale, dtype=np.float32)[ids[r]]).astype(np.float32)
        wts[r] = (w * sc).astype(np.float32)
    return ids, wts


# ------------------------------------------------------------------------------------------------ MoE expert kernels (restatements)
# ---------------------------------------------------------------- sampling: top-k over a large vocabulary (mistralrs-core/src/cuda/sort.cu:1502-1823)
def _tree32_lane0(v: np.ndarray) -> np.float32:
    """warp_reduce_sum_f32 (sort.cu:1470-1476) as lane 0 sees it: v += shfl_down(v, 16), 8, 4, 2, 1 over 32 lanes (a lane whose source is out of range adds itself)."""
    v = np.asarray(v, dtype=np.float32).copy()
    for off in (16, 8, 4, 2, 1):
        sh = v.copy()
        sh[: 32 - off] = v[off:]
        v = (v + sh).astype(np.float32)
    return v[0]


def _block_sum_256(partials: np.ndarray) -> np.float32:
    """block_reduce_sum_f32 (sort.cu:1478-1497) for a 256-thread block: 8 warp trees, then the tree of the 8 warp sums padded with zeros."""
    ws = np.zeros(32, dtype=np.float32)
    for w in range(8):
        ws[w] = _tree32_lane0(partials[w * 32:(w + 1) * 32])
    return _tree32_lane0(ws)


def topk_large_packed(logits: np.ndarray, k: int, inv_temperature: float, chunk_size: int = 2048) -> dict:
    """topk_large_stage1_f32 + topk_large_stage2_f32_packed (sort.cu:1502-1600, 1708-1823; host side ops.rs:691-828) for ONE f32 row, restated: the k largest
    logits in (value descending, index ascending) order -- NaN and -inf are never selected, missing entries are (-inf, 0) --, per-chunk maxima / partial softmax sums
    in the reference's f32 association (256 strided per-thread partials, 32-lane shuffle-down trees), denom and global max.  Returns the packed row
    [k values][k indices as f32][denom][max] and the stage-1 workspace."""
    x = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1)
    n = x.size
    it = np.float32(inv_temperature)
    nblocks = (n + chunk_size - 1) // chunk_size
    bv = np.full((nblocks, k), -np.inf, dtype=np.float32)
    bi = np.zeros((nblocks, k), dtype=np.uint32)
    bm = np.empty(nblocks, dtype=np.float32)
    bs = np.empty(nblocks, dtype=np.float32)

    def best_k(vals, ids):
        ok = ~np.isnan(vals) & (vals > -np.inf)
        order = np.lexsort((ids[ok], -vals[ok].astype(np.float64)))[:k]  # value descending (-0.0 == +0.0), index ascending
        return vals[ok][order], ids[ok][order]
    with np.errstate(over="ignore", invalid="ignore"):
        for c in range(nblocks):
            lo, hi = c * chunk_size, min((c + 1) * chunk_size, n)
            v, i = best_k(x[lo:hi], np.arange(lo, hi, dtype=np.uint32))
            bv[c, : v.size], bi[c, : v.size] = v, i
            bm[c] = bv[c, 0] * it if hi > lo else -np.inf
            part = np.zeros(256, dtype=np.float32)
            for t in range(256):
                seg = x[lo + t:hi:256]
                acc = np.float32(0)
                for cnd in seg:
                    if cnd != cnd:
                        acc = np.float32(np.nan)
                    elif bm[c] != -np.inf:
                        acc = np.float32(acc + np.exp(np.float32(cnd * it - bm[c]), dtype=np.float32))
                part[t] = acc
            bs[c] = _block_sum_256(part)
        gmax = np.float32(np.max(bm)) if nblocks else np.float32(-np.inf)
        part = np.zeros(256, dtype=np.float32)
        if gmax != -np.inf:
            for b in range(nblocks):
                part[b % 256] = np.float32(part[b % 256] + np.float32(bs[b] * np.exp(np.float32(bm[b] - gmax), dtype=np.float32)))
        denom = _block_sum_256(part)
    v, i = best_k(bv.reshape(-1), bi.reshape(-1))  # ties across chunks: lower candidate position = lower chunk = lower index
    packed = np.empty(2 * k + 2, dtype=np.float32)
    packed[:k], packed[k:2 * k] = -np.inf, 0.0
    packed[: v.size], packed[k:k + i.size] = v, i.astype(np.float32)
    packed[2 * k], packed[2 * k + 1] = denom, gmax
    return {"packed": packed, "block_values": bv, "block_indices": bi, "block_maxes": bm, "block_sums": bs}


def top1_large_packed(logits: np.ndarray, chunk_size: int = 2048):
    """top1_large_stage1_f32 + top1_large_stage2_f32_packed (sort.cu:1825-1912, 2071-2143) for ONE f32 row, restated: per chunk (max, lowest index) -- a chunk with a NaN
    reports (NaN, 0), a chunk without a selectable value (-inf, 0) --, then the best chunk (lowest position on ties).  Returns (packed [2] = max, token as f32; token id;
    block_values; block_indices): token 0xffffffff and packed (NaN, NaN) when the row holds a NaN, token 0 when nothing is selectable."""
    x = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1)
    n = x.size
    nblocks = (n + chunk_size - 1) // chunk_size
    bv, bi = np.empty(nblocks, dtype=np.float32), np.zeros(nblocks, dtype=np.uint32)
    for c in range(nblocks):
        seg = x[c * chunk_size:(c + 1) * chunk_size]
        if np.isnan(seg).any():
            bv[c], bi[c] = np.nan, 0
            continue
        ok = seg > -np.inf
        if not ok.any():
            bv[c], bi[c] = -np.inf, 0
            continue
        m = seg[ok].max()
        j = int(np.nonzero(ok & (seg == m))[0][0])  # `>` keeps the first of equal values (-0.0 == +0.0)
        bv[c], bi[c] = seg[j], c * chunk_size + j
    if np.isnan(bv).any():
        return np.array([np.nan, np.nan], dtype=np.float32), np.uint32(0xFFFFFFFF), bv, bi
    ok = bv > -np.inf
    if not ok.any():
        return np.array([-np.inf, 0.0], dtype=np.float32), np.uint32(0), bv, bi
    m = bv[ok].max()
    p = int(np.nonzero(ok & (bv == m))[0][0])
    return np.array([bv[p], np.float32(bi[p])], dtype=np.float32), np.uint32(bi[p]), bv, bi


def sample_topk_host(packed: np.ndarray, k: int, temperature: float, top_p: float = 1.0, min_p: float = 0.0):
    """The host half of Sampler::sample_topk_on_device (sampler.rs:1189-1236): probabilities of the k candidates under the FULL softmax, then the top-p cut
    (`top_p_cutoff` = top_p * sum of the kept probabilities, cumulative sum in candidate order) and the min-p cut (threshold = first probability * min_p).
    Returns (token ids [k], reporting_probs [k], filtered probs [k]) -- the weights WeightedIndex draws from."""
    vals, ids = packed[:k].astype(np.float32), packed[k:2 * k].astype(np.uint32)
    denom, gmax = np.float32(packed[2 * k]), np.float32(packed[2 * k + 1])
    inv_t = np.float32(1.0 / temperature)
    with np.errstate(over="ignore", invalid="ignore"):
        rep = (np.exp((vals * inv_t - gmax).astype(np.float32), dtype=np.float32) / denom).astype(np.float32)
    probs = rep.copy()
    if 0.0 < top_p < 1.0:
        cutoff = np.float32(top_p) * np.float32(sum(np.float32(p) for p in probs))  # f32 running sum in candidate order (Iterator::sum::<f32>)
        cum = np.float32(0)
        for j in range(k):
            if cum >= cutoff:
                probs[j] = 0.0
            else:
                cum = np.float32(cum + probs[j])
    if 0.0 < min_p < 1.0:
        thr = np.float32(probs[0] * np.float32(min_p)) if k else np.float32(0)
        probs[thr >= probs] = 0.0
    return ids, rep, probs


def _expert_rows(w: np.ndarray, e: int, n: int) -> np.ndarray:
    return w[e * n:(e + 1) * n]


def moe_act(x: np.ndarray, act_type: int) -> np.ndarray:
    """MoE activation codes (indexed_moe.cu:1168-1177, gguf/cuda.rs:1427-1428): 0 = gelu_pytorch_tanh, anything else = silu; f32."""
    x = np.asarray(x, dtype=np.float32)
    f = np.float32
    if act_type == 0:
        x3 = (x * x * x).astype(np.float32)
        inner = (f(0.7978845608028654) * (x + f(0.044715) * x3)).astype(np.float32)
        return (f(0.5) * x * (f(1.0) + np.tanh(inner, dtype=np.float32))).astype(np.float32)
    with np.errstate(over="ignore"):  # exp(-x) -> inf for very negative x: x / inf = -0, as on the device
        return (x / (f(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def moe_gemv_fused_gate_up(t: int, gate_w: np.ndarray, up_w: np.ndarray, n: int, k: int, y: np.ndarray, indices: np.ndarray, topk: int,
                           act_type: int, with_mag: bool = False):
    """moe_gemv_fused_gate_up_impl (kernels/indexed_moe/indexed_moe.cu:1342-1407): task = token * topk + slot, e = indices[task],
    out[task][row] = (up_w[e][row] . y[token]) * act(gate_w[e][row] . y[token]); y = Q8_1 rows per TOKEN.  gate_w / up_w: [E * n, row_bytes].
    with_mag: also return (gate, gate_mag, up, up_mag) for error budgeting."""
    tasks = len(indices)
    out = np.empty((tasks, n), dtype=np.float32)
    extra = []
    for task in range(tasks):
        e, tok = int(indices[task]), task // topk
        g, gm = matmul_q8_1_mag(t, _expert_rows(gate_w, e, n), n, k, y[tok:tok + 1])
        u, um = matmul_q8_1_mag(t, _expert_rows(up_w, e, n), n, k, y[tok:tok + 1])
        out[task] = (u[0] * moe_act(g[0], act_type)).astype(np.float32)
        extra.append((g[0], gm[0], u[0], um[0]))
    return (out, extra) if with_mag else out


def moe_gemv_down_aggregate(t: int, w: np.ndarray, n: int, k: int, y: np.ndarray, indices: np.ndarray, topk_weights: np.ndarray, topk: int,
                            out: np.ndarray | None = None, with_mag: bool = False):
    """moe_gemv_down_aggregate_impl (indexed_moe.cu:1420-1486): out[token][row] += (w[e][row] . y[task]) * topk_weights[task], slots added in
    order (the device's atomics may retire in any order; for topk == 2 on a zero-filled output the sum is order-independent)."""
    tasks = len(indices)
    batch = tasks // topk
    out = np.zeros((batch, n), dtype=np.float32) if out is None else out
    mag = np.zeros((batch, n), dtype=np.float64)
    tw = np.asarray(topk_weights, dtype=np.float32).reshape(-1)
    for task in range(tasks):
        e, tok = int(indices[task]), task // topk
        d, m = matmul_q8_1_mag(t, _expert_rows(w, e, n), n, k, y[task:task + 1])
        out[tok] = (out[tok] + (d[0] * tw[task]).astype(np.float32)).astype(np.float32)
        mag[tok] += m[0].astype(np.float64) * abs(float(tw[task]))
    return (out, mag) if with_mag else out


def moe_dispatch(topk_ids: np.ndarray, num_experts: int, topk: int):
    """launch_moe_dispatch (kernels/moe_grouped/moe_grouped.cu:630-676,1106-1132): counting sort of the flattened top-k ids by expert.
    Returns (expert_bounds [E + 1], sorted_token_ids, sorted_source_ids, expert_counts, expert_cursors (final = bounds[1:])).
    Order inside an expert's segment: ascending flat index (the reference's atomic cursors leave it unspecified; this is the order its
    kernels produce when the threads run in index order)."""
    ids = np.asarray(topk_ids, dtype=np.int32).reshape(-1)
    counts = np.bincount(ids, minlength=num_experts).astype(np.int32)
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    order = np.argsort(ids, kind="stable").astype(np.int32)
    return bounds, order, (order // topk).astype(np.int32), counts, bounds[1:].copy()


def moe_weighted_reduce_flat(inputs: np.ndarray, topk_weights: np.ndarray, out_kind: str = "f32") -> np.ndarray:
    """moe_weighted_reduce_flat_kernel (moe_grouped.cu:678-702): inputs [tokens, topk, hidden] (already widened to f32 exactly),
    acc = sum over slots IN ORDER of in * w, f32 multiply then f32 add; result rounded to out_kind ('f32' | 'f16' | 'bf16')."""
    x = np.asarray(inputs, dtype=np.float32)
    w = np.asarray(topk_weights, dtype=np.float32)
    acc = np.zeros((x.shape[0], x.shape[2]), dtype=np.float32)
    for s in range(x.shape[1]):
        acc = (acc + (x[:, s, :] * w[:, s:s + 1]).astype(np.float32)).astype(np.float32)
    if out_kind == "f16":
        return acc.astype(np.float16).astype(np.float32)
    if out_kind == "bf16":
        return round_bf16(acc)
    return acc


def moe_grouped_gemm(t: int, w: np.ndarray, n: int, k: int, y: np.ndarray, expert_bounds: np.ndarray, sorted_token_ids: np.ndarray,
                     topk_weights, topk: int, input_dim1: int, out: np.ndarray):
    """moe_tiled_gemm_impl (moe_grouped.cu:716-902): for sorted position ti of expert e, flat = sorted_token_ids[ti]:
    y row = ti (input_dim1 == 0) | flat // topk (== 1) | flat (else); acc = w[e] . y_row;
    topk_weights given: out[flat // topk] += acc * topk_weights[flat] (in ti order), else out[ti] = acc.  Returns (out, mag)."""
    mag = np.zeros(out.shape, dtype=np.float64)
    tw = None if topk_weights is None else np.asarray(topk_weights, dtype=np.float32).reshape(-1)
    for e in range(len(expert_bounds) - 1):
        for ti in range(int(expert_bounds[e]), int(expert_bounds[e + 1])):
            flat = int(sorted_token_ids[ti])
            row = ti if input_dim1 == 0 else (flat // topk if input_dim1 == 1 else flat)
            d, m = matmul_q8_1_mag(t, _expert_rows(w, e, n), n, k, y[row:row + 1])
            if tw is not None:
                tok = flat // topk
                out[tok] = (out[tok] + (d[0] * tw[flat]).astype(np.float32)).astype(np.float32)
                mag[tok] += m[0].astype(np.float64) * abs(float(tw[flat]))
            else:
                out[ti] = d[0]
                mag[ti] = m[0]
    return out, mag


def gemv_dense(a: np.ndarray, x: np.ndarray, bias=None):
    """Dense decode GEMV of gemv_kernel_batched (kernels/gemv/gemv.cu:50-160): y[b][row] = sum_k a[row][k] * x[b][k] + bias[row] with the
    inputs already widened to f32.  Returns (y in f64 before the final rounding to T, mag = sum |terms|): the kernel accumulates with f32
    fma in a thread-strided order and butterflies, so comparisons use the f32-accumulation bound on `mag` plus one rounding of T."""
    a64, x64 = np.asarray(a, dtype=np.float64), np.asarray(x, dtype=np.float64)
    y = x64 @ a64.T
    mag = np.abs(x64) @ np.abs(a64).T
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)[None, :]
        mag = mag + np.abs(np.asarray(bias, dtype=np.float64))[None, :]
    return y, mag
