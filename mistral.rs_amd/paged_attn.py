"""Host-side mirror of the `mistralrs-paged-attn` op surface
(mistralrs-paged-attn/src/cuda/backend/paged_attention.rs:453 `paged_attention`, :694 `reshape_and_cache`,
backend/gather_kv.rs `gather_kv_cache`, backend/cache.rs `copy_blocks`).

Same argument meaning, shape checks and v1/v2 selection rule as the reference; every call goes through the
C ABI of libmistralrspagedattention.so on the caller's current stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_CODE = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
_CACHE_CODE = {**_CODE, torch.uint8: 3}  # uint8 cache = fp8 E4M3 codes (cache_dtype 3): needs k_scale / v_scale (f32 scalars on the device)
if hasattr(torch, "float8_e4m3fn"):
    _CACHE_CODE[torch.float8_e4m3fn] = 3


def _scales(cache: torch.Tensor, k_scale, v_scale):
    """(k_scale ptr, v_scale ptr) for an fp8 cache, (None, None) otherwise; mirrors the Option<&Tensor> pair of the reference op."""
    if _CACHE_CODE.get(cache.dtype) != 3:
        return None, None
    if k_scale is None or v_scale is None:
        raise ValueError("an fp8 KV cache needs k_scale and v_scale")
    for t in (k_scale, v_scale):
        if t.dtype != torch.float32 or t.numel() != 1 or not t.is_cuda:
            raise ValueError("k_scale / v_scale must be single-element f32 GPU tensors")
    return k_scale.data_ptr(), v_scale.data_ptr()
_TAG = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
_HEAD_SIZES = (64, 80, 96, 112, 128, 192, 256, 512)
PARTITION_SIZE = 512
_vp, _i, _f, _u = C.c_void_p, C.c_int, C.c_float, C.c_uint32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_ws: dict = {}


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = (device.index, _stream())
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


def _align_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def kv_cache_shapes(num_blocks: int, num_kv_heads: int, head_size: int, block_size: int, dtype: torch.dtype):
    """K [nb, kvh, hd/x, bs, x], V [nb, kvh, hd, bs]  (mistralrs-core/src/paged_attention/cache_engine.rs:458-484)."""
    x = 16 // torch.empty((), dtype=dtype).element_size()
    return (num_blocks, num_kv_heads, head_size // x, block_size, x), (num_blocks, num_kv_heads, head_size, block_size)


def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                      slot_mapping: torch.Tensor, k_scale: torch.Tensor | None = None, v_scale: torch.Tensor | None = None) -> None:
    """key/value [num_tokens, kv_heads, head_size] (row stride may exceed kv_heads*head_size);
    slot_mapping int64 [num_tokens]; negative slots are skipped."""
    if key.dtype not in _CODE:
        raise ValueError(f"dtype {key.dtype} is not supported")
    if key.dim() != 3 or value.shape != key.shape:
        raise ValueError(f"shape mismatch k {tuple(key.shape)} and v {tuple(value.shape)}")
    num_tokens, num_heads, head_size = key.shape
    nb, kvh, hs_x, block_size, x = key_cache.shape
    if kvh != num_heads or hs_x * x != head_size:
        raise ValueError(f"shape mismatch k {tuple(key.shape)} and key_cache {tuple(key_cache.shape)}")
    if tuple(value_cache.shape) != (nb, kvh, head_size, block_size):
        raise ValueError(f"shape mismatch key_cache {tuple(key_cache.shape)} and value_cache {tuple(value_cache.shape)}")
    if slot_mapping.dtype != torch.int64 or slot_mapping.numel() != num_tokens:
        raise ValueError(f"shape mismatch slot_mapping {tuple(slot_mapping.shape)}, expected {(num_tokens,)}")
    if key.stride(2) != 1 or key.stride(1) != head_size or value.stride(2) != 1 or value.stride(1) != head_size:
        raise ValueError("key/value must be contiguous within a token")
    fn = _lib.sym("paged_attn", "reshape_and_cache",
                  [_vp] * 5 + [_i] * 7 + [_vp, _u, _u, _vp, _vp])
    ks, vs = _scales(key_cache, k_scale, v_scale)
    fn(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr(),
       num_tokens, num_heads, head_size, block_size, x, key.stride(0), value.stride(0), _stream(),
       _CODE[key.dtype], _CACHE_CODE[key_cache.dtype], ks, vs)


def paged_attention(q: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor, block_tables: torch.Tensor,
                    context_lens: torch.Tensor, max_context_len: int, softmax_scale: float, softcapping: float = 1.0,
                    alibi_slopes: torch.Tensor | None = None, sinks: torch.Tensor | None = None,
                    force: str | None = None, k_scale: torch.Tensor | None = None, v_scale: torch.Tensor | None = None) -> torch.Tensor:
    """q [num_seqs, num_heads, head_size] (row stride free) -> out, same shape.  block_tables uint32/int32
    [num_seqs, max_blocks], context_lens uint32/int32 [num_seqs].  `force` in {None, 'v1', 'v2'} (tests)."""
    if q.dtype not in _CODE:
        raise ValueError(f"dtype {q.dtype} is not supported")
    if q.dim() != 3:
        raise ValueError("q must have rank 3")
    num_seqs, num_heads, head_size = q.shape
    if head_size not in _HEAD_SIZES:
        raise ValueError("`head_size` must be one of 64, 80, 96, 112, 128, 192, 256 or 512")
    if block_tables.dim() != 2 or block_tables.shape[0] != num_seqs:
        raise ValueError(f"shape mismatch block_tables {tuple(block_tables.shape)}, expected {(num_seqs, 'max_blocks')}")
    max_blocks = block_tables.shape[1]
    nb, kvh, hs_x, block_size, x = key_cache.shape
    if hs_x != head_size // x:
        raise ValueError(f"shape mismatch key_cache {tuple(key_cache.shape)}")
    if tuple(value_cache.shape) != (nb, kvh, head_size, block_size):
        raise ValueError(f"shape mismatch key_cache {tuple(key_cache.shape)} and value_cache {tuple(value_cache.shape)}")
    if tuple(context_lens.shape) != (num_seqs,):
        raise ValueError(f"shape mismatch context_lens {tuple(context_lens.shape)}, expected {(num_seqs,)}")
    if q.stride(2) != 1 or q.stride(1) != head_size:
        raise ValueError("q must be contiguous within a sequence row")
    eff_max = min(max_blocks * block_size, max_context_len)
    max_parts = (eff_max + PARTITION_SIZE - 1) // PARTITION_SIZE
    use_v1 = (max_parts == 1 or num_seqs * num_heads > 512) and PARTITION_SIZE % block_size == 0
    if force is not None:
        use_v1 = force == "v1"
    out = torch.empty(num_seqs, num_heads, head_size, dtype=q.dtype, device=q.device)
    al = alibi_slopes.data_ptr() if alibi_slopes is not None else None
    sk = sinks.data_ptr() if sinks is not None else None
    ks, vs = _scales(key_cache, k_scale, v_scale)
    fp8 = ks is not None
    if fp8 and block_size not in (16, 32):
        raise ValueError("an fp8 KV cache needs block_size 16 or 32")
    mixed = key_cache.dtype != q.dtype and not fp8
    if mixed and not (q.dtype == torch.float32 and key_cache.dtype == torch.bfloat16):
        raise ValueError(f"unsupported (query, cache) dtype pair ({q.dtype}, {key_cache.dtype})")
    common_tail = [_i, _f, _f, _vp, _vp] + [_i] * 9 + [_vp]
    if mixed:
        # MI355X-native entry: f32 activations over a bf16 cache
        fn = _lib.sym("paged_attn", "mrs_paged_attention_f32_bf16", [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + common_tail + [_vp])
        exp_sums = max_logits = tmp_out = None
        if not use_v1:
            ws = _paged_ws(q, num_seqs, num_heads, max_parts, head_size)
            tmp_out, exp_sums, max_logits = ws
        fn(0 if use_v1 else 1, out.data_ptr(), exp_sums, max_logits, tmp_out, q.data_ptr(), key_cache.data_ptr(),
           value_cache.data_ptr(), al, kvh, softmax_scale, softcapping, block_tables.data_ptr(), context_lens.data_ptr(),
           block_size, eff_max, num_seqs, num_heads, head_size, max_blocks, q.stride(0), key_cache.stride(0),
           key_cache.stride(1), _stream(), sk)
        return out
    tail = common_tail + [_u, _vp, _vp, _vp]
    if use_v1:
        fn = _lib.sym("paged_attn", f"paged_attention_v1_{_TAG[q.dtype]}", [_vp] * 5 + tail)
        fn(out.data_ptr(), q.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), al, kvh, softmax_scale, softcapping,
           block_tables.data_ptr(), context_lens.data_ptr(), block_size, eff_max, num_seqs, num_heads, head_size,
           max_blocks, q.stride(0), key_cache.stride(0), key_cache.stride(1), _stream(), _CACHE_CODE[key_cache.dtype], ks, vs, sk)
    else:
        tmp_out, exp_sums, max_logits = _paged_ws(q, num_seqs, num_heads, max_parts, head_size)
        fn = _lib.sym("paged_attn", f"paged_attention_v2_{_TAG[q.dtype]}", [_vp] * 8 + tail)
        fn(out.data_ptr(), exp_sums, max_logits, tmp_out, q.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), al, kvh,
           softmax_scale, softcapping, block_tables.data_ptr(), context_lens.data_ptr(), block_size, eff_max, num_seqs,
           num_heads, head_size, max_blocks, q.stride(0), key_cache.stride(0), key_cache.stride(1), _stream(),
           _CACHE_CODE[key_cache.dtype], ks, vs, sk)
    return out


def _paged_ws(q, num_seqs, num_heads, max_parts, head_size):
    """v2 workspace carve-up, as backend/paged_attention.rs:353-365."""
    tmp_bytes = num_seqs * num_heads * max_parts * head_size * q.element_size()
    es_bytes = num_seqs * num_heads * max_parts * 4
    es_off = _align_up(tmp_bytes, 16)
    ml_off = _align_up(es_off + es_bytes, 16)
    ws = _workspace(q.device, ml_off + es_bytes)
    base = ws.data_ptr()
    return base, base + es_off, base + ml_off


def gather_kv_cache(key_cache: torch.Tensor, value_cache: torch.Tensor, block_table: torch.Tensor,
                    cu_seq_lens: torch.Tensor, out_dtype: torch.dtype, k_scale: torch.Tensor | None = None,
                    v_scale: torch.Tensor | None = None):
    """paged -> dense K/V [num_tokens, kv_heads, head_size]; cu_seq_lens int32 [num_seqs + 1]."""
    nb, kvh, hs_x, block_size, x = key_cache.shape
    head_size = hs_x * x
    num_seqs = cu_seq_lens.numel() - 1
    num_tokens = int(cu_seq_lens[-1].item())
    k_out = torch.empty(num_tokens, kvh, head_size, dtype=out_dtype, device=key_cache.device)
    v_out = torch.empty_like(k_out)
    fn = _lib.sym("paged_attn", "gather_kv_cache", [_vp] * 8 + [_i] * 7 + [_vp, _u, _u])
    ks, vs = _scales(key_cache, k_scale, v_scale)
    fn(key_cache.data_ptr(), value_cache.data_ptr(), k_out.data_ptr(), v_out.data_ptr(), ks, vs,
       block_table.data_ptr(), cu_seq_lens.data_ptr(), num_tokens, num_seqs, block_size, block_table.stride(0), kvh,
       head_size, x, _stream(), _CODE[out_dtype], _CACHE_CODE[key_cache.dtype])
    return k_out, v_out


def update_kv_scales(key: torch.Tensor, value: torch.Tensor, k_scale: torch.Tensor, v_scale: torch.Tensor) -> None:
    """k_scale = max(k_scale, absmax(key) / 240), same for v, in place (backend/scale_update.rs:81-105, update_kvscales.cu)."""
    if key.dtype not in _CODE or value.dtype != key.dtype or key.numel() != value.numel():
        raise ValueError("update_kv_scales: key / value must share dtype (f16 / bf16 / f32) and element count")
    if not (key.is_contiguous() and value.is_contiguous()):
        raise ValueError("update_kv_scales: key / value must be contiguous")
    for t in (k_scale, v_scale):
        if t.dtype != torch.float32 or t.numel() != 1:
            raise ValueError("k_scale / v_scale must be single-element f32 tensors")
    fn = _lib.sym("paged_attn", f"update_kv_scales_{_TAG[key.dtype]}", [_vp, _vp, C.c_long, _vp, _vp, C.c_int64])
    fn(key.data_ptr(), value.data_ptr(), key.numel(), k_scale.data_ptr(), v_scale.data_ptr(), _stream())


def copy_blocks(key_caches: list, value_caches: list, block_mapping: dict) -> None:
    """For every layer copy block src -> each dst (copy-on-write).  Mirrors backend/cache.rs `copy_blocks`."""
    if not key_caches or not block_mapping:
        return
    dev = key_caches[0].device
    pairs = [(s, d) for s, ds in block_mapping.items() for d in (ds if isinstance(ds, (list, tuple)) else [ds])]
    kp = torch.tensor([t.data_ptr() for t in key_caches], dtype=torch.int64, device=dev)
    vp = torch.tensor([t.data_ptr() for t in value_caches], dtype=torch.int64, device=dev)
    bm = torch.tensor(pairs, dtype=torch.int64, device=dev).reshape(-1)
    tag = {2: "bf16" if key_caches[0].dtype == torch.bfloat16 else "f16", 4: "f32", 1: "u8"}[key_caches[0].element_size()]
    fn = _lib.sym("paged_attn", f"copy_blocks_{tag}", [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64])
    fn(kp.data_ptr(), vp.data_ptr(), bm.data_ptr(), len(key_caches), len(pairs), key_caches[0][0].numel(),
       value_caches[0][0].numel(), _stream())
    torch.cuda.current_stream().synchronize()  # keep kp/vp/bm alive until the copy ran


def decode_attention_q8_1(q: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor, block_tables: torch.Tensor,
                          context_lens: torch.Tensor, max_context_len: int, softmax_scale: float) -> tuple[torch.Tensor, int]:
    """MI355X-native decode attention of the fused path (mrs_decode_attention_q8_1_f32_bf16): f32 q [seqs, heads, hd]
    over a bf16 paged cache -> Q8_1 blocks [seqs, stride_blocks * 36] uint8 (o_proj's activation format).
    Returns (blocks, stride_blocks).  Raises ValueError for shapes the kernel refuses."""
    if q.dtype != torch.float32 or key_cache.dtype != torch.bfloat16 or q.dim() != 3:
        raise ValueError("decode_attention_q8_1: f32 query [seqs, heads, hd] over a bf16 cache")
    num_seqs, num_heads, head_size = q.shape
    nb, kvh, hs_x, block_size, x = key_cache.shape
    max_blocks = block_tables.shape[1]
    eff_max = min(max_blocks * block_size, max_context_len)
    L = _lib.load("paged_attn")
    L.mrs_decode_attention_max_splits.argtypes = [_i]
    splits = L.mrs_decode_attention_max_splits(eff_max)
    nq = num_heads * head_size
    stride_blocks = _align_up(nq, 512) // 32
    y = torch.zeros(num_seqs, stride_blocks * 36, dtype=torch.uint8, device=q.device)
    tmp = torch.empty(num_seqs * num_heads * splits * head_size, dtype=torch.float32, device=q.device)
    es = torch.empty(num_seqs * num_heads * splits, dtype=torch.float32, device=q.device)
    ml = torch.empty_like(es)
    fn = _lib.sym("paged_attn", "mrs_decode_attention_q8_1_f32_bf16",
                  [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp] + [_i] * 9 + [_vp], _i)
    rc = fn(y.data_ptr(), stride_blocks, es.data_ptr(), ml.data_ptr(), tmp.data_ptr(), q.data_ptr(), key_cache.data_ptr(),
            value_cache.data_ptr(), kvh, softmax_scale, block_tables.data_ptr(), context_lens.data_ptr(), block_size, eff_max,
            num_seqs, num_heads, head_size, max_blocks, q.stride(0), key_cache.stride(0), key_cache.stride(1), _stream())
    if rc != 0:
        raise ValueError("decode_attention_q8_1: unsupported shape (block_size 32, head_size 64/128)")
    return y, stride_blocks


def prefill_attention(q: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor, block_table: torch.Tensor, start_pos: int,
                      softmax_scale: float) -> torch.Tensor:
    """Causal prompt attention on the matrix cores (mrs_prefill_attention_f32_bf16): f32 q [T, heads, hd] at positions
    start_pos .. start_pos+T-1 over the bf16 paged cache of ONE sequence (block_table [max_blocks] int32; keys 0 .. start_pos+T-1
    already scattered with reshape_and_cache) -> f32 [T, heads, hd].  Role of Sdpa::run_attention in the prompt branch of
    PagedAttention::forward (paged_attention.rs:1413-1475).  Raises ValueError for shapes the kernel refuses."""
    if q.dtype != torch.float32 or key_cache.dtype != torch.bfloat16 or q.dim() != 3 or not q.is_contiguous():
        raise ValueError("prefill_attention: contiguous f32 query [T, heads, hd] over a bf16 cache")
    T, num_heads, head_size = q.shape
    nb, kvh, hs_x, block_size, x = key_cache.shape
    if block_table.dim() != 1 or block_table.numel() * block_size < start_pos + T:
        raise ValueError("prefill_attention: block_table does not cover start_pos + T tokens")
    out = torch.empty_like(q)
    fn = _lib.sym("ext", "mrs_prefill_attention_f32_bf16", [_vp] * 5 + [_i] * 10 + [_f, _vp], _i)
    rc = fn(q.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), block_table.data_ptr(), out.data_ptr(), T, start_pos, num_heads, kvh,
            head_size, block_size, q.stride(0), out.stride(0), key_cache.stride(0), key_cache.stride(1), softmax_scale, _stream())
    if rc != 0:
        raise ValueError("prefill_attention: unsupported shape (bf16 cache, block_size 32, head_size 128, heads/kv_heads in 1/2/4/8)")
    return out
