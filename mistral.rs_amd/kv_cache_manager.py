"""Host-side mirror of the reference's paged KV cache management (same names, argument meaning and results):

    BlockPool        mistralrs-core/src/paged_attention/block_pool.rs:267-557
    KVCacheManager   mistralrs-core/src/paged_attention/kv_cache_manager.rs:43-437   (+ ComputedBlocks :19-25)
    hash_block_tokens / compute_block_hashes / compute_new_block_hashes   paged_attention/block_hash.rs:121-306

The logic is native C++ (csrc/host/kv_cache_manager.cpp in libmrs_hip_ext.so, C ABI in include/mrs_hip_ext.h); this file only adapts
types: `None` where the reference returns `None`, lists of ints for block ids.  It produces what the device side consumes: i64 slot
mappings for reshape_and_cache and u32/i32 block tables for paged_attention / the fused decode kernels.  No device work happens here."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

from . import _lib

_PAD_SLOT_ID = -1  # paged_attention/mod.rs:26
_P, _SZ, _U64, _I64 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int64


def _f(name, argtypes, restype=None):
    return _lib.sym("ext", name, argtypes, restype)


def _arr(ctype, values):
    values = list(values)
    return (ctype * max(len(values), 1))(*values), len(values)


def _salt(s: Optional[str]):
    return None if s is None else s.encode("utf-8")


def _gen(g: Optional[bytes]):
    if g is None:
        return None
    if len(g) != 32:
        raise ValueError("adapter generation id must be 32 bytes")
    return (C.c_uint8 * 32)(*g)


def siphash(data: bytes, k0: int = 0, k1: int = 0, c_rounds: int = 1, d_rounds: int = 3) -> int:
    return _f("mrs_kv_siphash", [C.c_char_p, _SZ, _U64, _U64, C.c_int, C.c_int], _U64)(data, len(data), k0, k1, c_rounds, d_rounds)


def hash_block_tokens(parent_hash: Optional[int], block_tokens: Sequence[int], adapter_generation: Optional[bytes] = None,
                      cache_salt: Optional[str] = None) -> int:
    toks, n = _arr(C.c_uint32, block_tokens)
    return _f("mrs_kv_hash_block_tokens", [C.c_int, _U64, _P, _SZ, _P, C.c_char_p], _U64)(
        parent_hash is not None, parent_hash or 0, toks, n, _gen(adapter_generation), _salt(cache_salt))


def compute_new_block_hashes(tokens: Sequence[int], block_size: int, existing_hashes: Sequence[int], adapter_generation: Optional[bytes] = None,
                             cache_salt: Optional[str] = None) -> List[int]:
    toks, n = _arr(C.c_uint32, tokens)
    ex, ne = _arr(_U64, existing_hashes)
    cap = n // block_size if block_size else 0
    out = (_U64 * max(cap, 1))()
    w = _f("mrs_kv_compute_block_hashes", [_P, _SZ, _SZ, _P, _SZ, _P, C.c_char_p, _P, _SZ], _SZ)(
        toks, n, block_size, ex, ne, _gen(adapter_generation), _salt(cache_salt), out, cap)
    return list(out[:w])


def compute_block_hashes(tokens: Sequence[int], block_size: int, adapter_generation: Optional[bytes] = None, cache_salt: Optional[str] = None) -> List[int]:
    return compute_new_block_hashes(tokens, block_size, [], adapter_generation, cache_salt)


class BlockPool:
    def __init__(self, num_gpu_blocks: int, enable_caching: bool, hash_block_size: int, _borrowed=None):
        if _borrowed is not None:
            self._h, self._own = _borrowed, False
            return
        assert num_gpu_blocks > 0, "Must have at least 1 GPU block"
        self._h = _f("mrs_kv_pool_create", [_SZ, C.c_int, _SZ], _P)(num_gpu_blocks, int(enable_caching), hash_block_size)
        self._own = True

    def __del__(self):
        if getattr(self, "_own", False) and self._h:
            _f("mrs_kv_pool_destroy", [_P])(self._h)
            self._h = None

    def null_block_id(self) -> int:
        return _f("mrs_kv_pool_null_block_id", [_P], _SZ)(self._h)

    def num_free_blocks(self) -> int:
        return _f("mrs_kv_pool_num_free_blocks", [_P], _SZ)(self._h)

    def num_gpu_blocks(self) -> int:
        return _f("mrs_kv_pool_num_gpu_blocks", [_P], _SZ)(self._h)

    def usage(self) -> float:
        return _f("mrs_kv_pool_usage", [_P], C.c_double)(self._h)

    def num_cached_blocks(self) -> int:
        return _f("mrs_kv_pool_num_cached_blocks", [_P], _SZ)(self._h)

    def hash_block_size(self) -> int:
        return _f("mrs_kv_pool_hash_block_size", [_P], _SZ)(self._h)

    def caching_enabled(self) -> bool:
        return bool(_f("mrs_kv_pool_caching_enabled", [_P], C.c_int)(self._h))

    def block_ref_cnt(self, block_id: int) -> int:
        r = _f("mrs_kv_pool_block_ref_cnt", [_P, _I64], _I64)(self._h, block_id)
        if r < 0:
            raise IndexError(block_id)
        return r

    def num_block_hashes(self, block_id: int) -> int:
        return _f("mrs_kv_pool_num_block_hashes", [_P, _I64], _I64)(self._h, block_id)

    def get_new_blocks(self, num_blocks: int) -> Optional[List[int]]:
        out = (_I64 * max(num_blocks, 1))()
        r = _f("mrs_kv_pool_get_new_blocks", [_P, _SZ, _P, _SZ], _I64)(self._h, num_blocks, out, num_blocks)
        return None if r < 0 else list(out[:r])

    def free_blocks(self, ordered_block_ids: Sequence[int]) -> None:
        a, n = _arr(_I64, ordered_block_ids)
        if _f("mrs_kv_pool_free_blocks", [_P, _P, _SZ], C.c_int)(self._h, a, n) != 0:
            raise IndexError("block id out of range")

    def touch(self, block_ids: Sequence[int]) -> None:
        a, n = _arr(_I64, block_ids)
        if _f("mrs_kv_pool_touch", [_P, _P, _SZ], C.c_int)(self._h, a, n) != 0:
            raise IndexError("block id out of range")

    def cache_full_blocks(self, block_ids: Sequence[int], block_hashes: Sequence[int], num_cached_blocks: int, num_full_blocks: int,
                          kv_cache_group_id: int) -> None:
        a, n = _arr(_I64, block_ids)
        h, nh = _arr(_U64, block_hashes)
        r = _f("mrs_kv_pool_cache_full_blocks", [_P, _P, _SZ, _P, _SZ, _SZ, _SZ, C.c_uint32], C.c_int)(self._h, a, n, h, nh, num_cached_blocks,
                                                                                                      num_full_blocks, kv_cache_group_id)
        if r == -1:
            raise AssertionError(f"Not enough block hashes ({nh}) for {num_full_blocks} full blocks")
        if r != 0:
            raise IndexError("block id out of range")

    def get_cached_block(self, block_hash: int, kv_cache_group_ids: Sequence[int]) -> Optional[List[int]]:
        g, n = _arr(C.c_uint32, kv_cache_group_ids)
        out = (_I64 * max(n, 1))()
        r = _f("mrs_kv_pool_get_cached_block", [_P, _U64, _P, _SZ, _P], _I64)(self._h, block_hash, g, n, out)
        return None if r < 0 else list(out[:r])

    def reset_prefix_cache(self) -> bool:
        return bool(_f("mrs_kv_pool_reset_prefix_cache", [_P], C.c_int)(self._h))

    def _set_ref_cnt_for_test(self, block_id: int, v: int) -> None:
        _f("mrs_kv_pool_set_ref_cnt_for_test", [_P, _I64, C.c_uint32])(self._h, block_id, v)


@dataclass
class ComputedBlocks:
    block_ids: List[int]
    num_computed_tokens: int


class KVCacheManager:
    def __init__(self, num_gpu_blocks: int, block_size: int, enable_caching: bool, kv_cache_group_ids: Sequence[int]):
        assert num_gpu_blocks > 0, "Must have at least 1 GPU block"
        g, n = _arr(C.c_uint32, kv_cache_group_ids)
        self._h = _f("mrs_kv_manager_create", [_SZ, _SZ, C.c_int, _P, _SZ], _P)(num_gpu_blocks, block_size, int(enable_caching), g, n)
        if not self._h:
            raise ValueError("KVCacheManager: bad sizes")
        self._block_size = block_size

    def __del__(self):
        if getattr(self, "_h", None):
            _f("mrs_kv_manager_destroy", [_P])(self._h)
            self._h = None

    def block_pool(self) -> BlockPool:
        p = BlockPool(0, False, 0, _borrowed=_f("mrs_kv_manager_pool", [_P], _P)(self._h))
        p._keepalive = self
        return p

    def null_block_id(self) -> int:
        return _f("mrs_kv_null_block_id", [_P], _SZ)(self._h)

    def block_size(self) -> int:
        return self._block_size

    def usage(self) -> float:
        return _f("mrs_kv_usage", [_P], C.c_double)(self._h)

    def num_free_blocks(self) -> int:
        return _f("mrs_kv_num_free_blocks", [_P], _SZ)(self._h)

    def num_usable_blocks(self) -> int:
        return _f("mrs_kv_num_usable_blocks", [_P], _SZ)(self._h)

    def num_gpu_blocks(self) -> int:
        return _f("mrs_kv_num_gpu_blocks", [_P], _SZ)(self._h)

    def caching_enabled(self) -> bool:
        return bool(_f("mrs_kv_caching_enabled", [_P], C.c_int)(self._h))

    def get_computed_blocks(self, block_hashes: Sequence[int], num_tokens: int) -> ComputedBlocks:
        h, n = _arr(_U64, block_hashes)
        out = (_I64 * max(n, 1))()
        t = _SZ(0)
        r = _f("mrs_kv_get_computed_blocks", [_P, _P, _SZ, _SZ, _P, _SZ, _P], _I64)(self._h, h, n, num_tokens, out, n, C.byref(t))
        return ComputedBlocks(list(out[:max(r, 0)]), t.value)

    def allocate_slots(self, request_id: int, num_tokens: int, computed_blocks: Sequence[int]) -> Optional[List[int]]:
        c, n = _arr(_I64, computed_blocks)
        cap = -(-num_tokens // self._block_size) + 1
        out = (_I64 * cap)()
        r = _f("mrs_kv_allocate_slots", [_P, _U64, _SZ, _P, _SZ, _P, _SZ], _I64)(self._h, request_id, num_tokens, c, n, out, cap)
        if r == -2:
            raise IndexError("computed block id out of range")
        return None if r < 0 else list(out[:r])

    def free(self, request_id: int) -> None:
        _f("mrs_kv_free", [_P, _U64])(self._h, request_id)

    def trim_request_to_num_tokens(self, request_id: int, num_tokens: int) -> None:
        _f("mrs_kv_trim_request_to_num_tokens", [_P, _U64, _SZ])(self._h, request_id, num_tokens)

    def cache_blocks(self, request_id: int, block_hashes: Sequence[int], num_computed_tokens: int) -> None:
        h, n = _arr(_U64, block_hashes)
        if _f("mrs_kv_cache_blocks", [_P, _U64, _P, _SZ, _SZ], C.c_int)(self._h, request_id, h, n, num_computed_tokens) != 0:
            raise AssertionError("Not enough block hashes for the full blocks")

    def get_block_ids(self, request_id: int) -> Optional[List[int]]:
        n = self.num_blocks_for_request(request_id)
        out = (_I64 * max(n, 1))()
        r = _f("mrs_kv_get_block_ids", [_P, _U64, _P, _SZ], _I64)(self._h, request_id, out, n)
        return None if r < 0 else list(out[:r])

    def num_blocks_for_request(self, request_id: int) -> int:
        return _f("mrs_kv_num_blocks_for_request", [_P, _U64], _SZ)(self._h, request_id)

    def has_request(self, request_id: int) -> bool:
        return bool(_f("mrs_kv_has_request", [_P, _U64], C.c_int)(self._h, request_id))

    def num_cached_blocks(self, request_id: int) -> int:
        return _f("mrs_kv_num_cached_blocks_for_request", [_P, _U64], _SZ)(self._h, request_id)

    def reset_prefix_cache(self) -> bool:
        return bool(_f("mrs_kv_reset_prefix_cache", [_P], C.c_int)(self._h))

    def get_slot_mapping(self, request_id: int, start_token: int, num_tokens: int) -> Optional[List[int]]:
        out = (_I64 * max(num_tokens, 1))()
        r = _f("mrs_kv_get_slot_mapping", [_P, _U64, _SZ, _SZ, _P], C.c_int)(self._h, request_id, start_token, num_tokens, out)
        return None if r != 0 else list(out[:num_tokens])

    def get_block_table(self, request_id: int, max_blocks: int) -> Optional[List[int]]:
        out = (C.c_int32 * max(max_blocks, 1))()
        r = _f("mrs_kv_get_block_table", [_P, _U64, _SZ, _P], C.c_int)(self._h, request_id, max_blocks, out)
        return None if r != 0 else list(out[:max_blocks])
