"""CPU (host logic): the C++ paged-KV block pool / manager (csrc/host/kv_cache_manager.cpp behind mistralrs_amd.kv_cache_manager) held to the
reference's OWN unit tests, restated one for one:
    mistralrs-core/src/paged_attention/block_pool.rs:558-787        (mod tests of BlockPool)
    mistralrs-core/src/paged_attention/kv_cache_manager.rs:439-680  (mod tests of KVCacheManager)
    mistralrs-core/src/paged_attention/block_hash.rs:309-441        (hash properties; multimodal keys are out of scope)
plus the SipHash reference vector that pins the hash core, and an LRU-order property the reference documents but does not test."""
import pytest

from mistralrs_amd.kv_cache_manager import (BlockPool, KVCacheManager, compute_block_hashes, compute_new_block_hashes, hash_block_tokens,
                                            siphash)


# ------------------------------------------------------------------------------------------------ hashing
def test_siphash_2_4_reference_vector():
    """SipHash paper, Appendix A: key 00..0f, message 00..0e -> a129ca6149be45e5 (the 1-3 variant used for block hashes shares this core)."""
    key = bytes(range(16))
    k0, k1 = int.from_bytes(key[:8], "little"), int.from_bytes(key[8:], "little")
    assert siphash(bytes(range(15)), k0, k1, 2, 4) == 0xA129CA6149BE45E5
    # streaming: block hashes feed the same bytes in pieces (parent, length prefix, tokens)
    toks = [1, 2, 3, 4]
    stream = (0).to_bytes(8, "little") + len(toks).to_bytes(8, "little") + b"".join(t.to_bytes(4, "little") for t in toks)
    assert hash_block_tokens(None, toks) == siphash(stream, 0, 0, 1, 3)


def test_hash_consistency():
    assert hash_block_tokens(None, [1, 2, 3, 4]) == hash_block_tokens(None, [1, 2, 3, 4])


def test_different_tokens_different_hash():
    assert hash_block_tokens(None, [1, 2, 3, 4]) != hash_block_tokens(None, [1, 2, 3, 5])


def test_chain_hashing():
    h1 = hash_block_tokens(None, [5, 6, 7, 8])
    assert hash_block_tokens(h1, [9, 10, 11, 12]) != hash_block_tokens(None, [9, 10, 11, 12])


def test_extra_keys_affect_hash():
    base = hash_block_tokens(None, [1, 2, 3, 4])
    assert hash_block_tokens(None, [1, 2, 3, 4], cache_salt="tenant-a") != base
    assert hash_block_tokens(None, [1, 2, 3, 4], cache_salt="tenant-a") != hash_block_tokens(None, [1, 2, 3, 4], cache_salt="tenant-b")


def test_compute_block_hashes():
    tokens = list(range(16))
    hashes = compute_block_hashes(tokens, 4)
    assert len(hashes) == 4
    assert hashes == compute_block_hashes(tokens, 4)
    assert hashes[1] == hash_block_tokens(hashes[0], tokens[4:8])


def test_adapter_generations_cannot_cross_hit_block_cache():
    tokens = [1, 2, 3, 4, 5, 6, 7, 8]
    base = compute_block_hashes(tokens, 4)
    a = compute_block_hashes(tokens, 4, adapter_generation=bytes([1]) * 32)
    b = compute_block_hashes(tokens, 4, adapter_generation=bytes([2]) * 32)
    assert base != a and base != b and a != b


def test_compute_block_hashes_partial_block_ignored():
    assert len(compute_block_hashes(list(range(10)), 4)) == 2


def test_incremental_hashing():
    tokens = list(range(16))
    all_hashes = compute_block_hashes(tokens, 4)
    initial = compute_block_hashes(tokens[:8], 4)
    assert len(initial) == 2
    new = compute_new_block_hashes(tokens, 4, initial)
    assert len(new) == 2
    assert initial + new == all_hashes
    assert compute_new_block_hashes(tokens[:8], 4, initial) == []


# ------------------------------------------------------------------------------------------------ BlockPool (block_pool.rs tests)
def _h3():
    h0 = hash_block_tokens(None, [1, 2, 3, 4])
    h1 = hash_block_tokens(h0, [5, 6, 7, 8])
    h2 = hash_block_tokens(h1, [9, 10, 11, 12])
    return h0, h1, h2


def test_pool_basic_allocation():
    pool = BlockPool(4, False, 16)
    assert pool.num_free_blocks() == 3  # 4 blocks, 1 is null
    blocks = pool.get_new_blocks(2)
    assert len(blocks) == 2 and pool.num_free_blocks() == 1
    assert all(pool.block_ref_cnt(b) == 1 for b in blocks)


def test_pool_free_returns_to_pool():
    pool = BlockPool(4, False, 16)
    blocks = pool.get_new_blocks(3)
    assert pool.num_free_blocks() == 0
    pool.free_blocks(blocks)
    assert pool.num_free_blocks() == 3
    assert all(pool.block_ref_cnt(b) == 0 for b in blocks)


def test_pool_allocation_fails_when_exhausted():
    pool = BlockPool(2, False, 16)
    assert pool.num_free_blocks() == 1
    assert pool.get_new_blocks(1) is not None
    assert pool.num_free_blocks() == 0
    assert pool.get_new_blocks(1) is None


def test_pool_prefix_cache_basic():
    pool = BlockPool(8, True, 4)
    block_ids = pool.get_new_blocks(3)
    h0, h1, h2 = _h3()
    pool.cache_full_blocks(block_ids, [h0, h1, h2], 0, 3, 0)
    assert pool.num_cached_blocks() == 3
    assert pool.get_cached_block(h0, [0]) == [block_ids[0]]


def test_pool_prefix_cache_reuse_after_free():
    pool = BlockPool(8, True, 4)
    block_ids = pool.get_new_blocks(2)
    h0, h1, _ = _h3()
    pool.cache_full_blocks(block_ids, [h0, h1], 0, 2, 0)
    pool.free_blocks(block_ids)
    assert pool.num_free_blocks() == 7
    cached = pool.get_cached_block(h0, [0])  # freed blocks keep their hash
    assert cached is not None
    pool.touch(cached)
    assert pool.block_ref_cnt(cached[0]) == 1
    assert pool.num_free_blocks() == 6


def test_pool_eviction_on_reallocation():
    pool = BlockPool(4, True, 4)
    block_ids = pool.get_new_blocks(3)
    h0 = hash_block_tokens(None, [1, 2, 3, 4])
    pool.cache_full_blocks(block_ids, [h0, h0, h0], 0, 1, 0)
    pool.free_blocks(block_ids)
    assert len(pool.get_new_blocks(3)) == 3
    assert pool.get_cached_block(h0, [0]) is None  # the block that held h0 was handed out again: evicted
    assert pool.num_cached_blocks() == 0


def test_pool_touch_ref_cnt_management():
    pool = BlockPool(8, True, 4)
    block_ids = pool.get_new_blocks(1)
    assert pool.block_ref_cnt(block_ids[0]) == 1
    pool.touch(block_ids)
    assert pool.block_ref_cnt(block_ids[0]) == 2
    pool.free_blocks(block_ids)
    assert pool.block_ref_cnt(block_ids[0]) == 1
    free_before = pool.num_free_blocks()
    pool.free_blocks(block_ids)
    assert pool.block_ref_cnt(block_ids[0]) == 0 and pool.num_free_blocks() == free_before + 1


def test_pool_null_block_never_freed():
    pool = BlockPool(4, False, 16)
    null_id = pool.null_block_id()
    pool._set_ref_cnt_for_test(null_id, 1)
    free_before = pool.num_free_blocks()
    pool.free_blocks([null_id])
    assert pool.block_ref_cnt(null_id) == 0 and pool.num_free_blocks() == free_before  # decremented, never listed
    assert null_id not in pool.get_new_blocks(3)


def test_pool_usage():
    pool = BlockPool(4, False, 16)
    assert pool.usage() < 0.01
    pool.get_new_blocks(3)
    assert abs(pool.usage() - 1.0) < 0.01


def test_pool_get_cached_block_multiple_groups():
    pool = BlockPool(8, True, 4)
    g0, g1 = pool.get_new_blocks(1), pool.get_new_blocks(1)
    h0 = hash_block_tokens(None, [1, 2, 3, 4])
    pool.cache_full_blocks(g0, [h0], 0, 1, 0)
    pool.cache_full_blocks(g1, [h0], 0, 1, 1)
    assert pool.get_cached_block(h0, [0, 1]) == [g0[0], g1[0]]
    assert pool.get_cached_block(h0, [0, 2]) is None


def test_pool_same_block_can_cache_multiple_groups():
    pool = BlockPool(8, True, 4)
    ids = pool.get_new_blocks(1)
    h0 = hash_block_tokens(None, [1, 2, 3, 4])
    pool.cache_full_blocks(ids, [h0], 0, 1, 0)
    pool.cache_full_blocks(ids, [h0], 0, 1, 1)
    assert pool.get_cached_block(h0, [0, 1]) == [ids[0], ids[0]]
    assert pool.num_block_hashes(ids[0]) == 2
    pool.free_blocks(ids)
    pool.get_new_blocks(pool.num_free_blocks())
    assert pool.get_cached_block(h0, [0]) is None and pool.get_cached_block(h0, [1]) is None


def test_pool_reset_prefix_cache():
    pool = BlockPool(4, True, 4)
    ids = pool.get_new_blocks(2)
    h0 = hash_block_tokens(None, [1, 2, 3, 4])
    pool.cache_full_blocks(ids, [h0, h0], 0, 1, 0)
    assert not pool.reset_prefix_cache()  # blocks in use
    pool.free_blocks(ids)
    assert pool.reset_prefix_cache()
    assert pool.num_cached_blocks() == 0


def test_pool_cache_full_blocks_needs_enough_hashes():
    pool = BlockPool(8, True, 4)
    ids = pool.get_new_blocks(3)
    with pytest.raises(AssertionError, match="Not enough block hashes"):
        pool.cache_full_blocks(ids, [1, 2], 0, 3, 0)


def test_pool_free_order_is_eviction_order():
    """free_blocks appends in the given order and get_new_blocks pops from the front: blocks freed first are reused (evicted) first --
    the reason KVCacheManager::free hands the request's blocks over in REVERSE (tail blocks of a sequence go before its shared prefix)."""
    pool = BlockPool(6, True, 4)
    ids = pool.get_new_blocks(5)
    pool.free_blocks([ids[3], ids[1]])
    assert pool.get_new_blocks(2) == [ids[3], ids[1]]


# ------------------------------------------------------------------------------------------------ KVCacheManager (kv_cache_manager.rs tests)
def test_basic_allocation():
    mgr = KVCacheManager(16, 4, False, [0])
    assert len(mgr.allocate_slots(1, 10, [])) == 3  # ceil(10 / 4)
    assert mgr.num_blocks_for_request(1) == 3


def test_running_request_extends():
    mgr = KVCacheManager(16, 4, False, [0])
    mgr.allocate_slots(1, 8, [])
    assert mgr.num_blocks_for_request(1) == 2
    assert len(mgr.allocate_slots(1, 12, [])) == 1
    assert mgr.num_blocks_for_request(1) == 3
    assert mgr.allocate_slots(1, 12, []) == []  # nothing more to allocate


def test_allocation_fails_when_full():
    mgr = KVCacheManager(4, 4, False, [0])
    mgr.allocate_slots(1, 12, [])  # takes all 3
    assert mgr.allocate_slots(2, 4, []) is None
    assert not mgr.has_request(2)


def test_free_returns_blocks():
    mgr = KVCacheManager(8, 4, False, [0])
    mgr.allocate_slots(1, 12, [])
    assert mgr.num_free_blocks() == 4
    mgr.free(1)
    assert mgr.num_free_blocks() == 7
    assert not mgr.has_request(1)


def _cached_two_blocks(groups=(0,)):
    mgr = KVCacheManager(16, 4, True, list(groups))
    hashes = compute_block_hashes(list(range(1, 9)), 4)
    mgr.allocate_slots(1, 8, [])
    mgr.cache_blocks(1, hashes, 8)
    mgr.free(1)  # blocks stay in the cache
    return mgr, hashes


def test_prefix_cache_hit():
    mgr, hashes = _cached_two_blocks()
    computed = mgr.get_computed_blocks(hashes, 12)
    assert computed.num_computed_tokens == 8 and len(computed.block_ids) == 2
    assert len(mgr.allocate_slots(2, 12, computed.block_ids)) == 1  # only one new block
    assert mgr.num_blocks_for_request(2) == 3
    assert mgr.get_block_ids(2)[:2] == computed.block_ids


def test_prefix_cache_partial_hit():
    mgr, _ = _cached_two_blocks()
    hashes_ext = compute_block_hashes(list(range(1, 13)), 4)
    assert mgr.get_computed_blocks(hashes_ext, 12).num_computed_tokens == 8


def test_prefix_cache_hit_with_group_aliases():
    mgr, hashes = _cached_two_blocks(groups=(0, 1))
    computed = mgr.get_computed_blocks(hashes, 12)
    assert computed.num_computed_tokens == 8 and len(computed.block_ids) == 2


def test_cache_blocks_incremental():
    mgr = KVCacheManager(16, 4, True, [0])
    hashes = compute_block_hashes(list(range(1, 17)), 4)
    mgr.allocate_slots(1, 16, [])
    mgr.cache_blocks(1, hashes, 8)
    assert mgr.num_cached_blocks(1) == 2
    mgr.cache_blocks(1, hashes, 16)
    assert mgr.num_cached_blocks(1) == 4
    mgr.cache_blocks(1, hashes, 400)  # token counts may run ahead of the allocation: clamped
    assert mgr.num_cached_blocks(1) == 4


def test_slot_mapping():
    mgr = KVCacheManager(16, 4, False, [0])
    mgr.allocate_slots(1, 8, [])
    block_ids = mgr.get_block_ids(1)
    slots = mgr.get_slot_mapping(1, 0, 8)
    assert slots == [block_ids[0] * 4 + i for i in range(4)] + [block_ids[1] * 4 + i for i in range(4)]
    assert mgr.get_slot_mapping(1, 6, 4) == [block_ids[1] * 4 + 2, block_ids[1] * 4 + 3, -1, -1]  # _PAD_SLOT_ID past the allocation
    assert mgr.get_slot_mapping(9, 0, 1) is None


def test_slot_mapping_skip_cached():
    mgr, hashes = _cached_two_blocks()
    computed = mgr.get_computed_blocks(hashes, 12)
    mgr.allocate_slots(2, 12, computed.block_ids)
    slots = mgr.get_slot_mapping(2, 8, 4)
    assert len(slots) == 4 and slots == [mgr.get_block_ids(2)[2] * 4 + i for i in range(4)]


def test_block_table():
    mgr = KVCacheManager(16, 4, False, [0])
    mgr.allocate_slots(1, 8, [])
    table = mgr.get_block_table(1, 5)
    assert len(table) == 5 and table[:2] == mgr.get_block_ids(1) and table[2:] == [0, 0, 0]
    assert mgr.get_block_table(7, 5) is None


def test_trim_request_allocation():
    mgr = KVCacheManager(8, 4, False, [0])
    mgr.allocate_slots(1, 12, [])
    assert mgr.num_blocks_for_request(1) == 3 and mgr.num_free_blocks() == 4
    mgr.trim_request_to_num_tokens(1, 8)
    assert mgr.num_blocks_for_request(1) == 2 and mgr.num_free_blocks() == 5


def test_trim_clamps_cached_blocks():
    mgr = KVCacheManager(16, 4, True, [0])
    hashes = compute_block_hashes(list(range(1, 17)), 4)
    mgr.allocate_slots(1, 16, [])
    mgr.cache_blocks(1, hashes, 16)
    assert mgr.num_cached_blocks(1) == 4
    mgr.trim_request_to_num_tokens(1, 8)
    assert mgr.num_blocks_for_request(1) == 2 and mgr.num_cached_blocks(1) == 2


def test_get_computed_blocks_caps_at_prompt_minus_one():
    mgr, hashes = _cached_two_blocks()
    computed = mgr.get_computed_blocks(hashes, 8)  # max hit = 7 tokens -> one block
    assert computed.num_computed_tokens == 4 and len(computed.block_ids) == 1


def test_reset_prefix_cache():
    mgr = KVCacheManager(8, 4, True, [0])
    hashes = compute_block_hashes([1, 2, 3, 4], 4)
    mgr.allocate_slots(1, 4, [])
    mgr.cache_blocks(1, hashes, 4)
    assert not mgr.reset_prefix_cache()
    mgr.free(1)
    assert mgr.reset_prefix_cache()
    assert mgr.get_computed_blocks(hashes, 8).num_computed_tokens == 0


def test_shared_prefix_two_live_requests_and_capacity_accounting():
    """Two live requests share cached prefix blocks (ref_cnt 2); the capacity check counts cached blocks that sit in the free list
    (kv_cache_manager.rs:214-229): a hit on an evictable block still consumes a free-list entry."""
    mgr, hashes = _cached_two_blocks()
    pool = mgr.block_pool()
    c = mgr.get_computed_blocks(hashes, 12)
    mgr.allocate_slots(2, 12, c.block_ids)
    mgr.allocate_slots(3, 12, c.block_ids)
    assert [pool.block_ref_cnt(b) for b in c.block_ids] == [2, 2]
    mgr.free(2)
    assert [pool.block_ref_cnt(b) for b in c.block_ids] == [1, 1]
    mgr.free(3)
    small = KVCacheManager(4, 4, True, [0])  # 3 usable blocks
    h = compute_block_hashes(list(range(1, 9)), 4)
    small.allocate_slots(1, 8, [])
    small.cache_blocks(1, h, 8)
    small.free(1)
    hit = small.get_computed_blocks(h, 16)
    assert len(hit.block_ids) == 2
    assert small.allocate_slots(2, 16, hit.block_ids) is None  # needs 2 new + 2 evictable hits > 3 free
    assert small.allocate_slots(2, 12, hit.block_ids) is not None  # 1 new + 2 evictable = 3
