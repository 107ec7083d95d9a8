"""Prefix caching end to end: the C++ KV cache manager (mistralrs_amd.kv_cache_manager: block pool, chain hashes, block tables) drives the
fused decode kernels of the C++ runner.  A second request that shares two full blocks of prompt with an earlier one re-uses those pages
(no recomputation, pages untouched) and must produce BIT-IDENTICAL logits to the same prompt computed from scratch on fresh pages; after
the pool is put under pressure the cached blocks are evicted and the hit disappears.  Reference flow: PagedAttentionScheduler ->
KVCacheManager::get_computed_blocks / allocate_slots / cache_blocks / free (mistralrs-core/src/paged_attention/kv_cache_manager.rs:129-351)
feeding PagedAttention::forward's block tables and slot mappings (paged_attention.rs:1845-1990).
Runs on the MI355X (`-m gpu`) and, unchanged, on the host emulation (`-m gpu --host-emulation`, ~5 min)."""
import numpy as np
import pytest

from tests.test_llama_runner import Q4KM, _mk

pytestmark = pytest.mark.gpu


def _prompt(n, salt):
    return [(37 * i + 11 * salt + (i * i) % 7) % 512 for i in range(n)]


def _serve(m, mgr, cfg, dev, req, tokens):
    """One request's prompt phase: look up the cached prefix, allocate, point the runner's block table at the manager's blocks, process
    only the uncached tokens (8 per step through the batch-8 decode kernels), publish the newly full blocks."""
    import torch
    from mistralrs_amd.kv_cache_manager import compute_block_hashes
    hashes = compute_block_hashes(tokens, cfg.block_size)
    hit = mgr.get_computed_blocks(hashes, len(tokens))
    new = mgr.allocate_slots(req, len(tokens), hit.block_ids)
    assert new is not None
    m.block_tables[0] = torch.tensor(mgr.get_block_table(req, cfg.max_blocks_per_seq), dtype=torch.int32, device=dev)
    logits = m.prefill_chunked(tokens[hit.num_computed_tokens:], start_pos=hit.num_computed_tokens, chunk=8)
    mgr.cache_blocks(req, hashes, len(tokens))
    return logits, hit, new


def test_prefix_cache_reuse_is_bit_exact(oracle, dev):
    import torch
    from mistralrs_amd.kv_cache_manager import KVCacheManager
    cfg, w, m, cos, sin = _mk(oracle, dev, True, Q4KM(oracle), max_batch=8)
    bs = cfg.block_size  # 32
    shared = _prompt(2 * bs, salt=1)                 # two full blocks of common prefix
    tok_a = shared + _prompt(6, salt=2)
    tok_b = shared + _prompt(10, salt=3)
    mgr = KVCacheManager(m.num_blocks, bs, True, [0])

    _, hit_a, new_a = _serve(m, mgr, cfg, dev, 1, tok_a)
    assert hit_a.num_computed_tokens == 0 and len(new_a) == 3
    blocks_a = mgr.get_block_ids(1)
    mgr.free(1)                                      # the blocks stay findable by hash
    pages = [(k[blocks_a[:2]].clone(), v[blocks_a[:2]].clone()) for k, v in zip(m.key_caches, m.value_caches)]

    logits_b, hit_b, new_b = _serve(m, mgr, cfg, dev, 2, tok_b)
    assert hit_b.num_computed_tokens == 2 * bs and hit_b.block_ids == blocks_a[:2] and len(new_b) == 1
    for (k0, v0), k, v in zip(pages, m.key_caches, m.value_caches):   # the shared pages were read, not rewritten
        assert torch.equal(k[blocks_a[:2]], k0) and torch.equal(v[blocks_a[:2]], v0)

    # the same prompt from scratch: fresh runner, fresh pool, no hit
    _, _, m2, _, _ = _mk(oracle, dev, True, Q4KM(oracle), max_batch=8)
    mgr2 = KVCacheManager(m2.num_blocks, bs, True, [0])
    logits_fresh, hit_f, _ = _serve(m2, mgr2, cfg, dev, 7, tok_b)
    assert hit_f.num_computed_tokens == 0
    assert torch.equal(logits_b, logits_fresh), float((logits_b - logits_fresh).abs().max())

    # pressure: a request that needs every free block evicts the cached prefix (LRU order), after which the hash lookup misses
    from mistralrs_amd.kv_cache_manager import compute_block_hashes
    mgr.free(2)
    big = mgr.allocate_slots(3, mgr.num_free_blocks() * bs, [])
    assert big is not None and mgr.num_free_blocks() == 0
    assert mgr.get_computed_blocks(compute_block_hashes(tok_b, bs), len(tok_b)).num_computed_tokens == 0
    assert mgr.allocate_slots(4, 1, []) is None
