// kv_cache_manager.cpp -- host side of the paged KV cache: block pool with prefix caching + per-request block tracking (C++, no device
// code; the tables it hands out -- block tables, slot mappings -- are what reshape_and_cache / paged_attention / the fused decode kernels
// consume).  Mirrors, behaviour for behaviour (names, argument meaning, None / false results):
//   mistralrs-core/src/paged_attention/block_pool.rs:26-557      KVCacheBlock, FreeKVCacheBlockQueue, BlockHashToBlockMap, BlockPool
//   mistralrs-core/src/paged_attention/kv_cache_manager.rs:19-437 ComputedBlocks, KVCacheManager
//   mistralrs-core/src/paged_attention/block_hash.rs:121-306     hash_block_tokens, compute_block_hashes, compute_new_block_hashes
// (a port of vLLM v1's KVCacheManager + FullAttentionManager, as the reference says of itself).  SURVEY.md 8(f).3.
//
// Block hashes: the reference feeds (parent hash | 0, token slice, extra keys) through Rust's `DefaultHasher`, which is SipHash-1-3 with a
// zero key over the byte stream u64 LE (parent or seed 0), usize LE length prefix, the u32 tokens LE, then per extra key the enum
// discriminant as isize LE followed by the payload (AdapterGeneration: 32-byte array hashed as a slice = length prefix + bytes; CacheSalt:
// the UTF-8 bytes + 0xff).  Restated here from the published SipHash definition (checked against the SipHash-2-4 reference vector in
// tests/test_kv_cache_manager.py); hash values never leave the process, only equality matters.  Multimodal extra keys are out of scope.
//
// Plain C ABI (include/mrs_hip_ext.h, "paged KV cache manager"): ids are int64_t, -1 = the reference's None.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace mrs_host {

// ------------------------------------------------------------------------------------------------ SipHash (c rounds per word, d final)
struct SipHasher {
  uint64_t v0, v1, v2, v3, tail = 0;
  size_t ntail = 0, length = 0;
  int c, d;
  SipHasher(uint64_t k0, uint64_t k1, int c_rounds, int d_rounds) : c(c_rounds), d(d_rounds) {
    v0 = k0 ^ 0x736f6d6570736575ull; v1 = k1 ^ 0x646f72616e646f6dull; v2 = k0 ^ 0x6c7967656e657261ull; v3 = k1 ^ 0x7465646279746573ull;
  }
  static uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
  void round() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  }
  void word(uint64_t m) { v3 ^= m; for (int i = 0; i < c; ++i) round(); v0 ^= m; }
  void write(const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    length += n;
    for (size_t i = 0; i < n; ++i) {
      tail |= (uint64_t)b[i] << (8 * ntail);
      if (++ntail == 8) { word(tail); tail = 0; ntail = 0; }
    }
  }
  void write_u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * i)); write(b, 8); }
  uint64_t finish() const {
    SipHasher s = *this;
    const uint64_t b = ((uint64_t)(length & 0xff) << 56) | tail;
    s.v3 ^= b; for (int i = 0; i < s.c; ++i) s.round(); s.v0 ^= b;
    s.v2 ^= 0xff;
    for (int i = 0; i < s.d; ++i) s.round();
    return s.v0 ^ s.v1 ^ s.v2 ^ s.v3;
  }
};

struct ExtraKeys {  // extra_keys_base of compute_block_hashes (block_hash.rs:236-241), in this order
  const uint8_t *adapter_generation = nullptr;  // 32 bytes: ExtraHashKey::AdapterGeneration (discriminant 1)
  const char *cache_salt = nullptr;             // ExtraHashKey::CacheSalt (discriminant 2)
};

// hash_block_tokens (block_hash.rs:121-146)
static uint64_t hash_block_tokens(bool has_parent, uint64_t parent, const uint32_t *tokens, size_t n, const ExtraKeys &extra) {
  SipHasher h(0, 0, 1, 3);
  h.write_u64(has_parent ? parent : 0);  // NONE_HASH_SEED = 0
  h.write_u64((uint64_t)n);
  for (size_t i = 0; i < n; ++i) { const uint32_t t = tokens[i]; uint8_t b[4] = {(uint8_t)t, (uint8_t)(t >> 8), (uint8_t)(t >> 16), (uint8_t)(t >> 24)}; h.write(b, 4); }
  if (extra.adapter_generation) { h.write_u64(1); h.write_u64(32); h.write(extra.adapter_generation, 32); }
  if (extra.cache_salt) { h.write_u64(2); h.write(extra.cache_salt, strlen(extra.cache_salt)); const uint8_t ff = 0xff; h.write(&ff, 1); }
  return h.finish();
}

// ------------------------------------------------------------------------------------------------ block pool (block_pool.rs)
static constexpr size_t NO_LINK = (size_t)-1;
struct HashWithGroup {
  uint64_t hash; uint32_t group;
  bool operator==(const HashWithGroup &o) const { return hash == o.hash && group == o.group; }
};
struct HashWithGroupHasher { size_t operator()(const HashWithGroup &k) const { return (size_t)(k.hash * 0x9e3779b97f4a7c15ull ^ k.group); } };

struct KVCacheBlock {
  uint32_t ref_cnt = 0;
  std::vector<HashWithGroup> block_hashes;
  size_t prev_free = NO_LINK, next_free = NO_LINK;
  bool is_null = false;
};

class BlockPool {
 public:
  BlockPool(size_t num_gpu_blocks, bool enable_caching, size_t hash_block_size)
      : blocks_(num_gpu_blocks + 2), enable_caching_(enable_caching), num_gpu_blocks_(num_gpu_blocks), hash_block_size_(hash_block_size) {
    // [0, n) real blocks, [n] fake head, [n + 1] fake tail; every real block starts in the free list in id order
    head_ = num_gpu_blocks; tail_ = num_gpu_blocks + 1;
    size_t prev = head_;
    for (size_t id = 0; id < num_gpu_blocks; ++id) { blocks_[id].prev_free = prev; blocks_[prev].next_free = id; prev = id; }
    blocks_[prev].next_free = tail_; blocks_[tail_].prev_free = prev;
    num_free_ = num_gpu_blocks;
    null_block_id_ = popleft();  // the null block: a placeholder that is never freed
    blocks_[null_block_id_].is_null = true;
  }
  size_t null_block_id() const { return null_block_id_; }
  size_t num_free_blocks() const { return num_free_; }
  size_t num_gpu_blocks() const { return num_gpu_blocks_; }
  double usage() const {
    const size_t total = num_gpu_blocks_ - 1;
    return total == 0 ? 0.0 : 1.0 - (double)num_free_ / (double)total;
  }
  // one block per group for `hash`, or false if any group misses (get_cached_block :355-370)
  bool get_cached_block(uint64_t hash, const uint32_t *groups, size_t n_groups, std::vector<size_t> &out) const {
    out.clear();
    for (size_t i = 0; i < n_groups; ++i) {
      auto it = cache_.find(HashWithGroup{hash, groups[i]});
      if (it == cache_.end() || it->second.empty()) return false;
      out.push_back(it->second.front());
    }
    return true;
  }
  void touch(const size_t *ids, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      KVCacheBlock &b = blocks_[ids[i]];
      if (b.ref_cnt == 0 && !b.is_null) remove(ids[i]);  // in the free list as an eviction candidate: take it back
      b.ref_cnt += 1;
    }
  }
  void free_blocks(const size_t *ids, size_t n) {  // `ids` ordered by eviction priority (first = evicted first)
    for (size_t i = 0; i < n; ++i) if (blocks_[ids[i]].ref_cnt > 0) blocks_[ids[i]].ref_cnt -= 1;  // saturating
    for (size_t i = 0; i < n; ++i) if (blocks_[ids[i]].ref_cnt == 0 && !blocks_[ids[i]].is_null && !in_free_list(ids[i])) append(ids[i]);
  }
  bool get_new_blocks(size_t n, std::vector<size_t> &out) {
    out.clear();
    if (n > num_free_) return false;
    for (size_t i = 0; i < n; ++i) {
      const size_t id = popleft();
      if (enable_caching_) maybe_evict_cached_block(id);
      blocks_[id].ref_cnt = 1;
      out.push_back(id);
    }
    return true;
  }
  // returns false when `hashes` is too short (the reference asserts)
  bool cache_full_blocks(const size_t *block_ids, const uint64_t *hashes, size_t n_hashes, size_t num_cached, size_t num_full, uint32_t group) {
    if (!enable_caching_ || num_cached >= num_full) return true;
    if (n_hashes < num_full) return false;
    for (size_t idx = num_cached; idx < num_full; ++idx) {
      KVCacheBlock &b = blocks_[block_ids[idx]];
      if (b.is_null) continue;
      const HashWithGroup key{hashes[idx], group};
      if (std::find(b.block_hashes.begin(), b.block_hashes.end(), key) != b.block_hashes.end()) continue;
      b.block_hashes.push_back(key);
      cache_[key].push_back(block_ids[idx]);
    }
    return true;
  }
  bool reset_prefix_cache() {
    if (num_gpu_blocks_ - num_free_ != 1) return false;  // only the null block may be in use
    cache_.clear();
    for (auto &b : blocks_) b.block_hashes.clear();
    return true;
  }
  size_t num_cached_blocks() const { return cache_.size(); }
  size_t hash_block_size() const { return hash_block_size_; }
  bool caching_enabled() const { return enable_caching_; }
  uint32_t block_ref_cnt(size_t id) const { return blocks_[id].ref_cnt; }
  const std::vector<HashWithGroup> &block_hashes(size_t id) const { return blocks_[id].block_hashes; }
  bool valid_id(int64_t id) const { return id >= 0 && (size_t)id < num_gpu_blocks_; }
  void set_ref_cnt_for_test(size_t id, uint32_t v) { blocks_[id].ref_cnt = v; }

 private:
  bool in_free_list(size_t id) const { return blocks_[id].prev_free != NO_LINK || blocks_[id].next_free != NO_LINK; }
  size_t popleft() {
    const size_t first = blocks_[head_].next_free, next = blocks_[first].next_free;
    blocks_[head_].next_free = next; blocks_[next].prev_free = head_;
    blocks_[first].prev_free = blocks_[first].next_free = NO_LINK;
    --num_free_;
    return first;
  }
  void remove(size_t id) {
    const size_t p = blocks_[id].prev_free, n = blocks_[id].next_free;
    blocks_[p].next_free = n; blocks_[n].prev_free = p;
    blocks_[id].prev_free = blocks_[id].next_free = NO_LINK;
    --num_free_;
  }
  void append(size_t id) {
    const size_t last = blocks_[tail_].prev_free;
    blocks_[last].next_free = id; blocks_[id].prev_free = last; blocks_[id].next_free = tail_; blocks_[tail_].prev_free = id;
    ++num_free_;
  }
  void maybe_evict_cached_block(size_t id) {
    std::vector<HashWithGroup> hs;
    hs.swap(blocks_[id].block_hashes);
    for (const auto &h : hs) {
      auto it = cache_.find(h);
      if (it == cache_.end()) continue;
      auto &v = it->second;
      v.erase(std::remove(v.begin(), v.end(), id), v.end());
      if (v.empty()) cache_.erase(it);
    }
  }
  std::vector<KVCacheBlock> blocks_;
  std::unordered_map<HashWithGroup, std::vector<size_t>, HashWithGroupHasher> cache_;  // BlockHashToBlockMap: hash -> blocks holding it
  bool enable_caching_;
  size_t num_gpu_blocks_, hash_block_size_, head_ = 0, tail_ = 0, num_free_ = 0, null_block_id_ = 0;
};

// ------------------------------------------------------------------------------------------------ manager (kv_cache_manager.rs)
class KVCacheManager {
 public:
  KVCacheManager(size_t num_gpu_blocks, size_t block_size, bool enable_caching, const uint32_t *groups, size_t n_groups)
      : pool(num_gpu_blocks, enable_caching, block_size), block_size_(block_size), enable_caching_(enable_caching), groups_(groups, groups + n_groups) {}
  BlockPool pool;
  size_t block_size() const { return block_size_; }
  bool caching_enabled() const { return enable_caching_; }

  // longest cached prefix, at most num_tokens - 1 tokens (the last token is recomputed for its logits)  (:129-180)
  size_t get_computed_blocks(const uint64_t *hashes, size_t n_hashes, size_t num_tokens, std::vector<size_t> &ids) const {
    ids.clear();
    if (!enable_caching_ || n_hashes == 0) return 0;
    const size_t max_blocks = (num_tokens == 0 ? 0 : num_tokens - 1) / block_size_;
    std::vector<size_t> hit;
    for (size_t i = 0; i < n_hashes && i < max_blocks; ++i) {
      if (!pool.get_cached_block(hashes[i], groups_.data(), groups_.size(), hit) || hit.empty()) break;
      bool same = true;
      for (size_t id : hit) same = same && id == hit[0];
      if (!same) break;
      ids.push_back(hit[0]);
    }
    return ids.size() * block_size_;
  }
  // :188-267.  false = not enough free blocks (None)
  bool allocate_slots(uint64_t req, size_t num_tokens, const size_t *computed, size_t n_computed, std::vector<size_t> &new_ids) {
    new_ids.clear();
    const size_t required = (num_tokens + block_size_ - 1) / block_size_;
    auto it = reqs_.find(req);
    if (it != reqs_.end()) {  // running request: only the additional blocks
      const size_t have = it->second.block_ids.size();
      if (required <= have) return true;
      if (!pool.get_new_blocks(required - have, new_ids)) return false;
      it->second.block_ids.insert(it->second.block_ids.end(), new_ids.begin(), new_ids.end());
      return true;
    }
    const size_t n_new = required > n_computed ? required - n_computed : 0;
    size_t evictable = 0;  // computed blocks that sit in the free list: touching them shrinks it
    if (enable_caching_)
      for (size_t i = 0; i < n_computed; ++i) evictable += pool.block_ref_cnt(computed[i]) == 0;
    if (n_new + evictable > pool.num_free_blocks()) return false;
    if (n_computed && enable_caching_) pool.touch(computed, n_computed);
    if (n_new) pool.get_new_blocks(n_new, new_ids);
    Request r;
    r.block_ids.assign(computed, computed + n_computed);
    r.block_ids.insert(r.block_ids.end(), new_ids.begin(), new_ids.end());
    r.num_cached_blocks = n_computed;
    reqs_.emplace(req, std::move(r));
    return true;
  }
  void free(uint64_t req) {
    auto it = reqs_.find(req);
    if (it == reqs_.end()) return;
    std::vector<size_t> rev(it->second.block_ids.rbegin(), it->second.block_ids.rend());  // tail blocks are evicted first
    reqs_.erase(it);
    pool.free_blocks(rev.data(), rev.size());
  }
  void trim_request_to_num_tokens(uint64_t req, size_t num_tokens) {
    auto it = reqs_.find(req);
    if (it == reqs_.end()) return;
    Request &r = it->second;
    const size_t required = (num_tokens + block_size_ - 1) / block_size_;
    std::vector<size_t> removed;
    if (required < r.block_ids.size()) {
      removed.assign(r.block_ids.rbegin(), r.block_ids.rend() - (ptrdiff_t)required);  // reversed tail
      r.block_ids.resize(required);
    }
    r.num_cached_blocks = std::min(r.num_cached_blocks, r.block_ids.size());
    if (!removed.empty()) pool.free_blocks(removed.data(), removed.size());
  }
  bool cache_blocks(uint64_t req, const uint64_t *hashes, size_t n_hashes, size_t num_computed_tokens) {
    if (!enable_caching_) return true;
    auto it = reqs_.find(req);
    if (it == reqs_.end()) return true;
    Request &r = it->second;
    const size_t full = std::min(num_computed_tokens / block_size_, r.block_ids.size());
    if (r.num_cached_blocks >= full) return true;
    for (uint32_t g : groups_)
      if (!pool.cache_full_blocks(r.block_ids.data(), hashes, n_hashes, r.num_cached_blocks, full, g)) return false;
    r.num_cached_blocks = full;
    return true;
  }
  struct Request { std::vector<size_t> block_ids; size_t num_cached_blocks = 0; };
  const Request *request(uint64_t req) const { auto it = reqs_.find(req); return it == reqs_.end() ? nullptr : &it->second; }

 private:
  size_t block_size_;
  bool enable_caching_;
  std::vector<uint32_t> groups_;
  std::unordered_map<uint64_t, Request> reqs_;
};

}  // namespace mrs_host

using mrs_host::BlockPool;
using mrs_host::KVCacheManager;

static int64_t copy_out(const std::vector<size_t> &v, int64_t *out, size_t cap) {
  if (v.size() > cap) return -2;
  for (size_t i = 0; i < v.size(); ++i) out[i] = (int64_t)v[i];
  return (int64_t)v.size();
}
static bool to_ids(const BlockPool &p, const int64_t *ids, size_t n, std::vector<size_t> &out) {
  out.resize(n);
  for (size_t i = 0; i < n; ++i) { if (!p.valid_id(ids[i])) return false; out[i] = (size_t)ids[i]; }
  return true;
}

extern "C" {
// ---- hashing
uint64_t mrs_kv_siphash(const void *data, size_t n, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds) {
  mrs_host::SipHasher h(k0, k1, c_rounds, d_rounds);
  h.write(data, n);
  return h.finish();
}
uint64_t mrs_kv_hash_block_tokens(int has_parent, uint64_t parent, const uint32_t *tokens, size_t n, const uint8_t *adapter_generation32,
                                  const char *cache_salt) {
  mrs_host::ExtraKeys e; e.adapter_generation = adapter_generation32; e.cache_salt = cache_salt;
  return mrs_host::hash_block_tokens(has_parent != 0, parent, tokens, n, e);
}
// compute_new_block_hashes (block_hash.rs:268-306); n_existing = 0 gives compute_block_hashes (:223-254).  Returns the number written.
size_t mrs_kv_compute_block_hashes(const uint32_t *tokens, size_t n_tokens, size_t block_size, const uint64_t *existing, size_t n_existing,
                                   const uint8_t *adapter_generation32, const char *cache_salt, uint64_t *out, size_t cap) {
  if (block_size == 0) return 0;
  const size_t full = n_tokens / block_size;
  if (full <= n_existing) return 0;
  mrs_host::ExtraKeys e; e.adapter_generation = adapter_generation32; e.cache_salt = cache_salt;
  bool has_parent = n_existing > 0;
  uint64_t parent = has_parent ? existing[n_existing - 1] : 0;
  size_t w = 0;
  for (size_t b = n_existing; b < full && w < cap; ++b) {
    parent = mrs_host::hash_block_tokens(has_parent, parent, tokens + b * block_size, block_size, e);
    has_parent = true;
    out[w++] = parent;
  }
  return w;
}

// ---- manager
void *mrs_kv_manager_create(size_t num_gpu_blocks, size_t block_size, int enable_caching, const uint32_t *group_ids, size_t n_groups) {
  if (num_gpu_blocks == 0 || block_size == 0) return nullptr;  // the reference asserts "Must have at least 1 GPU block"
  return new KVCacheManager(num_gpu_blocks, block_size, enable_caching != 0, group_ids, n_groups);
}
void mrs_kv_manager_destroy(void *m) { delete (KVCacheManager *)m; }
size_t mrs_kv_null_block_id(void *m) { return ((KVCacheManager *)m)->pool.null_block_id(); }
size_t mrs_kv_block_size(void *m) { return ((KVCacheManager *)m)->block_size(); }
double mrs_kv_usage(void *m) { return ((KVCacheManager *)m)->pool.usage(); }
size_t mrs_kv_num_free_blocks(void *m) { return ((KVCacheManager *)m)->pool.num_free_blocks(); }
size_t mrs_kv_num_usable_blocks(void *m) { const size_t n = ((KVCacheManager *)m)->pool.num_gpu_blocks(); return n ? n - 1 : 0; }
size_t mrs_kv_num_gpu_blocks(void *m) { return ((KVCacheManager *)m)->pool.num_gpu_blocks(); }
int mrs_kv_caching_enabled(void *m) { return ((KVCacheManager *)m)->caching_enabled(); }
// -> number of cached prefix blocks written to block_ids (cap >= n_hashes is always enough); *num_computed_tokens = blocks * block_size
int64_t mrs_kv_get_computed_blocks(void *m, const uint64_t *hashes, size_t n_hashes, size_t num_tokens, int64_t *block_ids, size_t cap,
                                   size_t *num_computed_tokens) {
  std::vector<size_t> ids;
  const size_t t = ((KVCacheManager *)m)->get_computed_blocks(hashes, n_hashes, num_tokens, ids);
  if (num_computed_tokens) *num_computed_tokens = t;
  return copy_out(ids, block_ids, cap);
}
// -> number of NEW block ids written, -1 = not enough free blocks (None), -2 = cap too small / bad id
int64_t mrs_kv_allocate_slots(void *m, uint64_t request_id, size_t num_tokens, const int64_t *computed_blocks, size_t n_computed,
                              int64_t *new_block_ids, size_t cap) {
  KVCacheManager *k = (KVCacheManager *)m;
  std::vector<size_t> comp, fresh;
  if (!to_ids(k->pool, computed_blocks, n_computed, comp)) return -2;
  const size_t required = (num_tokens + k->block_size() - 1) / k->block_size();
  const size_t have = k->request(request_id) ? k->request(request_id)->block_ids.size() : n_computed;
  if (required > have && required - have > cap) return -2;  // refuse before touching the pool
  if (!k->allocate_slots(request_id, num_tokens, comp.data(), comp.size(), fresh)) return -1;
  return copy_out(fresh, new_block_ids, cap);
}
void mrs_kv_free(void *m, uint64_t request_id) { ((KVCacheManager *)m)->free(request_id); }
void mrs_kv_trim_request_to_num_tokens(void *m, uint64_t request_id, size_t num_tokens) {
  ((KVCacheManager *)m)->trim_request_to_num_tokens(request_id, num_tokens);
}
int mrs_kv_cache_blocks(void *m, uint64_t request_id, const uint64_t *hashes, size_t n_hashes, size_t num_computed_tokens) {
  return ((KVCacheManager *)m)->cache_blocks(request_id, hashes, n_hashes, num_computed_tokens) ? 0 : -1;
}
int64_t mrs_kv_get_block_ids(void *m, uint64_t request_id, int64_t *out, size_t cap) {
  const auto *r = ((KVCacheManager *)m)->request(request_id);
  return r ? copy_out(r->block_ids, out, cap) : -1;
}
size_t mrs_kv_num_blocks_for_request(void *m, uint64_t request_id) {
  const auto *r = ((KVCacheManager *)m)->request(request_id);
  return r ? r->block_ids.size() : 0;
}
int mrs_kv_has_request(void *m, uint64_t request_id) { return ((KVCacheManager *)m)->request(request_id) != nullptr; }
size_t mrs_kv_num_cached_blocks_for_request(void *m, uint64_t request_id) {
  const auto *r = ((KVCacheManager *)m)->request(request_id);
  return r ? r->num_cached_blocks : 0;
}
int mrs_kv_reset_prefix_cache(void *m) { return ((KVCacheManager *)m)->pool.reset_prefix_cache(); }
// slots[i] = block_ids[(start + i) / bs] * bs + (start + i) % bs, _PAD_SLOT_ID (-1) past the allocation; -1 = unknown request
int mrs_kv_get_slot_mapping(void *m, uint64_t request_id, size_t start_token, size_t num_tokens, int64_t *slots) {
  KVCacheManager *k = (KVCacheManager *)m;
  const auto *r = k->request(request_id);
  if (!r) return -1;
  const size_t bs = k->block_size();
  for (size_t i = 0; i < num_tokens; ++i) {
    const size_t pos = start_token + i, b = pos / bs;
    slots[i] = b < r->block_ids.size() ? (int64_t)(r->block_ids[b] * bs + pos % bs) : (int64_t)-1;
  }
  return 0;
}
// the request's blocks, zero-padded to max_blocks; -1 = unknown request
int mrs_kv_get_block_table(void *m, uint64_t request_id, size_t max_blocks, int32_t *table) {
  const auto *r = ((KVCacheManager *)m)->request(request_id);
  if (!r) return -1;
  for (size_t i = 0; i < max_blocks; ++i) table[i] = i < r->block_ids.size() ? (int32_t)r->block_ids[i] : 0;
  return 0;
}

// ---- the pool on its own (BlockPool's public surface; the manager's pool is reachable through the same calls)
void *mrs_kv_pool_create(size_t num_gpu_blocks, int enable_caching, size_t hash_block_size) {
  return num_gpu_blocks ? new BlockPool(num_gpu_blocks, enable_caching != 0, hash_block_size) : nullptr;
}
void mrs_kv_pool_destroy(void *p) { delete (BlockPool *)p; }
void *mrs_kv_manager_pool(void *m) { return &((KVCacheManager *)m)->pool; }
size_t mrs_kv_pool_null_block_id(void *p) { return ((BlockPool *)p)->null_block_id(); }
size_t mrs_kv_pool_num_free_blocks(void *p) { return ((BlockPool *)p)->num_free_blocks(); }
size_t mrs_kv_pool_num_gpu_blocks(void *p) { return ((BlockPool *)p)->num_gpu_blocks(); }
double mrs_kv_pool_usage(void *p) { return ((BlockPool *)p)->usage(); }
size_t mrs_kv_pool_num_cached_blocks(void *p) { return ((BlockPool *)p)->num_cached_blocks(); }
size_t mrs_kv_pool_hash_block_size(void *p) { return ((BlockPool *)p)->hash_block_size(); }
int mrs_kv_pool_caching_enabled(void *p) { return ((BlockPool *)p)->caching_enabled(); }
int64_t mrs_kv_pool_block_ref_cnt(void *p, int64_t id) { return ((BlockPool *)p)->valid_id(id) ? (int64_t)((BlockPool *)p)->block_ref_cnt((size_t)id) : -1; }
int64_t mrs_kv_pool_num_block_hashes(void *p, int64_t id) { return ((BlockPool *)p)->valid_id(id) ? (int64_t)((BlockPool *)p)->block_hashes((size_t)id).size() : -1; }
int64_t mrs_kv_pool_get_new_blocks(void *p, size_t n, int64_t *out, size_t cap) {
  if (n > cap) return -2;
  std::vector<size_t> ids;
  return ((BlockPool *)p)->get_new_blocks(n, ids) ? copy_out(ids, out, cap) : -1;
}
int mrs_kv_pool_free_blocks(void *p, const int64_t *ids, size_t n) {
  std::vector<size_t> v;
  if (!to_ids(*(BlockPool *)p, ids, n, v)) return -2;
  ((BlockPool *)p)->free_blocks(v.data(), n);
  return 0;
}
int mrs_kv_pool_touch(void *p, const int64_t *ids, size_t n) {
  std::vector<size_t> v;
  if (!to_ids(*(BlockPool *)p, ids, n, v)) return -2;
  ((BlockPool *)p)->touch(v.data(), n);
  return 0;
}
int mrs_kv_pool_cache_full_blocks(void *p, const int64_t *block_ids, size_t n_ids, const uint64_t *hashes, size_t n_hashes, size_t num_cached_blocks,
                                  size_t num_full_blocks, uint32_t group_id) {
  std::vector<size_t> v;
  if (num_full_blocks > n_ids || !to_ids(*(BlockPool *)p, block_ids, n_ids, v)) return -2;
  return ((BlockPool *)p)->cache_full_blocks(v.data(), hashes, n_hashes, num_cached_blocks, num_full_blocks, group_id) ? 0 : -1;
}
// -> n_groups ids written, or -1 when any group misses (None)
int64_t mrs_kv_pool_get_cached_block(void *p, uint64_t hash, const uint32_t *group_ids, size_t n_groups, int64_t *out) {
  std::vector<size_t> ids;
  if (!((BlockPool *)p)->get_cached_block(hash, group_ids, n_groups, ids)) return -1;
  return copy_out(ids, out, n_groups);
}
int mrs_kv_pool_reset_prefix_cache(void *p) { return ((BlockPool *)p)->reset_prefix_cache(); }
void mrs_kv_pool_set_ref_cnt_for_test(void *p, int64_t id, uint32_t v) { if (((BlockPool *)p)->valid_id(id)) ((BlockPool *)p)->set_ref_cnt_for_test((size_t)id, v); }
}
