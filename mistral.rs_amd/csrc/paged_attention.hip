// paged_attention.hip -- C-ABI launchers `paged_attention_v1_{f16,bf16,f32}` / `paged_attention_v2_*`.
// Replaces mistralrs-paged-attn/src/cuda/pagedattention_v{1,2}_{f16,bf16,f32}.cu (+ launchers in
// pagedattention.cuh:685-880).  Rust FFI: mistralrs-paged-attn/src/cuda/ffi.rs:269-378; caller:
// backend/paged_attention.rs:103-410 (v1 when one 512-token partition or seqs*heads > 512, else v2).
//
// One translation unit per (query dtype, cache dtype) pair: build passes
//   -DMRS_PA_T=mrs::bf16_t -DMRS_PA_CT=mrs::bf16_t -DMRS_PA_TAG=bf16 [-DMRS_PA_EXPORT_ABI]
#include "paged_attention.cuh"
#include <stdio.h>
#include <stdlib.h>

namespace mrs {

constexpr int PA_PARTITION = 512;  // paged_attention.rs:302-307 sizes the v2 workspace for 512-token partitions
constexpr int PA_NW = 4;
constexpr size_t PA_LDS_MAX = 150 * 1024;

static void pa_check(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { fprintf(stderr, "HIP error in %s: %s\n", what, hipGetErrorString(e)); exit((int)e); }
}

static size_t pa_lds_bytes(int G, int HD, int LS, int nw = PA_NW) {
  const size_t body = (size_t)G * LS > (size_t)nw * G * HD ? (size_t)G * LS : (size_t)nw * G * HD;
  return ((size_t)G * HD + body + 2 * (size_t)G * nw) * sizeof(float);
}
// a launch that leaves most CUs without a workgroup (batch-1 decode: heads x partitions) is a latency chain per workgroup: more waves walk the
// partition's 32-token blocks in parallel.  MRS_PA_WIDE = 0 (off) / 8 / 16 waves (default 8).
static int pa_wide() {  // -1: default (16 waves with one head per workgroup: 104 VGPRs; 8 with two: 246)
  static const int w = [] { const char *e = getenv("MRS_PA_WIDE"); const int v = e ? atoi(e) : -1; return v == 16 ? 16 : (v == 0 ? 0 : (v == 8 ? 8 : -1)); }();
  return w;
}

template <class T, class CT, int HD, int BS, int G, int PART, int NW>
static void pa_launch_nw(const PagedAttnArgs &a, dim3 grid, hipStream_t s) {
  auto kern = paged_attention_kernel<T, CT, HD, BS, G, PART, true, NW>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), pa_lds_bytes(G, HD, a.logits_stride, NW), s, a);
}

template <class T, class CT, int HD, int BS, int G, int PART>
static void pa_launch(const PagedAttnArgs &a, int num_seqs, int max_parts, hipStream_t s) {
  const dim3 grid(a.num_heads / G, num_seqs, PART > 0 ? max_parts : 1);
  if constexpr (HD == 128 && BS == 32 && G <= 2) {  // the shapes of the headline models; elsewhere the 4-wave kernel stays the only instantiation
    int w = pa_wide();
    if (w < 0) w = G == 1 ? 16 : 8;
    if (w == 16 && G != 1) w = 8;  // 16 waves cap the kernel at 128 VGPRs: only the one-head instantiation fits without scratch
    if (w && (long)grid.x * grid.y * grid.z <= 256 && pa_lds_bytes(G, HD, a.logits_stride, w) <= PA_LDS_MAX) {
      if constexpr (G == 1) { if (w == 16) return pa_launch_nw<T, CT, HD, BS, G, PART, 16>(a, grid, s); }
      return pa_launch_nw<T, CT, HD, BS, G, PART, 8>(a, grid, s);
    }
  }
  pa_launch_nw<T, CT, HD, BS, G, PART, PA_NW>(a, grid, s);
}

template <class T, class CT, int HD, int BS, int PART>
static void pa_pick_g(const PagedAttnArgs &a, int num_seqs, int max_parts, hipStream_t s) {
  const int qpk = a.num_heads / a.num_kv_heads;
  // few (sequence, kv head, partition) triples: one query head per workgroup puts 4-8 x the workgroups on the chip (K / V are re-read per head out of L2);
  // MRS_PA_MAX_G overrides (measurements)
  static const int env_g = [] { const char *e = getenv("MRS_PA_MAX_G"); return e ? atoi(e) : 0; }();
  const long triples = (long)a.num_kv_heads * num_seqs * (PART > 0 ? max_parts : 1);
  const int max_g = env_g > 0 ? env_g : (triples <= 32 ? 1 : 8);
  auto fits = [&](int G) { return G <= max_g && qpk % G == 0 && pa_lds_bytes(G, HD, a.logits_stride) <= PA_LDS_MAX; };
  if constexpr (HD <= 128 && BS >= 16) {
    if (fits(8)) return pa_launch<T, CT, HD, BS, 8, PART>(a, num_seqs, max_parts, s);
    if (fits(4)) return pa_launch<T, CT, HD, BS, 4, PART>(a, num_seqs, max_parts, s);
    if (fits(2)) return pa_launch<T, CT, HD, BS, 2, PART>(a, num_seqs, max_parts, s);
  }
  if (pa_lds_bytes(1, HD, a.logits_stride) > PA_LDS_MAX) {
    fprintf(stderr, "paged_attention (gfx950): context of %d tokens does not fit the 160 KiB LDS logits buffer; use v2\n", a.logits_stride);
    exit(2);
  }
  pa_launch<T, CT, HD, BS, 1, PART>(a, num_seqs, max_parts, s);
}

template <class T, class CT, int BS, int PART>
static void pa_pick_hd(const PagedAttnArgs &a, int head_size, int num_seqs, int max_parts, hipStream_t s) {
  switch (head_size) {  // the reference's head-size table (pagedattention.cuh:712-745)
  case 64: pa_pick_g<T, CT, 64, BS, PART>(a, num_seqs, max_parts, s); break;
  case 80: pa_pick_g<T, CT, 80, BS, PART>(a, num_seqs, max_parts, s); break;
  case 96: pa_pick_g<T, CT, 96, BS, PART>(a, num_seqs, max_parts, s); break;
  case 112: pa_pick_g<T, CT, 112, BS, PART>(a, num_seqs, max_parts, s); break;
  case 128: pa_pick_g<T, CT, 128, BS, PART>(a, num_seqs, max_parts, s); break;
  case 192: pa_pick_g<T, CT, 192, BS, PART>(a, num_seqs, max_parts, s); break;
  case 256: pa_pick_g<T, CT, 256, BS, PART>(a, num_seqs, max_parts, s); break;
  case 512: pa_pick_g<T, CT, 512, BS, PART>(a, num_seqs, max_parts, s); break;
  default: break;  // silently ignored by the reference as well
  }
}

template <class T, int PART> static void pa_reduce(void *out, const float *exp_sums, const float *max_logits, const void *tmp_out,
                                                   const uint32_t *context_lens, int max_parts, const float *sinks, int head_size,
                                                   int num_heads, int num_seqs, hipStream_t s) {
  const dim3 grid(num_heads, num_seqs), block(128);
  const size_t lds = 2 * (size_t)max_parts * sizeof(float);
#define RED(HD) hipLaunchKernelGGL((paged_attention_reduce_kernel<T, HD, PART>), grid, block, lds, s, (T *)out, exp_sums, max_logits, (const T *)tmp_out, context_lens, max_parts, sinks)
  switch (head_size) {
  case 64: RED(64); break; case 80: RED(80); break; case 96: RED(96); break; case 112: RED(112); break;
  case 128: RED(128); break; case 192: RED(192); break; case 256: RED(256); break; case 512: RED(512); break;
  default: break;
  }
#undef RED
}

template <class T, class CT>
void paged_attention_dispatch(bool v2, void *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query,
                              const void *key_cache, const void *value_cache, const void *alibi_slopes, int num_kv_heads,
                              float scale, float softcapping, const uint32_t *block_tables, const uint32_t *context_lens,
                              int block_size, int max_context_len, int num_seqs, int num_heads, int head_size,
                              int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
                              hipStream_t stream, const float *sinks, const float *k_scale = nullptr, const float *v_scale = nullptr) {
  if (num_seqs <= 0 || num_heads <= 0) return;
  PagedAttnArgs a{};
  a.k_scale = k_scale; a.v_scale = v_scale;
  a.exp_sums = exp_sums; a.max_logits = max_logits; a.out = v2 ? tmp_out : out; a.q = query;
  a.k_cache = key_cache; a.v_cache = value_cache; a.block_tables = block_tables; a.context_lens = context_lens;
  a.alibi_slopes = (const float *)alibi_slopes; a.sinks = sinks;
  a.num_heads = num_heads; a.num_kv_heads = num_kv_heads; a.max_num_blocks_per_seq = max_num_blocks_per_seq;
  a.q_stride = q_stride; a.kv_block_stride = kv_block_stride; a.kv_head_stride = kv_head_stride;
  a.scale = scale; a.softcapping = softcapping;
  const int max_parts = (max_context_len + PA_PARTITION - 1) / PA_PARTITION;
  a.logits_stride = v2 ? PA_PARTITION : (max_context_len + block_size - 1) / block_size * block_size;
  if (a.logits_stride < block_size) a.logits_stride = block_size;
#define BS_CASE(BS)                                                                                    \
  case BS:                                                                                             \
    if (v2) pa_pick_hd<T, CT, BS, PA_PARTITION>(a, head_size, num_seqs, max_parts, stream);            \
    else pa_pick_hd<T, CT, BS, 0>(a, head_size, num_seqs, max_parts, stream);                          \
    break;
  if constexpr (sizeof(CT) == 1) {  // fp8 cache: 16 tokens per 16-byte group, block sizes 16 and 32
    if (block_size != 16 && block_size != 32) { fprintf(stderr, "paged_attention (gfx950): fp8 KV cache needs block_size 16 or 32, got %d\n", block_size); exit(2); }
    if (!k_scale || !v_scale) { fprintf(stderr, "paged_attention (gfx950): fp8 KV cache needs k_scale / v_scale\n"); exit(2); }
    switch (block_size) { BS_CASE(16) BS_CASE(32) default: break; }
  } else {
    switch (block_size) { BS_CASE(8) BS_CASE(16) BS_CASE(32) default: break; }
  }
#undef BS_CASE
  if (v2) pa_reduce<T, PA_PARTITION>(out, exp_sums, max_logits, tmp_out, context_lens, max_parts, sinks, head_size, num_heads, num_seqs, stream);
  pa_check(v2 ? "paged_attention_v2" : "paged_attention_v1");
}

}  // namespace mrs

using mrs::bf16_t;
using mrs::f16_t;

#define PA_CAT_(a, b) a##b
#define PA_CAT(a, b) PA_CAT_(a, b)

// MI355X-native entry: explicit (query dtype, cache dtype) pair, used by the fused decode path
#ifndef MRS_PA_FP8
extern "C" void PA_CAT(mrs_paged_attention_, MRS_PA_TAG)(
    int v2, void *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache,
    const void *value_cache, const void *alibi_slopes, int num_kv_heads, float scale, float softcapping,
    const uint32_t *block_tables, const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs,
    int num_heads, int head_size, int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
    void *stream, const float *sinks) {
  mrs::paged_attention_dispatch<MRS_PA_T, MRS_PA_CT>(v2 != 0, out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                                                     alibi_slopes, num_kv_heads, scale, softcapping, block_tables, context_lens,
                                                     block_size, max_context_len, num_seqs, num_heads, head_size,
                                                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride,
                                                     (hipStream_t)stream, sinks);
}
#endif

#ifdef MRS_PA_FP8
// fp8 (E4M3) cache: same entry plus the two scale pointers; the reference-ABI exports of the matching query dtype forward here when
// cache_dtype == 3 (pagedattention.cuh:752-790 picks the <scalar_t, uint8_t, kFp8E4M3> instantiation)
extern "C" void PA_CAT(mrs_paged_attention_fp8_, MRS_PA_TAG)(
    int v2, void *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache,
    const void *value_cache, const void *alibi_slopes, int num_kv_heads, float scale, float softcapping,
    const uint32_t *block_tables, const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs,
    int num_heads, int head_size, int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
    void *stream, const float *sinks, const float *k_scale, const float *v_scale) {
  mrs::paged_attention_dispatch<MRS_PA_T, MRS_PA_CT>(v2 != 0, out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                                                     alibi_slopes, num_kv_heads, scale, softcapping, block_tables, context_lens,
                                                     block_size, max_context_len, num_seqs, num_heads, head_size,
                                                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride,
                                                     (hipStream_t)stream, sinks, k_scale, v_scale);
}
#endif
#ifdef MRS_PA_DECODE_Q8_1
// MI355X decode attention for the fused path.  head_size 128 / block 32 / bf16 cache: the wave-per-KV-chunk kernel
// (paged_attention.cuh, "decode attention v3") + one merge kernel that writes Q8_1 blocks for o_proj.  Other shapes:
// the reference-style kernel over DEC_PART-token partitions + merge.  Workspace (caller-owned):
//   tmp_out [seqs, heads, max_splits, hd] f32, exp_sums / max_logits [seqs, heads, max_splits]
//   with max_splits = mrs_decode_attention_max_splits(max_context_len).
namespace mrs {
constexpr int DEC_PART = 128;
constexpr int DEC_MAX_SPLITS = 64;
static int dec_bpw(int max_context_len) {  // KV blocks (32 tokens) per wave
  const int nblk = (max_context_len + 31) / 32;
  return nblk <= DEC_MAX_SPLITS ? 1 : (nblk + DEC_MAX_SPLITS - 1) / DEC_MAX_SPLITS;
}
}  // namespace mrs
extern "C" int mrs_decode_attention_max_splits(int max_context_len) {
  const int nblk = (max_context_len + 31) / 32, bpw = mrs::dec_bpw(max_context_len);
  const int v3 = (nblk + bpw - 1) / bpw, v2 = (max_context_len + mrs::DEC_PART - 1) / mrs::DEC_PART;
  return v3 > v2 ? v3 : v2;
}
extern "C" int PA_CAT(mrs_decode_attention_q8_1_, MRS_PA_TAG)(
    void *y_q8_1, int y_stride_blocks, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache,
    const void *value_cache, int num_kv_heads, float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size,
    int max_context_len, int num_seqs, int num_heads, int head_size, int max_num_blocks_per_seq, int q_stride, int kv_block_stride,
    int kv_head_stride, void *stream) {
  using namespace mrs;
  if (block_size != 32 || (head_size != 128 && head_size != 64) || num_seqs <= 0 || num_heads % num_kv_heads) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int max_splits = mrs_decode_attention_max_splits(max_context_len);
  const int qpk = num_heads / num_kv_heads;
  if (head_size == 128 && (qpk == 1 || qpk == 2 || qpk == 4 || qpk == 8)) {
    const int bpw = dec_bpw(max_context_len);
    const int nsplit = ((max_context_len + 31) / 32 + bpw - 1) / bpw;
    const dim3 grid(num_kv_heads, num_seqs, (nsplit + 3) / 4);
#define DEC(G) hipLaunchKernelGGL((decode_attn_wave_kernel<G>), grid, dim3(256), 0, s, (const float *)query, (const uint16_t *)key_cache,          \
                                  (const uint16_t *)value_cache, block_tables, context_lens, (float *)tmp_out, max_logits, exp_sums, num_heads, \
                                  num_kv_heads, max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, bpw, max_splits, scale)
    switch (qpk) { case 1: DEC(1); break; case 2: DEC(2); break; case 4: DEC(4); break; default: DEC(8); break; }
#undef DEC
    hipLaunchKernelGGL(decode_attn_merge_q8_1_kernel<128>, dim3(num_heads, num_seqs), dim3(128), 0, s, (uint8_t *)y_q8_1, y_stride_blocks,
                       (const float *)tmp_out, max_logits, exp_sums, context_lens, bpw, max_splits);
    pa_check("mrs_decode_attention_q8_1 (v3)");
    return 0;
  }
  PagedAttnArgs a{};
  a.exp_sums = exp_sums; a.max_logits = max_logits; a.out = tmp_out; a.q = query; a.k_cache = key_cache; a.v_cache = value_cache;
  a.block_tables = block_tables; a.context_lens = context_lens; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads;
  a.max_num_blocks_per_seq = max_num_blocks_per_seq; a.q_stride = q_stride; a.kv_block_stride = kv_block_stride;
  a.kv_head_stride = kv_head_stride; a.scale = scale; a.softcapping = 1.0f; a.logits_stride = DEC_PART;
  if (head_size == 128) {
    pa_pick_g<MRS_PA_T, MRS_PA_CT, 128, 32, DEC_PART>(a, num_seqs, max_splits, s);
    hipLaunchKernelGGL((paged_attention_reduce_q8_1_kernel<128, DEC_PART>), dim3(num_heads, num_seqs), dim3(128), 0, s, (uint8_t *)y_q8_1,
                       y_stride_blocks, exp_sums, max_logits, (const float *)tmp_out, context_lens, max_splits);
  } else {
    pa_pick_g<MRS_PA_T, MRS_PA_CT, 64, 32, DEC_PART>(a, num_seqs, max_splits, s);
    hipLaunchKernelGGL((paged_attention_reduce_q8_1_kernel<64, DEC_PART>), dim3(num_heads, num_seqs), dim3(64), 0, s, (uint8_t *)y_q8_1,
                       y_stride_blocks, exp_sums, max_logits, (const float *)tmp_out, context_lens, max_splits);
  }
  pa_check("mrs_decode_attention_q8_1");
  return 0;
}
// Decode engine: same split-KV kernel with f32 probabilities (the reference CPU path keeps them in f32) and an f32 result
// out [seqs][heads * head_size]; kv_dtype 1 = bf16 pages, 0 = f16 pages.  head_size 128, block 32, GQA group 1 / 2 / 4 / 8.
extern "C" int PA_CAT(mrs_decode_attention_f32_, MRS_PA_TAG)(
    float *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache, const void *value_cache, int num_kv_heads,
    float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs, int num_heads,
    int head_size, int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride, int kv_dtype, void *stream) {
  using namespace mrs;
  if (block_size != 32 || head_size != 128 || num_seqs <= 0 || num_heads % num_kv_heads || (kv_dtype != 0 && kv_dtype != 1)) return -1;
  const int qpk = num_heads / num_kv_heads;
  if (qpk != 1 && qpk != 2 && qpk != 4 && qpk != 8) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int max_splits = mrs_decode_attention_max_splits(max_context_len);
  const int bpw = dec_bpw(max_context_len);
  const int nsplit = ((max_context_len + 31) / 32 + bpw - 1) / bpw;
  const dim3 grid(num_kv_heads, num_seqs, (nsplit + 3) / 4);
#define DEC(G, CT) hipLaunchKernelGGL((decode_attn_wave_kernel<G, CT, false>), grid, dim3(256), 0, s, (const float *)query, (const uint16_t *)key_cache,   \
                                      (const uint16_t *)value_cache, block_tables, context_lens, (float *)tmp_out, max_logits, exp_sums, num_heads,        \
                                      num_kv_heads, max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, bpw, max_splits, scale)
#define DECG(CT) switch (qpk) { case 1: DEC(1, CT); break; case 2: DEC(2, CT); break; case 4: DEC(4, CT); break; default: DEC(8, CT); break; }
  if (kv_dtype == 1) { DECG(bf16_t) } else { DECG(f16_t) }
#undef DECG
#undef DEC
  hipLaunchKernelGGL((decode_attn_merge_q8_1_kernel<128, true>), dim3(num_heads, num_seqs), dim3(128), 0, s, (uint8_t *)out, 0, (const float *)tmp_out,
                     max_logits, exp_sums, context_lens, bpw, max_splits);
  pa_check("mrs_decode_attention_f32");
  return 0;
}
#endif

#ifdef MRS_PA_EXPORT_ABI
// The reference ABI: query dtype in the symbol name, cache dtype code 0 f16 / 1 bf16 / 2 f32 / 3 fp8-e4m3.
// Like the reference, a non-fp8 cache is read as the query dtype (pagedattention_v1_bf16.cu:22-29).
extern "C" void PA_CAT(mrs_paged_attention_fp8_, MRS_PA_TAG)(
    int v2, void *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache,
    const void *value_cache, const void *alibi_slopes, int num_kv_heads, float scale, float softcapping,
    const uint32_t *block_tables, const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs,
    int num_heads, int head_size, int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
    void *stream, const float *sinks, const float *k_scale, const float *v_scale);  // paged_attention_<tag>_fp8.o
extern "C" void PA_CAT(paged_attention_v1_, MRS_PA_TAG)(
    void *out, void *query, void *key_cache, void *value_cache, void *alibi_slopes, int32_t num_kv_heads, float scale,
    float softcapping, uint32_t *block_tables, uint32_t *context_lens, int32_t block_size, int32_t max_context_len,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t max_num_blocks_per_seq, int32_t q_stride,
    int32_t kv_block_stride, int32_t kv_head_stride, hipStream_t stream, uint32_t cache_dtype, float *k_scale,
    float *v_scale, const float *sinks) {
  if (cache_dtype == 3)
    return PA_CAT(mrs_paged_attention_fp8_, MRS_PA_TAG)(0, out, nullptr, nullptr, nullptr, query, key_cache, value_cache, alibi_slopes, num_kv_heads,
                                                        scale, softcapping, block_tables, context_lens, block_size, max_context_len, num_seqs,
                                                        num_heads, head_size, max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride,
                                                        stream, sinks, k_scale, v_scale);
  mrs::paged_attention_dispatch<MRS_PA_T, MRS_PA_CT>(false, out, nullptr, nullptr, nullptr, query, key_cache, value_cache, alibi_slopes,
                                                     num_kv_heads, scale, softcapping, block_tables, context_lens, block_size,
                                                     max_context_len, num_seqs, num_heads, head_size, max_num_blocks_per_seq,
                                                     q_stride, kv_block_stride, kv_head_stride, stream, sinks);
}
extern "C" void PA_CAT(paged_attention_v2_, MRS_PA_TAG)(
    void *out, float *exp_sums, float *max_logits, void *tmp_out, void *query, void *key_cache, void *value_cache,
    void *alibi_slopes, int32_t num_kv_heads, float scale, float softcapping, uint32_t *block_tables,
    uint32_t *context_lens, int32_t block_size, int32_t max_context_len, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t max_num_blocks_per_seq, int32_t q_stride, int32_t kv_block_stride, int32_t kv_head_stride,
    hipStream_t stream, uint32_t cache_dtype, float *k_scale, float *v_scale, const float *sinks) {
  if (cache_dtype == 3)
    return PA_CAT(mrs_paged_attention_fp8_, MRS_PA_TAG)(1, out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, alibi_slopes,
                                                        num_kv_heads, scale, softcapping, block_tables, context_lens, block_size, max_context_len,
                                                        num_seqs, num_heads, head_size, max_num_blocks_per_seq, q_stride, kv_block_stride,
                                                        kv_head_stride, stream, sinks, k_scale, v_scale);
  mrs::paged_attention_dispatch<MRS_PA_T, MRS_PA_CT>(true, out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                                                     alibi_slopes, num_kv_heads, scale, softcapping, block_tables, context_lens,
                                                     block_size, max_context_len, num_seqs, num_heads, head_size,
                                                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, stream, sinks);
}
#endif
