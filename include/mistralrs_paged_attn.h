/*
 * include/mistralrs_paged_attn.h -- C ABI of libmistralrspagedattention.so (gfx950 / MI355X).
 *
 * Drop-in for `libmistralrspagedattention.a` (mistralrs-paged-attn/build.rs:146-158,213-215).
 * Same symbols and argument lists as mistralrs-paged-attn/src/cuda/ffi.rs; `stream` is a hipStream_t.
 * dtype codes: 0 = f16, 1 = bf16, 2 = f32, 3 = fp8-e4m3 cache (OCP E4M3FN; block sizes 16 / 32; needs k_scale / v_scale:
 * stored = fp8_sat_rne(x / scale), read as T(float(fp8) * scale), quantization/fp8/nvidia/quant_utils.cuh:24-29,187-217).
 * Cache layouts (mistralrs-core/src/paged_attention/cache_engine.rs:458-484):
 *   K cache [num_blocks, kv_heads, head_size/x, block_size, x],  x = 16 / sizeof(cache element)
 *   V cache [num_blocks, kv_heads, head_size, block_size]
 * Launch errors terminate the process like the reference's CUDA_CHECK (reshape_and_cache_kernel.cu:16-24).
 */
#ifndef MISTRALRS_PAGED_ATTN_H
#define MISTRALRS_PAGED_ATTN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
typedef void *mrs_stream_t; /* hipStream_t */

/* ffi.rs:96-116 ; kernel reshape_and_cache_kernel.cu.  slot < 0 (= _PAD_SLOT_ID) is skipped. */
void reshape_and_cache(void *key, void *value, void *key_cache, void *value_cache, int64_t *slot_mapping,
                       int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                       int32_t key_stride, int32_t value_stride, mrs_stream_t stream, uint32_t dtype,
                       uint32_t cache_dtype, float *k_scale, float *v_scale);

/* ffi.rs:248-267 ; gather_kv_cache_kernel.cu: paged -> dense [num_tokens, kv_heads, head_size] */
void gather_kv_cache(void *key_cache, void *value_cache, void *k_out, void *v_out, float *k_scale, float *v_scale,
                     const int *block_table, const int *cu_seq_lens, int32_t num_tokens, int32_t num_seqs,
                     int32_t block_size, int32_t block_table_stride, int32_t num_kv_heads, int32_t head_size, int32_t x,
                     mrs_stream_t stream, uint32_t out_dtype, uint32_t cache_dtype);

/* ffi.rs:269-348 ; pagedattention_v1_*.cu.  One query token per sequence. softcapping == 1.0 disables it. */
#define MRS_DECL_PA_V1(d)                                                                                           \
  void paged_attention_v1_##d(void *out, void *query, void *key_cache, void *value_cache, void *alibi_slopes,       \
                              int32_t num_kv_heads, float scale, float softcapping, uint32_t *block_tables,         \
                              uint32_t *context_lens, int32_t block_size, int32_t max_context_len, int32_t num_seqs, \
                              int32_t num_heads, int32_t head_size, int32_t max_num_blocks_per_seq, int32_t q_stride, \
                              int32_t kv_block_stride, int32_t kv_head_stride, mrs_stream_t stream,                  \
                              uint32_t cache_dtype, float *k_scale, float *v_scale, const float *sinks);
MRS_DECL_PA_V1(f16) MRS_DECL_PA_V1(bf16) MRS_DECL_PA_V1(f32)
#undef MRS_DECL_PA_V1

/* ffi.rs:350-438 ; pagedattention_v2_*.cu: 512-token partitions + log-sum-exp merge.
 * exp_sums/max_logits [seqs, heads, parts] f32, tmp_out [seqs, heads, parts, head_size] in the query dtype,
 * parts = ceil(max_context_len / 512) (backend/paged_attention.rs:353-365). */
#define MRS_DECL_PA_V2(d)                                                                                            \
  void paged_attention_v2_##d(void *out, float *exp_sums, float *max_logits, void *tmp_out, void *query,             \
                              void *key_cache, void *value_cache, void *alibi_slopes, int32_t num_kv_heads,          \
                              float scale, float softcapping, uint32_t *block_tables, uint32_t *context_lens,        \
                              int32_t block_size, int32_t max_context_len, int32_t num_seqs, int32_t num_heads,      \
                              int32_t head_size, int32_t max_num_blocks_per_seq, int32_t q_stride,                   \
                              int32_t kv_block_stride, int32_t kv_head_stride, mrs_stream_t stream,                  \
                              uint32_t cache_dtype, float *k_scale, float *v_scale, const float *sinks);
MRS_DECL_PA_V2(f16) MRS_DECL_PA_V2(bf16) MRS_DECL_PA_V2(f32)
#undef MRS_DECL_PA_V2

/* ffi.rs:440-... ; copy_blocks_kernel.cu: block_mapping = int64 (src, dst) pairs; *_cache_ptrs = per-layer base addresses */
void copy_blocks_bf16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int32_t num_layers,
                      int32_t num_pairs, int32_t numel_per_block_key, int32_t numel_per_block_value, int64_t stream);
void copy_blocks_f16(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int32_t num_layers,
                     int32_t num_pairs, int32_t numel_per_block_key, int32_t numel_per_block_value, int64_t stream);
void copy_blocks_f32(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int32_t num_layers,
                     int32_t num_pairs, int32_t numel_per_block_key, int32_t numel_per_block_value, int64_t stream);
void copy_blocks_u8(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int32_t num_layers,
                    int32_t num_pairs, int32_t numel_per_block_key, int32_t numel_per_block_value, int64_t stream);

/* MI355X-native (not in the reference): explicit (query dtype, cache dtype) pair -- f32 activations over a
 * bf16 cache for the fused decode path.  v2 != 0 selects the partitioned kernel + reduce. */
void mrs_paged_attention_f32_bf16(int v2, void *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query,
                                  const void *key_cache, const void *value_cache, const void *alibi_slopes,
                                  int num_kv_heads, float scale, float softcapping, const uint32_t *block_tables,
                                  const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs,
                                  int num_heads, int head_size, int max_num_blocks_per_seq, int q_stride,
                                  int kv_block_stride, int kv_head_stride, void *stream, const float *sinks);
/* MI355X-native decode attention of the fused path: split-KV (one wave per 32-token KV chunk of a GQA group), then one
 * merge kernel that writes the result as Q8_1 blocks (o_proj's activation format).  Replaces the run
 * paged_attention_v1/v2 -> launch_mmvq_gguf_quantize_q8_1 of the reference decode step (paged_attention.rs:1477-1561,
 * fast_mmvq.rs:340-383).  Workspace sizes use max_parts = mrs_decode_attention_max_splits(max_context_len).  Returns 0, or -1 if the shape is
 * not supported (block_size 32, head_size 64/128). */
int mrs_decode_attention_max_splits(int max_context_len);
int mrs_decode_attention_q8_1_f32_bf16(void *y_q8_1, int y_stride_blocks, float *exp_sums, float *max_logits, void *tmp_out,
                                       const void *query, const void *key_cache, const void *value_cache, int num_kv_heads,
                                       float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size,
                                       int max_context_len, int num_seqs, int num_heads, int head_size,
                                       int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
                                       void *stream);
/* decode engine: the same split-KV kernel with f32 probabilities (reference CPU path) and an f32 result out [seqs][heads * 128];
 * kv_dtype 1 = bf16 pages, 0 = f16 pages; head_size 128, block_size 32 */
int mrs_decode_attention_f32_f32_bf16(float *out, float *exp_sums, float *max_logits, void *tmp_out, const void *query, const void *key_cache,
                                      const void *value_cache, int num_kv_heads, float scale, const uint32_t *block_tables,
                                      const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs, int num_heads, int head_size,
                                      int max_num_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride, int kv_dtype, void *stream);
/* ffi.rs:484-510 ; update_kvscales.cu:46-150: *k_scales = max(*k_scales, absmax(k) / 240), same for v (fp8 KV-cache scale tracking,
 * backend/scale_update.rs:81-105); k, v: num_elements values of the named dtype */
void update_kv_scales_f32(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream);
void update_kv_scales_f16(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream);
void update_kv_scales_bf16(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream);

#ifdef __cplusplus
}
#endif
#endif
