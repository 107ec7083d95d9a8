/*
 * include/mistralrs_core.h -- C ABI of libmistralrscuda.so, hot-path subset (gfx950 / MI355X).
 *
 * Drop-in symbols for the part of `libmistralrscuda.a` (mistralrs-core/build.rs:59-70) that the
 * Llama / Mistral / Mixtral decode graph touches.  Rust declarations: mistralrs-core/src/cuda/ffi.rs:75-140.
 * GDN / SSM / dflash / unquantized MoE kernels of that library are out of scope.
 */
#ifndef MISTRALRS_CORE_H
#define MISTRALRS_CORE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* residual_dst = T(x + residual); norm_dst = T(residual_dst * rsqrt(mean(residual_dst^2) + eps) * weight)
 * replaces mistralrs-core/src/cuda/sort.cu:352-460,701-727 ; ffi.rs:108-140 */
#define MRS_DECL_NORMS(d)                                                                                              \
  void add_rms_norm_##d(const void *x, const void *residual, const void *weight, void *residual_dst, void *norm_dst,   \
                        int nrows, int ncols, float eps, int64_t stream);                                              \
  /* dst = T((residual + x * rsqrt(mean(x^2) + eps) * weight) * scale[0])   sort.cu:244-350 ; ffi.rs:75-107 */         \
  void rms_norm_residual_##d(const void *x, const void *residual, const void *weight, const void *scale, void *dst,    \
                             int nrows, int ncols, float eps, int64_t stream);                                         \
  /* MI355X-native: plain RMSNorm = candle_nn::ops::rms_norm as called by RmsNorm::forward (layers.rs:403-414) */      \
  void mrs_rms_norm_##d(const void *x, const void *weight, void *dst, int nrows, int ncols, float eps, int64_t stream);
MRS_DECL_NORMS(f32) MRS_DECL_NORMS(f16) MRS_DECL_NORMS(bf16)
#undef MRS_DECL_NORMS

/* MoE router: softmax / sigmoid / raw scores over n_experts logits per row -> top_k (ids, weights), ties to the lowest expert id.
 * score_mode 0 raw, 1 softmax, 2 sigmoid; weight_mode 0 score, 1 softmax over the picked raw logits, 2 sigmoid(raw); optional
 * selection_bias / expert_scale [n_experts] (NULL = none), clamp, renormalise by max(sum, norm_min), output_scale.  n_experts in
 * {1,2,4,8,16,32,64,128,256,512,576}, other values are ignored like the reference's switch.
 * replaces mistralrs-core/src/cuda/sort.cu:1097-1470 ; ffi.rs:523-579 ; caller ops.rs:259-336 (moe_router_topk) */
#include <stdbool.h>
void moe_router_topk_f32(const void *logits, float *weights, uint32_t *ids, const float *selection_bias, const float *expert_scale, int n_rows,
                         int n_experts, int top_k, int score_mode, int weight_mode, bool renormalize, bool clamp_logits, float clamp_min,
                         float clamp_max, float norm_min, float output_scale, int64_t stream);
void moe_router_topk_f16(const void *logits, float *weights, uint32_t *ids, const float *selection_bias, const float *expert_scale, int n_rows,
                         int n_experts, int top_k, int score_mode, int weight_mode, bool renormalize, bool clamp_logits, float clamp_min,
                         float clamp_max, float norm_min, float output_scale, int64_t stream);
void moe_router_topk_bf16(const void *logits, float *weights, uint32_t *ids, const float *selection_bias, const float *expert_scale, int n_rows,
                          int n_experts, int top_k, int score_mode, int weight_mode, bool renormalize, bool clamp_logits, float clamp_min,
                          float clamp_max, float norm_min, float output_scale, int64_t stream);

/* Sampling: top-k of one f32 logits row over a large vocabulary + the pieces of the full-softmax normaliser (Sampler::sample_topk_on_device, sampler.rs:1171-1260;
 * top-p / min-p / the draw stay on the host).  The caller owns every buffer: block_values / block_indices [nrows][nblocks][k], block_maxes / block_sums
 * [nrows][nblocks] (workspace, nblocks = ceil(ncols / chunk_size)), packed_out [nrows][2k + 2] = k values, k indices as f32, denom, max(x / T).
 * Order: value descending, index ascending on ties; NaN and -inf are never selected, missing entries are (-inf, 0).  1 <= k <= 128, chunk_size <= 4096 (the
 * reference's host wrapper passes 2048).  replaces mistralrs-core/src/cuda/sort.cu:1502-1823,2146-2206 ; ffi.rs:583-624 ; caller ops.rs:691-1000 */
void topk_large_f32(const float *input, float *block_values, uint32_t *block_indices, float *block_maxes, float *block_sums, float *values_out,
                    uint32_t *indices_out, float *softmax_info_out, int ncols, int k, int chunk_size, int nblocks, float inv_temperature, int64_t stream);
void topk_large_f32_packed(const float *input, float *block_values, uint32_t *block_indices, float *block_maxes, float *block_sums, float *packed_out,
                           int ncols, int k, int chunk_size, int nblocks, float inv_temperature, int64_t stream);
void topk_large_f32_packed_batched(const float *input, const float *inv_temperatures, float *block_values, uint32_t *block_indices, float *block_maxes,
                                   float *block_sums, float *packed_out, int nrows, int ncols, int k, int chunk_size, int nblocks, int64_t stream);

/* Greedy sampling: arg-max of f32 logits rows.  block_values / block_indices [nrows][nblocks] are workspace; packed_out [nrows][2] = (max logit, token id as f32),
 * token_ids_out [nrows] (either may be NULL).  Lowest index on ties; a row holding a NaN reports token 0xffffffff and (NaN, NaN); a row of -inf reports token 0.
 * replaces mistralrs-core/src/cuda/sort.cu:1825-1912,2071-2143,2207-2238 ; ffi.rs:643-665 ; callers ops.rs:1232-2050 (cuda_top1_logits_f32_*) */
void top1_large_f32_packed(const float *input, float *block_values, uint32_t *block_indices, float *packed_out, uint32_t *token_ids_out, int ncols, int chunk_size,
                           int nblocks, int64_t stream);
void top1_large_f32_packed_batched(const float *input, float *block_values, uint32_t *block_indices, float *packed_out, uint32_t *token_ids_out, int nrows, int ncols,
                                   int chunk_size, int nblocks, int64_t stream);

/* Sampler pre-processing: dst [n] = x [n] (f32), then for the n_tokens listed token ids (ids >= n ignored): penalties -- skipped where count <= 0;
 * v -= count * frequency_penalty + presence_penalty; if repetition_penalty != 1: v = v > 0 ? v / rp : v * rp -- or additive biases.
 * replaces mistralrs-core/src/cuda/sort.cu:8-110 ; ffi.rs:45-65 ; callers sampler.rs:1113-1169 */
void apply_sparse_penalties_f32(const void *x, void *dst, const uint32_t *token_ids, const float *counts, int n, int n_tokens, float frequency_penalty,
                                float presence_penalty, float repetition_penalty, int64_t stream);
void apply_sparse_logits_bias_f32(const void *x, void *dst, const uint32_t *token_ids, const float *biases, int n, int n_tokens, int64_t stream);

#ifdef __cplusplus
}
#endif
#endif
