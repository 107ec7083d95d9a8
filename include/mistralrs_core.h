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

#ifdef __cplusplus
}
#endif
#endif
