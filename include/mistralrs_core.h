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

#ifdef __cplusplus
}
#endif
#endif
