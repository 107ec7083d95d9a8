// quant_ops.hip -- elementwise glue ops of libmistralrsquant for gfx950.
//   rotary_embedding / rotary_embedding_positions   mistralrs-quant/kernels/rotary/rotary.cu:9-196 ; src/rotary/ffi.rs
//   fused_glu_{f16,bf16,f32}                         mistralrs-quant/kernels/ops/ops.cu:860-971 ; src/utils/ffi.rs:274-306
// Both are tiny HBM/L2-bound passes; they exist for drop-in completeness -- the fused decode path
// (ext_decode.hip) folds RoPE and the GLU into the GEMV epilogues instead.
#include "common.cuh"

namespace mrs {

// In-place RoPE on q [tokens, heads, head_size] and k [tokens, kv_heads, head_size].
// `rot_dim` is the number of rotated PAIRS (cos/sin row length), exactly as in the reference ABI.
// Arithmetic is done in the tensor dtype: each product and the sum round to T (rotary.cu:9-33).
template <class T, bool NEOX, bool POS>
__global__ void __launch_bounds__(512) rotary_kernel(T *__restrict__ query, T *__restrict__ key, const T *__restrict__ cos_cache,
                                                     const T *__restrict__ sin_cache, const uint32_t *__restrict__ positions,
                                                     int rot_dim, int64_t query_stride, int64_t key_stride, int num_heads,
                                                     int num_kv_heads, int head_size) {
  const int64_t token = blockIdx.x;
  const int64_t rowi = POS ? (int64_t)positions[token] : token;
  const T *cos_ptr = cos_cache + rowi * rot_dim, *sin_ptr = sin_cache + rowi * rot_dim;
  auto apply = [&](T *arr, int rot_offset) {
    const int xi = NEOX ? rot_offset : 2 * rot_offset;
    const int yi = NEOX ? rot_dim + rot_offset : 2 * rot_offset + 1;
    const float c = to_f<T>(cos_ptr[rot_offset]), s = to_f<T>(sin_ptr[rot_offset]);
    const float x = to_f<T>(arr[xi]), y = to_f<T>(arr[yi]);
    float xo, yo;
    rope_pair<T>(x, y, c, s, xo, yo);
    arr[xi] = from_f<T>(xo);
    arr[yi] = from_f<T>(yo);
  };
  const int nq = num_heads * rot_dim;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) apply(query + token * query_stride + (int64_t)(i / rot_dim) * head_size, i % rot_dim);
  const int nk = num_kv_heads * rot_dim;
  for (int i = threadIdx.x; i < nk; i += blockDim.x) apply(key + token * key_stride + (int64_t)(i / rot_dim) * head_size, i % rot_dim);
}

template <bool POS>
static void launch_rotary(void *query, void *key, void *cos_cache, void *sin_cache, void *positions, int is_neox, int head_size,
                          int64_t num_tokens, int rot_dim, int num_heads, int num_kv_heads, int64_t query_stride,
                          int64_t key_stride, uint32_t dtype, int64_t stream) {
  if (num_tokens <= 0) return;
  int threads = num_heads * rot_dim;
  threads = threads > 512 ? 512 : (threads + 63) / 64 * 64;
#define ROT(T, NX)                                                                                                        \
  hipLaunchKernelGGL((rotary_kernel<T, NX, POS>), dim3((unsigned)num_tokens), dim3(threads), 0, (hipStream_t)stream,      \
                     (T *)query, (T *)key, (const T *)cos_cache, (const T *)sin_cache, (const uint32_t *)positions, rot_dim, \
                     query_stride, key_stride, num_heads, num_kv_heads, head_size)
  if (is_neox) {
    if (dtype == 0) ROT(f16_t, true); else if (dtype == 1) ROT(bf16_t, true); else if (dtype == 2) ROT(float, true);
  } else {
    if (dtype == 0) ROT(f16_t, false); else if (dtype == 1) ROT(bf16_t, false); else if (dtype == 2) ROT(float, false);
  }
#undef ROT
}

// out[r, c] = T(act(float a[r, c])) * b[r, c]  (product rounded to T), rows may be strided
template <class T>
__global__ void __launch_bounds__(256) fused_glu_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                        uint32_t cols, uint32_t a_row_stride, uint32_t b_row_stride,
                                                        uint64_t n, int activation) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / cols, c = i % cols;
    const float av = to_f<T>(a[r * a_row_stride + c]), bv = to_f<T>(b[r * b_row_stride + c]);
    out[i] = from_f<T>(round_to<T>(glu_act(av, activation)) * bv);
  }
}

template <class T>
static void launch_fused_glu(const void *a, const void *b, void *out, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                             uint32_t b_row_stride, int activation, hipStream_t stream) {
  if (rows == 0 || cols == 0) return;
  const uint64_t n = (uint64_t)rows * cols;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((fused_glu_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T *)a, (const T *)b, (T *)out, cols,
                     a_row_stride, b_row_stride, n, activation);
}

}  // namespace mrs

extern "C" void rotary_embedding(void *query, void *key, void *cos_cache, void *sin_cache, int32_t is_neox, int32_t head_size,
                                 int64_t num_tokens, int32_t rot_dim, int32_t num_heads, int32_t num_kv_heads,
                                 int64_t query_stride, int64_t key_stride, uint32_t dtype, int64_t stream) {
  mrs::launch_rotary<false>(query, key, cos_cache, sin_cache, nullptr, is_neox, head_size, num_tokens, rot_dim, num_heads,
                            num_kv_heads, query_stride, key_stride, dtype, stream);
}
extern "C" void rotary_embedding_positions(void *query, void *key, void *cos_cache, void *sin_cache, void *positions,
                                           int32_t is_neox, int32_t head_size, int64_t num_tokens, int32_t rot_dim,
                                           int32_t seq_len, int32_t num_heads, int32_t num_kv_heads, int64_t query_stride,
                                           int64_t key_stride, uint32_t dtype, int64_t stream) {
  (void)seq_len;
  mrs::launch_rotary<true>(query, key, cos_cache, sin_cache, positions, is_neox, head_size, num_tokens, rot_dim, num_heads,
                           num_kv_heads, query_stride, key_stride, dtype, stream);
}
extern "C" void fused_glu_f16(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                              uint32_t b_row_stride, int activation, hipStream_t stream) {
  mrs::launch_fused_glu<mrs::f16_t>(a, b, output, rows, cols, a_row_stride, b_row_stride, activation, stream);
}
extern "C" void fused_glu_bf16(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                               uint32_t b_row_stride, int activation, hipStream_t stream) {
  mrs::launch_fused_glu<mrs::bf16_t>(a, b, output, rows, cols, a_row_stride, b_row_stride, activation, stream);
}
extern "C" void fused_glu_f32(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                              uint32_t b_row_stride, int activation, hipStream_t stream) {
  mrs::launch_fused_glu<float>(a, b, output, rows, cols, a_row_stride, b_row_stride, activation, stream);
}
