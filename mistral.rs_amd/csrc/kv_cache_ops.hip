// kv_cache_ops.hip -- paged KV cache data movement for gfx950:
//   reshape_and_cache   scatter K/V of new tokens into the paged cache
//   gather_kv_cache     paged -> dense K/V (prefix-cached prefill, sliding window)
//   copy_blocks_*       block copies for copy-on-write
// Reference: mistralrs-paged-attn/src/cuda/{reshape_and_cache_kernel.cu, gather_kv_cache_kernel.cu,
// copy_blocks_kernel.cu}; Rust FFI mistralrs-paged-attn/src/cuda/ffi.rs:96,248,440.
// Cache layouts (cache_engine.rs:458-484):
//   K cache [num_blocks, kv_heads, head_size/x, block_size, x]   x = 16 / sizeof(cache elem)
//   V cache [num_blocks, kv_heads, head_size, block_size]
// All kernels are pure byte movement (HBM-bound, tiny): coalescing is what matters -- the K
// scatter writes x-element (16-byte) groups, the V scatter is a stride-block_size transpose.
#include "common.cuh"
#include <stdio.h>
#include <stdlib.h>

namespace mrs {

template <class T> struct to_cache;

template <class SRC, class DST> __device__ __forceinline__ DST conv(SRC v) { return from_f<DST>(to_f<SRC>(v)); }

// one workgroup per token
template <class T, class CT>
__global__ void __launch_bounds__(512) reshape_and_cache_kernel(const T *__restrict__ key, const T *__restrict__ value,
                                                                CT *__restrict__ key_cache, CT *__restrict__ value_cache,
                                                                const int64_t *__restrict__ slot_mapping, int key_stride,
                                                                int value_stride, int num_heads, int head_size,
                                                                int block_size, int x) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // _PAD_SLOT_ID = -1: padding token (paged_attention/mod.rs:26)
  const int64_t block_idx = slot / block_size, block_off = slot % block_size;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size, off = i % head_size;
    const int64_t kdst = ((block_idx * num_heads + head) * (head_size / x) + off / x) * block_size * x + block_off * x + off % x;
    const int64_t vdst = ((block_idx * num_heads + head) * head_size + off) * block_size + block_off;
    key_cache[kdst] = conv<T, CT>(key[token * key_stride + i]);
    value_cache[vdst] = conv<T, CT>(value[token * value_stride + i]);
  }
}

// one workgroup per output token: seq found by binary search in cu_seq_lens
template <class CT, class OT>
__global__ void __launch_bounds__(512) gather_kv_cache_kernel(const CT *__restrict__ key_cache, const CT *__restrict__ value_cache,
                                                              OT *__restrict__ k_out, OT *__restrict__ v_out,
                                                              const int *__restrict__ block_table, const int *__restrict__ cu_seq_lens,
                                                              int num_seqs, int block_size, int block_table_stride,
                                                              int num_kv_heads, int head_size, int x) {
  const int token = blockIdx.x;
  int lo = 0, hi = num_seqs;  // largest s with cu_seq_lens[s] <= token
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cu_seq_lens[mid] <= token) lo = mid; else hi = mid; }
  const int seq = lo, pos = token - cu_seq_lens[seq];
  const int64_t block_idx = block_table[(int64_t)seq * block_table_stride + pos / block_size];
  const int block_off = pos % block_size;
  const int n = num_kv_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size, off = i % head_size;
    const int64_t ksrc = ((block_idx * num_kv_heads + head) * (head_size / x) + off / x) * block_size * x + block_off * x + off % x;
    const int64_t vsrc = ((block_idx * num_kv_heads + head) * head_size + off) * block_size + block_off;
    k_out[(int64_t)token * n + i] = conv<CT, OT>(key_cache[ksrc]);
    v_out[(int64_t)token * n + i] = conv<CT, OT>(value_cache[vsrc]);
  }
}

// grid (layer, pair): copy one block of K and V inside each layer's cache
template <class T>
__global__ void __launch_bounds__(512) copy_blocks_kernel(int64_t *key_cache_ptrs, int64_t *value_cache_ptrs,
                                                          const int64_t *__restrict__ block_mapping, int numel_key, int numel_value) {
  const int layer = blockIdx.x, pair = blockIdx.y;
  T *kc = (T *)key_cache_ptrs[layer];
  T *vc = (T *)value_cache_ptrs[layer];
  const int64_t src = block_mapping[2 * pair], dst = block_mapping[2 * pair + 1];
  for (int i = threadIdx.x; i < numel_key; i += blockDim.x) kc[dst * numel_key + i] = kc[src * numel_key + i];
  for (int i = threadIdx.x; i < numel_value; i += blockDim.x) vc[dst * numel_value + i] = vc[src * numel_value + i];
}

static void check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {  // the reference launchers exit() on a launch error (CUDA_CHECK, reshape_and_cache_kernel.cu:16-24)
    fprintf(stderr, "HIP error in %s: %s\n", what, hipGetErrorString(e));
    exit((int)e);
  }
}

}  // namespace mrs

using mrs::bf16_t;
using mrs::f16_t;

// dtype codes: 0 f16, 1 bf16, 2 f32 (3 = fp8 e4m3 cache: not built yet -> loud failure)
#define MRS_DISPATCH_T_CT(dtype, cache_dtype, CALL)                                          \
  do {                                                                                       \
    if ((cache_dtype) == 3) { fprintf(stderr, "mistralrs paged-attn (gfx950): fp8 KV cache not supported yet\n"); exit(2); } \
    else if ((dtype) == 0 && (cache_dtype) == 0) { CALL(f16_t, f16_t); }                     \
    else if ((dtype) == 1 && (cache_dtype) == 1) { CALL(bf16_t, bf16_t); }                   \
    else if ((dtype) == 2 && (cache_dtype) == 2) { CALL(float, float); }                     \
    else if ((dtype) == 2 && (cache_dtype) == 1) { CALL(float, bf16_t); } /* MI355X extra: f32 activations, bf16 cache */ \
    else if ((dtype) == 2 && (cache_dtype) == 0) { CALL(float, f16_t); }                     \
    else { fprintf(stderr, "mistralrs paged-attn (gfx950): unsupported dtype pair (%u, %u)\n", (unsigned)(dtype), (unsigned)(cache_dtype)); exit(2); } \
  } while (0)

extern "C" void reshape_and_cache(void *key, void *value, void *key_cache, void *value_cache, int64_t *slot_mapping,
                                  int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                                  int32_t key_stride, int32_t value_stride, hipStream_t stream, uint32_t dtype,
                                  uint32_t cache_dtype, float *k_scale, float *v_scale) {
  (void)k_scale; (void)v_scale;
  if (num_tokens <= 0) return;
  const int threads = num_heads * head_size < 512 ? ((num_heads * head_size + 63) / 64) * 64 : 512;
#define CALL(T, CT)                                                                                                    \
  hipLaunchKernelGGL((mrs::reshape_and_cache_kernel<T, CT>), dim3(num_tokens), dim3(threads), 0, stream, (const T *)key, \
                     (const T *)value, (CT *)key_cache, (CT *)value_cache, slot_mapping, key_stride, value_stride,      \
                     num_heads, head_size, block_size, x)
  MRS_DISPATCH_T_CT(dtype, cache_dtype, CALL);
#undef CALL
  mrs::check_launch("reshape_and_cache");
}

extern "C" void gather_kv_cache(void *key_cache, void *value_cache, void *k_out, void *v_out, float *k_scale, float *v_scale,
                                const int *block_table, const int *cu_seq_lens, int32_t num_tokens, int32_t num_seqs,
                                int32_t block_size, int32_t block_table_stride, int32_t num_kv_heads, int32_t head_size,
                                int32_t x, hipStream_t stream, uint32_t out_dtype, uint32_t cache_dtype) {
  (void)k_scale; (void)v_scale;
  if (num_tokens <= 0) return;
  const int n = num_kv_heads * head_size;
  const int threads = n < 512 ? ((n + 63) / 64) * 64 : 512;
#define CALL(OT, CT)                                                                                                      \
  hipLaunchKernelGGL((mrs::gather_kv_cache_kernel<CT, OT>), dim3(num_tokens), dim3(threads), 0, stream,                   \
                     (const CT *)key_cache, (const CT *)value_cache, (OT *)k_out, (OT *)v_out, block_table, cu_seq_lens, \
                     num_seqs, block_size, block_table_stride, num_kv_heads, head_size, x)
  MRS_DISPATCH_T_CT(out_dtype, cache_dtype, CALL);
#undef CALL
  mrs::check_launch("gather_kv_cache");
}

#define MRS_COPY_BLOCKS(name, T)                                                                                        \
  extern "C" void name(void *key_cache_ptrs, void *value_cache_ptrs, const void *block_mapping, int32_t num_layers,     \
                       int32_t num_pairs, int32_t numel_per_block_key, int32_t numel_per_block_value, int64_t stream) { \
    if (num_layers <= 0 || num_pairs <= 0) return;                                                                      \
    hipLaunchKernelGGL((mrs::copy_blocks_kernel<T>), dim3(num_layers, num_pairs), dim3(512), 0, (hipStream_t)stream,    \
                       (int64_t *)key_cache_ptrs, (int64_t *)value_cache_ptrs, (const int64_t *)block_mapping,          \
                       numel_per_block_key, numel_per_block_value);                                                     \
    mrs::check_launch(#name);                                                                                           \
  }
MRS_COPY_BLOCKS(copy_blocks_bf16, uint16_t)
MRS_COPY_BLOCKS(copy_blocks_f16, uint16_t)
MRS_COPY_BLOCKS(copy_blocks_f32, uint32_t)
MRS_COPY_BLOCKS(copy_blocks_u8, uint8_t)
