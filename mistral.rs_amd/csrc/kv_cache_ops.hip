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

// scale: only used by the fp8 (E4M3) cache -- store fp8(x / scale) saturating (quant_utils.cuh:187-217), load T(float(fp8) * scale)
template <class SRC, class DST> __device__ __forceinline__ DST conv(SRC v, float scale) { (void)scale; return from_f<DST>(to_f<SRC>(v)); }
template <> __device__ __forceinline__ fp8_t conv<float, fp8_t>(float v, float scale) { return fp8_t{float_to_fp8_e4m3(v / scale)}; }
template <> __device__ __forceinline__ fp8_t conv<f16_t, fp8_t>(f16_t v, float scale) { return fp8_t{float_to_fp8_e4m3((float)v / scale)}; }
template <> __device__ __forceinline__ fp8_t conv<bf16_t, fp8_t>(bf16_t v, float scale) { return fp8_t{float_to_fp8_e4m3(to_f<bf16_t>(v) / scale)}; }
template <> __device__ __forceinline__ float conv<fp8_t, float>(fp8_t v, float scale) { return fp8_e4m3_to_float(v.v) * scale; }
template <> __device__ __forceinline__ f16_t conv<fp8_t, f16_t>(fp8_t v, float scale) { return from_f<f16_t>(fp8_e4m3_to_float(v.v) * scale); }
template <> __device__ __forceinline__ bf16_t conv<fp8_t, bf16_t>(fp8_t v, float scale) { return from_f<bf16_t>(fp8_e4m3_to_float(v.v) * scale); }

// one workgroup per token
template <class T, class CT>
__global__ void __launch_bounds__(512) reshape_and_cache_kernel(const T *__restrict__ key, const T *__restrict__ value,
                                                                CT *__restrict__ key_cache, CT *__restrict__ value_cache,
                                                                const int64_t *__restrict__ slot_mapping, int key_stride,
                                                                int value_stride, int num_heads, int head_size,
                                                                int block_size, int x, const float *__restrict__ k_scale,
                                                                const float *__restrict__ v_scale) {
  const float ks = k_scale ? *k_scale : 1.0f, vs = v_scale ? *v_scale : 1.0f;
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // _PAD_SLOT_ID = -1: padding token (paged_attention/mod.rs:26)
  const int64_t block_idx = slot / block_size, block_off = slot % block_size;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size, off = i % head_size;
    const int64_t kdst = ((block_idx * num_heads + head) * (head_size / x) + off / x) * block_size * x + block_off * x + off % x;
    const int64_t vdst = ((block_idx * num_heads + head) * head_size + off) * block_size + block_off;
    key_cache[kdst] = conv<T, CT>(key[token * key_stride + i], ks);
    value_cache[vdst] = conv<T, CT>(value[token * value_stride + i], vs);
  }
}

// one workgroup per output token: seq found by binary search in cu_seq_lens
template <class CT, class OT>
__global__ void __launch_bounds__(512) gather_kv_cache_kernel(const CT *__restrict__ key_cache, const CT *__restrict__ value_cache,
                                                              OT *__restrict__ k_out, OT *__restrict__ v_out,
                                                              const int *__restrict__ block_table, const int *__restrict__ cu_seq_lens,
                                                              int num_seqs, int block_size, int block_table_stride,
                                                              int num_kv_heads, int head_size, int x, const float *__restrict__ k_scale,
                                                              const float *__restrict__ v_scale) {
  const float ks = k_scale ? *k_scale : 1.0f, vs = v_scale ? *v_scale : 1.0f;
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
    k_out[(int64_t)token * n + i] = conv<CT, OT>(key_cache[ksrc], ks);
    v_out[(int64_t)token * n + i] = conv<CT, OT>(value_cache[vsrc], vs);
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

// dtype codes: 0 f16, 1 bf16, 2 f32, cache only: 3 = fp8 e4m3 (needs k_scale / v_scale)
#define MRS_DISPATCH_T_CT(dtype, cache_dtype, CALL)                                          \
  do {                                                                                       \
    if ((cache_dtype) == 3 && (!k_scale || !v_scale)) { fprintf(stderr, "mistralrs paged-attn (gfx950): fp8 KV cache needs k_scale / v_scale\n"); exit(2); } \
    else if ((dtype) == 0 && (cache_dtype) == 3) { CALL(f16_t, mrs::fp8_t); }                \
    else if ((dtype) == 1 && (cache_dtype) == 3) { CALL(bf16_t, mrs::fp8_t); }               \
    else if ((dtype) == 2 && (cache_dtype) == 3) { CALL(float, mrs::fp8_t); }                \
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
  if (num_tokens <= 0) return;
  const float *ksp = cache_dtype == 3 ? k_scale : nullptr, *vsp = cache_dtype == 3 ? v_scale : nullptr;
  const int threads = num_heads * head_size < 512 ? ((num_heads * head_size + 63) / 64) * 64 : 512;
#define CALL(T, CT)                                                                                                    \
  hipLaunchKernelGGL((mrs::reshape_and_cache_kernel<T, CT>), dim3(num_tokens), dim3(threads), 0, stream, (const T *)key, \
                     (const T *)value, (CT *)key_cache, (CT *)value_cache, slot_mapping, key_stride, value_stride,      \
                     num_heads, head_size, block_size, x, ksp, vsp)
  MRS_DISPATCH_T_CT(dtype, cache_dtype, CALL);
#undef CALL
  mrs::check_launch("reshape_and_cache");
}

extern "C" void gather_kv_cache(void *key_cache, void *value_cache, void *k_out, void *v_out, float *k_scale, float *v_scale,
                                const int *block_table, const int *cu_seq_lens, int32_t num_tokens, int32_t num_seqs,
                                int32_t block_size, int32_t block_table_stride, int32_t num_kv_heads, int32_t head_size,
                                int32_t x, hipStream_t stream, uint32_t out_dtype, uint32_t cache_dtype) {
  if (num_tokens <= 0) return;
  const float *ksp = cache_dtype == 3 ? k_scale : nullptr, *vsp = cache_dtype == 3 ? v_scale : nullptr;
  const int n = num_kv_heads * head_size;
  const int threads = n < 512 ? ((n + 63) / 64) * 64 : 512;
#define CALL(OT, CT)                                                                                                      \
  hipLaunchKernelGGL((mrs::gather_kv_cache_kernel<CT, OT>), dim3(num_tokens), dim3(threads), 0, stream,                   \
                     (const CT *)key_cache, (const CT *)value_cache, (OT *)k_out, (OT *)v_out, block_table, cu_seq_lens, \
                     num_seqs, block_size, block_table_stride, num_kv_heads, head_size, x, ksp, vsp)
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

// ---- update_kv_scales_{f32,f16,bf16}: k_scale = max(k_scale, absmax(k) / 240), same for v (update_kvscales.cu:46-150; Rust ffi.rs:484-510;
//      caller backend/scale_update.rs:81-105).  Non-negative floats order like their bit patterns, so the update is an integer atomicMax.
namespace mrs {
template <class T>
__global__ void __launch_bounds__(512) update_kv_scales_kernel(const T *__restrict__ k, const T *__restrict__ v, long n, float *k_scale, float *v_scale) {
  __shared__ float red[2][8];
  float mk = 0.f, mv = 0.f;
  for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n; i += (long)gridDim.x * 512) {
    const float a = fabsf(to_f<T>(k[i])), b = fabsf(to_f<T>(v[i]));
    if (a > mk) mk = a;   // comparisons as the reference: NaNs never win
    if (b > mv) mv = b;
  }
  mk = wave_max(mk); mv = wave_max(mv);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mk; red[1][threadIdx.x >> 6] = mv; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 8; ++w) { mk = fmaxf(mk, red[0][w]); mv = fmaxf(mv, red[1][w]); }
    const float ck = mk / 240.0f, cv = mv / 240.0f;
    if (ck > 0.0f) atomicMax((int *)k_scale, __float_as_int(ck));
    if (cv > 0.0f) atomicMax((int *)v_scale, __float_as_int(cv));
  }
}
template <class T> static void update_kv_scales(void *k, void *v, long n, float *ks, float *vs, int64_t stream) {
  long blocks = (n + 511) / 512;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((update_kv_scales_kernel<T>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, (const T *)k, (const T *)v, n, ks, vs);
  check_launch("update_kv_scales");
}
}  // namespace mrs
extern "C" void update_kv_scales_f32(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream) {
  mrs::update_kv_scales<float>(k, v, num_elements, k_scales, v_scales, stream);
}
extern "C" void update_kv_scales_f16(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream) {
  mrs::update_kv_scales<f16_t>(k, v, num_elements, k_scales, v_scales, stream);
}
extern "C" void update_kv_scales_bf16(void *k, void *v, const long num_elements, float *k_scales, float *v_scales, int64_t stream) {
  mrs::update_kv_scales<bf16_t>(k, v, num_elements, k_scales, v_scales, stream);
}
