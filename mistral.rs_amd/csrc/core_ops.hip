// core_ops.hip -- hot-path subset of libmistralrscuda (mistralrs-core/src/cuda): RMSNorm family.
//   add_rms_norm_{f32,f16,bf16}          mistralrs-core/src/cuda/sort.cu:352-460,701-727 ; ffi.rs:108-140
//   rms_norm_residual_{f32,f16,bf16}     sort.cu:244-350 ; ffi.rs:75-107
//   mrs_rms_norm_{f32,f16,bf16}          plain RMSNorm = candle_nn::ops::rms_norm, which Llama's
//                                        RmsNorm::forward calls (mistralrs-core/src/layers.rs:403-414);
//                                        candle is not in-tree, so this symbol is MI355X-native.
// One workgroup (256 threads, 4 waves) per row; rows are streamed with 16-byte loads, the row is
// kept in registers between the reduction and the scale pass (<= 8 x 16 B per thread, i.e. rows up
// to 16K elements in 16-bit types / 8K in f32; longer rows re-read from L2).  f32 accumulation.
#include "common.cuh"

namespace mrs {

constexpr int RN_THREADS = 256;
constexpr int RN_MAXV = 8;  // 16-byte vectors cached per thread

template <class T> struct vec16 { static constexpr int N = 16 / sizeof(T); };

template <class T> __device__ __forceinline__ void load_vec(const T *p, float *o) {
  constexpr int N = vec16<T>::N;
  const int4 raw = *(const int4 *)p;
  const T *e = (const T *)&raw;
#pragma unroll
  for (int i = 0; i < N; ++i) o[i] = to_f<T>(e[i]);
}
template <class T> __device__ __forceinline__ void store_vec(T *p, const float *v) {
  constexpr int N = vec16<T>::N;
  int4 raw;
  T *e = (T *)&raw;
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = from_f<T>(v[i]);
  *(int4 *)p = raw;
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

// MODE 0: norm_dst = rms(x) * w
// MODE 1: r = round_T(x + residual); residual_dst = r; norm_dst = rms(r) * w            (add_rms_norm)
// MODE 2: dst = (residual + x * inv_rms(x) * w) * scale                                 (rms_norm_residual)
template <class T, int MODE, bool VEC>
__global__ void __launch_bounds__(RN_THREADS) rms_norm_kernel(const T *__restrict__ x, const T *__restrict__ residual,
                                                              const T *__restrict__ weight, const T *__restrict__ scale,
                                                              T *__restrict__ residual_dst, T *__restrict__ norm_dst, int ncols, float eps) {
  __shared__ float red[4];
  constexpr int N = vec16<T>::N;
  const size_t row = (size_t)blockIdx.x * ncols;
  const int tid = threadIdx.x;
  float sum = 0.f;
  if constexpr (VEC) {
    const int nvec = ncols / N;
    float cache[RN_MAXV][N];
#pragma unroll
    for (int j = 0; j < RN_MAXV; ++j) {
      const int v = tid + j * RN_THREADS;
      if (v < nvec) {
        load_vec<T>(x + row + (size_t)v * N, cache[j]);
        if constexpr (MODE == 1) {
          float r[N];
          load_vec<T>(residual + row + (size_t)v * N, r);
#pragma unroll
          for (int i = 0; i < N; ++i) cache[j][i] = round_to<T>(cache[j][i] + r[i]);
          store_vec<T>(residual_dst + row + (size_t)v * N, cache[j]);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) sum = fmaf(cache[j][i], cache[j][i], sum);
      }
    }
    for (int v = tid + RN_MAXV * RN_THREADS; v < nvec; v += RN_THREADS) {  // very long rows: uncached tail
      float t[N];
      load_vec<T>(x + row + (size_t)v * N, t);
      if constexpr (MODE == 1) {
        float r[N];
        load_vec<T>(residual + row + (size_t)v * N, r);
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = round_to<T>(t[i] + r[i]);
        store_vec<T>(residual_dst + row + (size_t)v * N, t);
      }
#pragma unroll
      for (int i = 0; i < N; ++i) sum = fmaf(t[i], t[i], sum);
    }
    const float inv = rsqrtf(block_sum_256(sum, red) / (float)ncols + eps);
    const float sc = (MODE == 2 && scale) ? to_f<T>(scale[0]) : 1.0f;
    auto finish = [&](int v, const float *xv) {
      float w[N], o[N];
      load_vec<T>(weight + (size_t)v * N, w);
      if constexpr (MODE == 2) {
        float r[N];
        load_vec<T>(residual + row + (size_t)v * N, r);
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = (r[i] + xv[i] * inv * w[i]) * sc;
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = xv[i] * inv * w[i];
      }
      store_vec<T>(norm_dst + row + (size_t)v * N, o);
    };
#pragma unroll
    for (int j = 0; j < RN_MAXV; ++j) {
      const int v = tid + j * RN_THREADS;
      if (v < nvec) finish(v, cache[j]);
    }
    for (int v = tid + RN_MAXV * RN_THREADS; v < nvec; v += RN_THREADS) {
      float t[N];
      load_vec<T>((MODE == 1 ? (const T *)residual_dst : x) + row + (size_t)v * N, t);
      finish(v, t);
    }
  } else {
    for (int c = tid; c < ncols; c += RN_THREADS) {
      float v = to_f<T>(x[row + c]);
      if constexpr (MODE == 1) { v = round_to<T>(v + to_f<T>(residual[row + c])); residual_dst[row + c] = from_f<T>(v); }
      sum += v * v;
    }
    const float inv = rsqrtf(block_sum_256(sum, red) / (float)ncols + eps);
    const float sc = (MODE == 2 && scale) ? to_f<T>(scale[0]) : 1.0f;
    for (int c = tid; c < ncols; c += RN_THREADS) {
      const float v = to_f<T>((MODE == 1 ? (const T *)residual_dst : x)[row + c]);
      const float n = v * inv * to_f<T>(weight[c]);
      norm_dst[row + c] = from_f<T>(MODE == 2 ? (to_f<T>(residual[row + c]) + n) * sc : n);
    }
  }
}

template <class T, int MODE>
static void launch_rms(const void *x, const void *residual, const void *weight, const void *scale, void *residual_dst,
                       void *norm_dst, int nrows, int ncols, float eps, int64_t stream) {
  if (nrows <= 0 || ncols <= 0) return;
  constexpr int N = vec16<T>::N;
  const uintptr_t al = (uintptr_t)x | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)residual_dst | (uintptr_t)norm_dst;
  const bool vec = (ncols % N == 0) && (al % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((rms_norm_kernel<T, MODE, true>), dim3(nrows), dim3(RN_THREADS), 0, (hipStream_t)stream, (const T *)x,
                       (const T *)residual, (const T *)weight, (const T *)scale, (T *)residual_dst, (T *)norm_dst, ncols, eps);
  else
    hipLaunchKernelGGL((rms_norm_kernel<T, MODE, false>), dim3(nrows), dim3(RN_THREADS), 0, (hipStream_t)stream, (const T *)x,
                       (const T *)residual, (const T *)weight, (const T *)scale, (T *)residual_dst, (T *)norm_dst, ncols, eps);
}

}  // namespace mrs

#define MRS_RMS_FAMILY(tag, T)                                                                                              \
  extern "C" void mrs_rms_norm_##tag(const void *x, const void *weight, void *dst, int nrows, int ncols, float eps,          \
                                     int64_t stream) {                                                                      \
    mrs::launch_rms<T, 0>(x, nullptr, weight, nullptr, nullptr, dst, nrows, ncols, eps, stream);                            \
  }                                                                                                                         \
  extern "C" void add_rms_norm_##tag(const void *x, const void *residual, const void *weight, void *residual_dst,           \
                                     void *norm_dst, int nrows, int ncols, float eps, int64_t stream) {                     \
    mrs::launch_rms<T, 1>(x, residual, weight, nullptr, residual_dst, norm_dst, nrows, ncols, eps, stream);                 \
  }                                                                                                                         \
  extern "C" void rms_norm_residual_##tag(const void *x, const void *residual, const void *weight, const void *scale,       \
                                          void *dst, int nrows, int ncols, float eps, int64_t stream) {                     \
    mrs::launch_rms<T, 2>(x, residual, weight, scale, nullptr, dst, nrows, ncols, eps, stream);                             \
  }
MRS_RMS_FAMILY(f32, float)
MRS_RMS_FAMILY(f16, mrs::f16_t)
MRS_RMS_FAMILY(bf16, mrs::bf16_t)

// ---------------------------------------------------------------------------------------------------- MoE router top-k
// moe_router_topk_{f32,f16,bf16}: drop-in for mistralrs-core/src/cuda/sort.cu:1097-1470 (Rust: cuda/ffi.rs:523-579; caller
// ops.rs:259-336 moe_router_topk).  Per row of router logits [n_experts]: optional clamp, NaN -> -inf; score = raw / softmax /
// sigmoid; selection = score (+ bias); top_k rounds of arg-max (ties: lowest expert id); weight = score / raw (then softmax over
// the k picks) / sigmoid(raw); optional renormalisation by max(sum, norm_min); times output_scale (* expert_scale[id]).
// n_experts outside {1,2,4,...,512,576} is silently ignored like the reference's switch.
// wave64 mapping: a row is owned by 32 lanes (half a wave), expert e lives in lane e % 32 slot e / 32 -- the reference's layout, so the
// strided partial sums and the xor-butterfly reductions (masks 16..1 never leave a 32-lane half) add in the reference's order.
namespace mrs {
constexpr int ROUTER_MAX_SLOTS = 18;  // 576 / 32
__device__ __forceinline__ float half_sum32(float v) {
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float half_max32(float v) {
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
// softmax over the first `limit` entries (entry index = lane + 32 i); entries >= limit become 0 when use_limit
__device__ __forceinline__ void router_softmax(float (&v)[ROUTER_MAX_SLOTS], int slots, int limit, bool use_limit, int lane) {
  float mx = -INFINITY;
  for (int i = 0; i < slots; ++i) if (!use_limit || lane + 32 * i < limit) mx = fmaxf(mx, v[i]);
  mx = half_max32(mx);
  float sum = 0.f;
  for (int i = 0; i < slots; ++i) {
    if (!use_limit || lane + 32 * i < limit) { v[i] = expf(v[i] - mx); sum += v[i]; } else v[i] = 0.f;
  }
  sum = half_sum32(sum);
  const float inv = 1.0f / sum;
  for (int i = 0; i < slots; ++i) if (!use_limit || lane + 32 * i < limit) v[i] *= inv;
}

template <class T>
__global__ void __launch_bounds__(256) moe_router_topk_kernel(const T *__restrict__ logits, float *__restrict__ weights, uint32_t *__restrict__ ids,
                                                              const float *__restrict__ bias, const float *__restrict__ expert_scale, int n_rows,
                                                              int n_experts, int top_k, int score_mode, int weight_mode, bool renormalize,
                                                              bool clamp_logits, float clamp_min, float clamp_max, float norm_min, float output_scale) {
  const int lane = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= n_rows) return;  // whole 32-lane halves leave together; the shuffles below stay inside a half
  logits += (size_t)row * n_experts; weights += (size_t)row * top_k; ids += (size_t)row * top_k;
  const int slots = n_experts > 32 ? n_experts / 32 : 1;
  float raw[ROUTER_MAX_SLOTS], score[ROUTER_MAX_SLOTS], sel[ROUTER_MAX_SLOTS], ow[ROUTER_MAX_SLOTS];
  uint32_t oid[ROUTER_MAX_SLOTS];
  for (int i = 0; i < slots; ++i) {
    const int e = lane + 32 * i;
    float v = e < n_experts ? to_f<T>(logits[e]) : -INFINITY;
    if (clamp_logits && e < n_experts) v = fminf(fmaxf(v, clamp_min), clamp_max);
    if (v != v) v = -INFINITY;
    raw[i] = v; score[i] = v; ow[i] = 0.f; oid[i] = 0;
  }
  if (score_mode == 1) router_softmax(score, slots, n_experts, false, lane);
  else if (score_mode == 2) for (int i = 0; i < slots; ++i) score[i] = 1.0f / (1.0f + expf(-score[i]));
  for (int i = 0; i < slots; ++i) {
    const int e = lane + 32 * i;
    sel[i] = score[i];
    if (bias && e < n_experts) sel[i] += bias[e];
    if (sel[i] != sel[i]) sel[i] = -INFINITY;
  }
  for (int k = 0; k < top_k; ++k) {
    float bs = sel[0], bsc = score[0], br = raw[0];
    int be = lane;
    for (int i = 1; i < slots; ++i) {
      const int e = lane + 32 * i;
      if (e < n_experts && sel[i] > bs) { bs = sel[i]; bsc = score[i]; br = raw[i]; be = e; }
    }
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) {
      const float os = __shfl_xor(bs, m, 64), osc = __shfl_xor(bsc, m, 64), orw = __shfl_xor(br, m, 64);
      const int oe = __shfl_xor(be, m, 64);
      if (os > bs || (os == bs && oe < be)) { bs = os; bsc = osc; br = orw; be = oe; }
    }
    float out = bsc;
    if (weight_mode == 1) out = br;
    else if (weight_mode == 2) out = 1.0f / (1.0f + expf(-br));
    if ((k & 31) == lane) { ow[k / 32] = out; oid[k / 32] = (uint32_t)be; }
    if ((be & 31) == lane) sel[be / 32] = -INFINITY;
  }
  if (weight_mode == 1) router_softmax(ow, slots, top_k, true, lane);
  if (renormalize) {
    float sum = 0.f;
    for (int i = 0; i < slots; ++i) if (lane + 32 * i < top_k) sum += ow[i];
    sum = fmaxf(half_sum32(sum), norm_min);
    const float inv = 1.0f / sum;
    for (int i = 0; i < slots; ++i) ow[i] *= inv;
  }
  for (int i = 0; i < slots; ++i) {
    const int idx = lane + 32 * i;
    if (idx < top_k) {
      float sc = output_scale;
      if (expert_scale) sc *= expert_scale[oid[i]];
      weights[idx] = ow[i] * sc;
      ids[idx] = oid[i];
    }
  }
}
template <class T>
static void launch_moe_router(const void *logits, float *weights, uint32_t *ids, const float *bias, const float *expert_scale, int n_rows, int n_experts,
                              int top_k, int score_mode, int weight_mode, bool renormalize, bool clamp_logits, float clamp_min, float clamp_max,
                              float norm_min, float output_scale, int64_t stream) {
  if (n_rows <= 0) return;
  switch (n_experts) { case 1: case 2: case 4: case 8: case 16: case 32: case 64: case 128: case 256: case 512: case 576: break; default: return; }
  hipLaunchKernelGGL((moe_router_topk_kernel<T>), dim3((n_rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const T *)logits, weights, ids, bias,
                     expert_scale, n_rows, n_experts, top_k, score_mode, weight_mode, renormalize, clamp_logits, clamp_min, clamp_max, norm_min,
                     output_scale);
}
}  // namespace mrs
#define MRS_ROUTER(tag, T)                                                                                                                       \
  extern "C" void moe_router_topk_##tag(const void *logits, float *weights, uint32_t *ids, const float *selection_bias, const float *expert_scale, \
                                        int n_rows, int n_experts, int top_k, int score_mode, int weight_mode, bool renormalize,                  \
                                        bool clamp_logits, float clamp_min, float clamp_max, float norm_min, float output_scale, int64_t stream) { \
    mrs::launch_moe_router<T>(logits, weights, ids, selection_bias, expert_scale, n_rows, n_experts, top_k, score_mode, weight_mode, renormalize,   \
                              clamp_logits, clamp_min, clamp_max, norm_min, output_scale, stream);                                               \
  }
MRS_ROUTER(f32, float)
MRS_ROUTER(f16, mrs::f16_t)
MRS_ROUTER(bf16, mrs::bf16_t)
