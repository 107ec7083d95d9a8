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
