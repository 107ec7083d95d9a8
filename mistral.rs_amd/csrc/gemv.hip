// gemv.hip -- dense (unquantized) decode GEMV behind launch_gemv_{bf16,f16,f32}:  Y[b][row] = T( sum_k A[row][k] * X[b][k] + bias[row] ),
// A [M, K] row-major, X [B, K], B = 1..8, f32 fma accumulation, one rounding to T at the end.
//   replaces mistralrs-quant/kernels/gemv/gemv.cu:50-282 ; Rust: src/gemv/ffi.rs:12-56 ; callers src/gemv/mod.rs:250-470
//   (UnquantLinear decode path: lm_head / router gate / any bf16 layer left unquantized, batch <= 8).
// MI355X formulation (HBM-bound, each weight byte read once per token): a wave owns a run of consecutive rows (one contiguous byte
// range), a lane reads 16-byte pieces of a row with non-temporal loads (64 lanes x 16 B = 1 KiB per load instruction), the activations
// (B x K, a few KiB) come from L1/L2, products accumulate in f32 with v_fma, the wave reduction is DPP.  ~16 waves per CU in the grid.
// The reference sums a thread's strided pairs in order and then butterflies; the order here differs, so results agree to f32
// accumulation error before the final rounding to T (tests: tolerance = that bound + one rounding of T).
#include "common.cuh"

namespace mrs {

template <int CTRL> __device__ __forceinline__ float gemv_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float gemv_wave_sum(float v) {  // quads -> rows of 16 -> the four rows (as mmvq_core.cuh wave_sum_dpp)
  v += gemv_dpp<0xB1>(v);
  v += gemv_dpp<0x4E>(v);
  v += gemv_dpp<0x141>(v);
  v += gemv_dpp<0x140>(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

template <class T> struct Piece;  // 16 bytes of T -> floats
template <> struct Piece<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const int4 v, float (&f)[4]) {
    f[0] = __uint_as_float((uint32_t)v.x); f[1] = __uint_as_float((uint32_t)v.y); f[2] = __uint_as_float((uint32_t)v.z); f[3] = __uint_as_float((uint32_t)v.w);
  }
};
template <> struct Piece<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const int4 v, float (&f)[8]) {
    const uint32_t w[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};
template <> struct Piece<f16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const int4 v, float (&f)[8]) {
    const uint32_t w[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = half_bits_to_float((uint16_t)(w[i] & 0xffff)); f[2 * i + 1] = half_bits_to_float((uint16_t)(w[i] >> 16)); }
  }
};

template <class T, int B, bool VEC>
__global__ void __launch_bounds__(256) gemv_kernel(const T *__restrict__ A, const T *__restrict__ X, const T *__restrict__ bias, T *__restrict__ Y,
                                                   int M, int K, int rows_per_wave, bool has_bias) {
  constexpr int N = Piece<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int first = (blockIdx.x * 4 + wave) * rows_per_wave;
  const int last = min(M, first + rows_per_wave);
  const int npieces = VEC ? K / N : 0;  // VEC: K % N == 0 and 16-byte aligned bases => every row starts on a 16-byte boundary
  for (int row = first; row < last; ++row) {  // wave-uniform
    const T *a = A + (size_t)row * K;
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b] = 0.0f;
    if constexpr (VEC) {
      for (int p = lane; p < npieces; p += 64) {
        float af[N];
        Piece<T>::unpack(ld16nt_a4(a + (size_t)p * N), af);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          float xf[N];
          Piece<T>::unpack(*(const int4 *)(X + (size_t)b * K + (size_t)p * N), xf);
#pragma unroll
          for (int i = 0; i < N; ++i) acc[b] = fmaf(af[i], xf[i], acc[b]);
        }
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        const float av = to_f<T>(a[k]);
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b] = fmaf(av, to_f<T>(X[(size_t)b * K + k]), acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b] = gemv_wave_sum(acc[b]);
    if (lane == 0) {
      const float bv = has_bias ? to_f<T>(bias[row]) : 0.0f;
#pragma unroll
      for (int b = 0; b < B; ++b) Y[(size_t)b * M + row] = from_f<T>(acc[b] + bv);
    }
  }
}

template <class T, int B> static void gemv_go(const T *A, const T *X, const T *bias, T *Y, int M, int K, bool has_bias, hipStream_t s) {
  constexpr int N = Piece<T>::N;
  int rpw = (M + 4095) / 4096;  // ~4096 waves = 16 per CU
  if (rpw < 1) rpw = 1;
  const int grid = (M + 4 * rpw - 1) / (4 * rpw);
  const bool vec = (K % N) == 0 && (((uintptr_t)A | (uintptr_t)X) & 15) == 0;
  if (vec) hipLaunchKernelGGL((gemv_kernel<T, B, true>), dim3(grid), dim3(256), 0, s, A, X, bias, Y, M, K, rpw, has_bias);
  else hipLaunchKernelGGL((gemv_kernel<T, B, false>), dim3(grid), dim3(256), 0, s, A, X, bias, Y, M, K, rpw, has_bias);
}

template <class T> static void gemv_run(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream) {
  if (M <= 0) return;
  const T *a = (const T *)A, *x = (const T *)X, *bi = (const T *)bias;
  T *y = (T *)Y;
  hipStream_t s = (hipStream_t)stream;
  switch (batch_size) {
  case 2: gemv_go<T, 2>(a, x, bi, y, M, K, has_bias, s); break;
  case 3: gemv_go<T, 3>(a, x, bi, y, M, K, has_bias, s); break;
  case 4: gemv_go<T, 4>(a, x, bi, y, M, K, has_bias, s); break;
  case 5: gemv_go<T, 5>(a, x, bi, y, M, K, has_bias, s); break;
  case 6: gemv_go<T, 6>(a, x, bi, y, M, K, has_bias, s); break;
  case 7: gemv_go<T, 7>(a, x, bi, y, M, K, has_bias, s); break;
  case 8: gemv_go<T, 8>(a, x, bi, y, M, K, has_bias, s); break;
  default: gemv_go<T, 1>(a, x, bi, y, M, K, has_bias, s); break;  // 1, and (as the reference's dispatch) anything outside 1..8
  }
}

}  // namespace mrs

extern "C" void launch_gemv_bf16(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream) {
  mrs::gemv_run<mrs::bf16_t>(A, X, bias, Y, M, K, batch_size, has_bias, stream);
}
extern "C" void launch_gemv_f16(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream) {
  mrs::gemv_run<mrs::f16_t>(A, X, bias, Y, M, K, batch_size, has_bias, stream);
}
extern "C" void launch_gemv_f32(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream) {
  mrs::gemv_run<float>(A, X, bias, Y, M, K, batch_size, has_bias, stream);
}
