// hqq.hip -- HQQ (half-quadratic quantization) weight unpack + dequantize and the bit-packing kernels, MI355X / gfx950.
//
// Drop-in for the C ABI of mistralrs-quant/src/hqq/ffi.rs:1-74 (dequantize_{8,4,2,1}bit_u8_kernel_{f32,f16,bf16},
// dequantize_3bit_32_kernel_{f32,f16,bf16}; NO stream argument: the reference launches on the default stream, hqq.cu:37-80) and
// of src/hqq/bitpack_ffi.rs:1-40 (launch_pack_{1,2,3,4,8}bit_kernel with a stream), behind HqqLayer::dequantize /
// HqqBits::bitpack_type (hqq/mod.rs:874-1090,150-400); HqqLayer::forward = dequantize_w + dense matmul (mod.rs:1092-1100,1163-1171).
//
// Layout (axis 0 grouping): packed Wq [h][w] (u8, or i32 holding ten 3-bit values); scale, zero [w] (one per group column);
// out [P*h][w] where chunk c of a packed element (most significant first) lands at row c*h + r:
//     out[c*h*w + i] = (T(q_c(i)) - zero[i % w]) * scale[i % w]          arithmetic in T (hqq.cu:26-35,94-110,171-186,278-301,399-427)
// i.e. for f16 / bf16 both the subtraction and the product are rounded to T (evaluated here in f32 and rounded after each
// operation: exact double rounding for + and x of two T operands).
//
// MI355X notes: pure HBM streams (1 B in, P x sizeof(T) out per packed byte).  A thread owns FOUR consecutive packed elements of a
// row (one 4-byte load, or 16 B for the 3-bit i32 format) and stores 4 consecutive outputs per chunk (16 B f32 / 8 B f16,bf16), so
// every wave store covers 1 KiB / 512 B contiguous; scale / zero are read once per thread as 4-wide vectors.  Shapes with w % 4 != 0
// (never produced by the quantizer for real layers) take the scalar kernel.
#include "common.cuh"

namespace mrs {

template <class T> __device__ __forceinline__ T hqq_one(unsigned q, float z, float s) {
  return from_f<T>(round_to<T>((float)q - z) * s);
}

template <int BITS> struct HqqFmt;
template <> struct HqqFmt<8> { static constexpr int P = 1; typedef uint8_t packed_t; };
template <> struct HqqFmt<4> { static constexpr int P = 2; typedef uint8_t packed_t; };
template <> struct HqqFmt<2> { static constexpr int P = 4; typedef uint8_t packed_t; };
template <> struct HqqFmt<1> { static constexpr int P = 8; typedef uint8_t packed_t; };
template <> struct HqqFmt<3> { static constexpr int P = 10; typedef int32_t packed_t; };

// chunk c (0 = most significant) of a packed element
template <int BITS> __device__ __forceinline__ unsigned hqq_chunk(unsigned v, int c) {
  if constexpr (BITS == 3) return (v >> (27 - 3 * c)) & 7u;
  else return (v >> (8 - BITS * (c + 1))) & ((1u << BITS) - 1u);
}

template <int BITS, class T>
__global__ void __launch_bounds__(256) hqq_dequant_vec4_kernel(const typename HqqFmt<BITS>::packed_t *__restrict__ wq, const T *__restrict__ scale,
                                                               const T *__restrict__ zero, T *__restrict__ out, size_t n, int w) {
  constexpr int P = HqqFmt<BITS>::P;
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int j = (int)(i % (size_t)w);
  unsigned v[4];
  if constexpr (BITS == 3) {
    const int4 t = *(const int4 *)(wq + i);
    v[0] = (unsigned)t.x; v[1] = (unsigned)t.y; v[2] = (unsigned)t.z; v[3] = (unsigned)t.w;
  } else {
    const unsigned t = *(const unsigned *)(wq + i);
    v[0] = t & 0xff; v[1] = (t >> 8) & 0xff; v[2] = (t >> 16) & 0xff; v[3] = t >> 24;
  }
  float z[4], s[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { z[e] = to_f<T>(zero[j + e]); s[e] = to_f<T>(scale[j + e]); }
#pragma unroll
  for (int c = 0; c < P; ++c) {
    T o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hqq_one<T>(hqq_chunk<BITS>(v[e], c), z[e], s[e]);
    T *dst = out + (size_t)c * n + i;
    if constexpr (sizeof(T) == 4) *(float4 *)dst = *(const float4 *)o;
    else *(uint2 *)dst = *(const uint2 *)o;
  }
}

template <int BITS, class T>
__global__ void __launch_bounds__(256) hqq_dequant_scalar_kernel(const typename HqqFmt<BITS>::packed_t *__restrict__ wq, const T *__restrict__ scale,
                                                                 const T *__restrict__ zero, T *__restrict__ out, size_t n, int w) {
  constexpr int P = HqqFmt<BITS>::P;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = (int)(i % (size_t)w);
  const unsigned v = (unsigned)wq[i] & (BITS == 3 ? 0xffffffffu : 0xffu);
  const float z = to_f<T>(zero[j]), s = to_f<T>(scale[j]);
#pragma unroll
  for (int c = 0; c < P; ++c) out[(size_t)c * n + i] = hqq_one<T>(hqq_chunk<BITS>(v, c), z, s);
}

template <int BITS, class T> static void hqq_dequant_launch(const void *wq, const void *scale, const void *zero, void *out, int h, int w) {
  if (h <= 0 || w <= 0) return;
  const size_t n = (size_t)h * w;
  typedef typename HqqFmt<BITS>::packed_t P;
  const uintptr_t al = (uintptr_t)wq | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)zero;
  hipStream_t s = nullptr;  // the reference ABI has no stream parameter: default stream (hqq.cu:37-45)
  if (w % 4 == 0 && al % 16 == 0)
    hipLaunchKernelGGL((hqq_dequant_vec4_kernel<BITS, T>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const P *)wq, (const T *)scale,
                       (const T *)zero, (T *)out, n, w);
  else
    hipLaunchKernelGGL((hqq_dequant_scalar_kernel<BITS, T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const P *)wq, (const T *)scale,
                       (const T *)zero, (T *)out, n, w);
}

// pack: out[row][col] = OR_i (in[row + i*step][col] & mask) << shift_i, step = rows / P, i = 0 is the most significant chunk
// (hqq_bitpack.cu:7-35,37-65,67-95,97-124); rows beyond P*step are ignored like the reference does
template <int BITS, class IN>
__global__ void __launch_bounds__(256) hqq_pack_kernel(const IN *__restrict__ in, typename HqqFmt<BITS>::packed_t *__restrict__ out, size_t rows, size_t width) {
  constexpr int P = HqqFmt<BITS>::P;
  const size_t step = rows / P;
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= step * width) return;
  unsigned packed = 0;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const unsigned v = (unsigned)in[tid + (size_t)i * step * width] & ((1u << BITS) - 1u);
    packed |= BITS == 3 ? v << (27 - 3 * i) : v << (8 - BITS * (i + 1));
  }
  out[tid] = (typename HqqFmt<BITS>::packed_t)packed;
}
__global__ void __launch_bounds__(256) hqq_copy_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (tid < n) out[tid] = in[tid];
}
template <int BITS, class IN> static void hqq_pack_launch(const void *in, void *out, size_t rows, size_t width, void *stream) {
  const size_t total = rows / HqqFmt<BITS>::P * width;
  if (total == 0) return;
  hipLaunchKernelGGL((hqq_pack_kernel<BITS, IN>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const IN *)in,
                     (typename HqqFmt<BITS>::packed_t *)out, rows, width);
}

}  // namespace mrs

#define MRS_HQQ_DEQUANT(bits, name)                                                                                                        \
  extern "C" void dequantize_##name##_f32(const void *wq, const void *scale, const void *zero, void *out, int h, int w) {                  \
    mrs::hqq_dequant_launch<bits, float>(wq, scale, zero, out, h, w);                                                                      \
  }                                                                                                                                        \
  extern "C" void dequantize_##name##_f16(const void *wq, const void *scale, const void *zero, void *out, int h, int w) {                  \
    mrs::hqq_dequant_launch<bits, mrs::f16_t>(wq, scale, zero, out, h, w);                                                                 \
  }                                                                                                                                        \
  extern "C" void dequantize_##name##_bf16(const void *wq, const void *scale, const void *zero, void *out, int h, int w) {                 \
    mrs::hqq_dequant_launch<bits, mrs::bf16_t>(wq, scale, zero, out, h, w);                                                                \
  }
MRS_HQQ_DEQUANT(8, 8bit_u8_kernel)
MRS_HQQ_DEQUANT(4, 4bit_u8_kernel)
MRS_HQQ_DEQUANT(2, 2bit_u8_kernel)
MRS_HQQ_DEQUANT(1, 1bit_u8_kernel)
MRS_HQQ_DEQUANT(3, 3bit_32_kernel)

extern "C" void launch_pack_1bit_kernel(const uint8_t *in, uint8_t *out, size_t num_input_elements, size_t input_width, void *stream) {
  mrs::hqq_pack_launch<1, uint8_t>(in, out, num_input_elements, input_width, stream);
}
extern "C" void launch_pack_2bit_kernel(const uint8_t *in, uint8_t *out, size_t num_input_elements, size_t input_width, void *stream) {
  mrs::hqq_pack_launch<2, uint8_t>(in, out, num_input_elements, input_width, stream);
}
extern "C" void launch_pack_3bit_kernel(const uint32_t *in, int32_t *out, size_t num_input_elements, size_t input_width, void *stream) {
  mrs::hqq_pack_launch<3, uint32_t>(in, out, num_input_elements, input_width, stream);
}
extern "C" void launch_pack_4bit_kernel(const uint8_t *in, uint8_t *out, size_t num_input_elements, size_t input_width, void *stream) {
  mrs::hqq_pack_launch<4, uint8_t>(in, out, num_input_elements, input_width, stream);
}
extern "C" void launch_pack_8bit_kernel(const uint8_t *in, uint8_t *out, size_t num_elements, void *stream) {
  if (num_elements == 0) return;
  hipLaunchKernelGGL(mrs::hqq_copy_kernel, dim3((unsigned)((num_elements + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, num_elements);
}
