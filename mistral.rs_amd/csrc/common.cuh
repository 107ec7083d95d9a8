// common.cuh -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
// Written for MI355X only: 64-lane wavefronts are assumed everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MRS_WAVE 64
#ifndef MRS_WAVE_SYNC
#define MRS_WAVE_SYNC() __builtin_amdgcn_wave_barrier() /* lanes of a wave exchange through LDS in lockstep; the host emulation maps this to a fiber sync */
#endif

namespace mrs {

// ---------------------------------------------------------------------------------- dtypes
struct bf16_t { uint16_t v; };
using f16_t = _Float16;

__device__ __forceinline__ float bf16_bits_to_float(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {  // round-to-nearest-even
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
  x += 0x7fffu + ((x >> 16) & 1u);
  return (uint16_t)(x >> 16);
}

// FP8 E4M3 (OCP "E4M3FN": bias 7, no infinities, S.1111.111 = NaN, max 448) -- the KV-cache element of cache_dtype 3.  Conversions are
// done in integer arithmetic so that the result is the defined one on every ASIC (the hardware fp8 converts differ between gfx942 FNUZ
// and gfx950 OCP): decode is exact; encode is round-to-nearest-even with saturation to +-448 (the reference's __NV_SATFINITE,
// quantization/fp8/nvidia/quant_utils.cuh:187-217), NaN -> 0x7f.
struct fp8_t { uint8_t v; };
__device__ __forceinline__ float fp8_e4m3_to_float(uint8_t v) {
  const uint32_t s = (uint32_t)(v & 0x80) << 24, e = (v >> 3) & 15, m = v & 7;
  if (e == 0) return __uint_as_float(s | __float_as_uint((float)m * 0.001953125f));  // subnormal: m * 2^-9
  if (e == 15 && m == 7) return __uint_as_float(s | 0x7fc00000u);
  return __uint_as_float(s | ((e + 120) << 23) | (m << 20));
}
__device__ __forceinline__ uint8_t float_to_fp8_e4m3(float x) {
  const uint32_t b = __float_as_uint(x), sign = (b >> 24) & 0x80;
  const float a = fabsf(x);
  if (!(a == a)) return (uint8_t)(sign | 0x7f);
  if (a >= 464.0f) return (uint8_t)(sign | 0x7e);           // >= halfway between 448 and the (absent) 480: saturate
  if (a < 0.015625f) return (uint8_t)(sign | (uint32_t)rintf(a * 512.0f));  // below 2^-6: subnormal grid of 2^-9 (8 -> the first normal)
  uint32_t r = __float_as_uint(a);
  r += 0x7ffffu + ((r >> 20) & 1u);                         // RNE to 3 mantissa bits
  const uint32_t code = (((r >> 23) - 120u) << 3) | ((r >> 20) & 7u);
  return (uint8_t)(sign | (code > 0x7eu ? 0x7eu : code));
}

template <class T> struct cvt;
template <> struct cvt<float> {
  static __device__ __forceinline__ float to_f(float x) { return x; }
  static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <> struct cvt<bf16_t> {
  static __device__ __forceinline__ float to_f(bf16_t x) { return bf16_bits_to_float(x.v); }
  static __device__ __forceinline__ bf16_t from_f(float x) { return bf16_t{float_to_bf16_bits(x)}; }
};
template <> struct cvt<f16_t> {
  static __device__ __forceinline__ float to_f(f16_t x) { return (float)x; }
  static __device__ __forceinline__ f16_t from_f(float x) { return (f16_t)x; }
};
template <class T> __device__ __forceinline__ float to_f(T x) { return cvt<T>::to_f(x); }
template <class T> __device__ __forceinline__ T from_f(float x) { return cvt<T>::from_f(x); }
// round a float through T (what `(dst_t)x` then `(float)` does in the reference kernels)
template <class T> __device__ __forceinline__ float round_to(float x) { return to_f<T>(from_f<T>(x)); }

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ uint16_t float_to_half_bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// ---------------------------------------------------------------------------------- wave ops
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Thread id the optimiser cannot look through: the persistent decode step runs ~20 phase bodies inside one loop; with a plain threadIdx.x
// hipcc hoists every phase's lane-constant address arithmetic out of that loop and keeps it live in VGPRs across all phases (spills).
#ifndef MRS_OPAQUE_TID
#define MRS_OPAQUE_TID(t) asm volatile("" : "+v"(t))
#endif
__device__ __forceinline__ int tid_opaque() { int t = (int)threadIdx.x; MRS_OPAQUE_TID(t); return t; }
__device__ __forceinline__ int lane_opaque() { return tid_opaque() & 63; }

template <class T> __device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, 64));
  return x;
}

// signed 4x int8 dot with int32 accumulate (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// 16-byte load with only 2-byte guaranteed alignment (GGUF blocks are 2-byte aligned: 34/210/110 B).
// gfx950 global memory tolerates unaligned dword accesses; the packed typedef makes hipcc emit
// one global_load_dwordx4 rather than byte loads.
typedef int int4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef int int2_a2 __attribute__((ext_vector_type(2), aligned(2)));
typedef int int1_a2 __attribute__((aligned(2)));
typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int int4_a16 __attribute__((ext_vector_type(4), aligned(16)));

__device__ __forceinline__ int4 ld16_a2(const void *p) { int4_a2 v = *(const int4_a2 *)p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int4 ld16_a4(const void *p) { int4_a4 v = *(const int4_a4 *)p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int4 ld16_a16(const void *p) { int4_a16 v = *(const int4_a16 *)p; return make_int4(v.x, v.y, v.z, v.w); }
// non-temporal ("streaming") variants for weight bytes that ONE wave reads once per token: keeps the stream from
// evicting the activations / KV that other waves re-read (MI355X: nt weight stream measured +10..15 % read rate)
__device__ __forceinline__ int4 ld16nt_a2(const void *p) { int4_a2 v = __builtin_nontemporal_load((const int4_a2 *)p); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int4 ld16nt_a4(const void *p) { int4_a4 v = __builtin_nontemporal_load((const int4_a4 *)p); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int ld4_a2(const void *p) { return *(const int1_a2 *)p; }
__device__ __forceinline__ uint16_t ld2(const void *p) { return *(const uint16_t *)p; }

// RoPE rotation of one pair: every product and the final sum round to T (the reference does the arithmetic
// in the tensor dtype, rotary.cu:9-33) and nothing is FMA-contracted, so the fused decode epilogue, the
// standalone rotary kernel and the CPU oracle agree bit-for-bit.  Results still need from_f<T>().
template <class T> __device__ __forceinline__ void rope_pair(float x, float y, float c, float s, float &xo, float &yo) {
#pragma clang fp contract(off)
  const float xc = round_to<T>(x * c), ys = round_to<T>(y * s), yc = round_to<T>(y * c), xs = round_to<T>(x * s);
  xo = xc - ys;
  yo = yc + xs;
}

// GLU activations -- codes as mistralrs-quant/src/utils/ops.rs:2601-2607 / mmvq_gguf.cu:44-85
__device__ __forceinline__ float glu_act(float x, int act) {
  switch (act) {
  case 1: { const float x3 = x * x * x; return 0.5f * x * (1.0f + tanhf(0.7978845608f * (x + 0.044715f * x3))); }
  case 2: return fmaxf(x, 0.0f);
  case 3: return x * 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  case 4: return 1.0f / (1.0f + expf(-x));
  default: return x / (1.0f + expf(-x));
  }
}

// The reference's Cephes-style exp of its CPU attention (mistralrs-core/src/attention/backends/cpu/elem.rs:417-433), operation for operation
// (the build pins -ffp-contract=off): the decode engine takes every exponential through it, so that a CPU evaluates the same bits
// (oracle/cpu_path_oracle.c orc_fast_exp).  f32::round = half away from zero.
__device__ __forceinline__ float fast_exp_ref(float x) {
  const float LOG2E = 1.44269504088896340736f, C0 = 0.6933594f, C1 = -2.1219444e-4f;
  x = fminf(fmaxf(x, -87.0f), 87.0f);
  const float zx = x * LOG2E;
  const float z = truncf(zx + copysignf(0.49999997f, zx));  // == roundf(zx) for every |zx| <= 129 (exhaustive: tests/test_oracle.py)
  const float r = x - z * C0 - z * C1;
  const float r2 = r * r;
  const float p = r + r2 * (0.5f + r * (0.16666546f + r * (0.041665795f + r * (0.00833345f + r * 0.0013920345f))));
  const float e = __uint_as_float((unsigned)(((int)z + 127) << 23));
  return e * (1.0f + p);
}
// SiLU of the decode engine: x / (1 + exp(-x)) (mistralrs-quant/src/utils/ops.rs:2601-2612) with the exponential above
__device__ __forceinline__ float silu_engine(float x) { return x / (1.0f + fast_exp_ref(-x)); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per KERNEL (keyed by its address: a `static bool` inside a generic lambda is shared by every
// kernel with the same function-pointer type); not an operation a stream capture tolerates on every call.  Host only.
inline void lds_attr_once(const void *kern, int bytes) {
  static const void *done[64];
  static int n = 0;
  for (int i = 0; i < n; ++i)
    if (done[i] == kern) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (n < 64) done[n++] = kern;
}

}  // namespace mrs
