// oracle/ref_shim/cuda_shim.h -- TEST INFRASTRUCTURE ONLY.
// Minimal host stand-ins for the CUDA device vocabulary used by the reference's *device functions*
// (block decode + MMVQ vec_dot in mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu, format spec in
// kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu), so that oracle/build_ref.sh can compile those functions
// FROM THE REFERENCE TREE (streamed into g++, never copied into this repo) into oracle/_ref/*.so and the
// tests can pin the oracle against the reference's own arithmetic.  Nothing here is product code.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __align__(n) alignas(n)
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct float2 { float x, y; };
// thread coordinates for the reference __global__ kernels that hqq_driver.inc runs one thread at a time
struct ShimDim { unsigned x, y, z; };
static thread_local ShimDim blockIdx, blockDim, threadIdx;
#include <stddef.h>

static inline float shim_h2f(uint16_t h) {  // IEEE binary16 -> binary32, exact
  const uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
  uint32_t b;
  if (e == 0) {
    if (m == 0) b = s;
    else { int sh = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; ++sh; } b = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13); }
  } else if (e == 31) b = s | 0x7f800000u | (m << 13);
  else b = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &b, 4); return f;
}
static inline uint16_t shim_f2h(float f) {  // binary32 -> IEEE binary16, round to nearest even (what `(half)x` does on the device)
  uint32_t b; memcpy(&b, &f, 4);
  const uint32_t s = (b >> 16) & 0x8000u;
  const uint32_t a = b & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);             // NaN
  if (a >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);            // >= 65520 -> inf
  if (a < 0x33000001u) return (uint16_t)s;                         // <= 2^-25 -> 0 (ties to even)
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  if (e < -14) {                                                    // subnormal half
    const int sh = -14 - e + 13;                                   // bits to drop from the 24-bit significand
    const uint32_t q = m >> sh, r = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    return (uint16_t)(s | (q + ((r > half) || (r == half && (q & 1u)))));
  }
  const uint32_t q = m >> 13, r = m & 0x1fffu;
  uint32_t h = ((uint32_t)(e + 15) << 10) + (q & 0x3ffu) + ((r > 0x1000u) || (r == 0x1000u && (q & 1u)));
  return (uint16_t)(s | h);                                         // a mantissa carry bumps the exponent correctly
}
struct __half {
  uint16_t bits;
  __half() = default;
  __half(float f) : bits(shim_f2h(f)) {}  // implicit, like the device type
  operator float() const { return shim_h2f(bits); }
};
#ifdef SHIM_HALF_OPS  /* device `__half` arithmetic: every + - * rounds to half (sm_53+ native ops are correctly rounded) */
static inline __half operator+(__half a, __half b) { return __half((float)a + (float)b); }
static inline __half operator-(__half a, __half b) { return __half((float)a - (float)b); }
static inline __half operator*(__half a, __half b) { return __half((float)a * (float)b); }
#endif
struct __half2 { __half x, y; };
typedef __half half;
typedef __half2 half2;
static inline float __half2float(__half h) { return shim_h2f(h.bits); }
static inline __half __low2half(__half2 v) { return v.x; }
static inline __half __high2half(__half2 v) { return v.y; }
static inline float __low2float(__half2 v) { return shim_h2f(v.x.bits); }
static inline float __high2float(__half2 v) { return shim_h2f(v.y.bits); }
static inline float2 __half22float2(__half2 v) { return float2{shim_h2f(v.x.bits), shim_h2f(v.y.bits)}; }
// per-byte signed saturating subtract (CUDA SIMD intrinsic)
static inline int __vsubss4(int a, int b) {
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    int d = (int)(int8_t)((uint32_t)a >> (8 * i)) - (int)(int8_t)((uint32_t)b >> (8 * i));
    d = d > 127 ? 127 : (d < -128 ? -128 : d);
    r |= (uint32_t)(uint8_t)(int8_t)d << (8 * i);
  }
  return (int)r;
}
#ifndef SHIM_FIBERS  /* libref_mmvq: only referenced by helpers its driver never calls; fiber_shim.h provides the real exchange */
template <class T> static inline T __shfl_xor_sync(unsigned, T v, int, int = 32) { return v; }
#endif
static inline float normcdff(float x) { return 0.5f * erfcf(-x * 0.70710678118654752440f); }
