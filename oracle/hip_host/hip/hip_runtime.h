// oracle/hip_host/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (never linked into the product libraries).
// A wave64 host stand-in for the slice of HIP / gfx950 device vocabulary that this repo's kernels use, so that the PRODUCT kernel sources
// (mistral.rs_amd/csrc/*.cuh, *.hip) can be compiled for the host (oracle/build_hip_host.sh -> oracle/_hiphost/*.so) and executed thread by
// thread before GPU time is spent on them: every thread of a workgroup is a fiber (ucontext), __syncthreads() is a workgroup barrier,
// cross-lane operations (DPP, readlane, shuffles, ballots) are exchanges between the 64 fibers of a wave, `__shared__` variables are
// function-local statics (one workgroup runs at a time), dynamic LDS is one global buffer, global memory is host memory.
// What it checks: indexing, layouts, barriers, lane-exchange patterns, arithmetic order (same f32 operations, contraction off).
// What it cannot check: occupancy, LDS capacity / bank conflicts, memory-model races between waves that the sequential schedule hides,
// the device's transcendental functions (libm here).  tests/test_hip_host_emulation.py compares the emulated launchers with the oracle.
#pragma once
#include <algorithm>
#include <functional>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static /* `extern __shared__ T name[];` lines are rewritten by build_hip_host.sh to point at hiphost::dyn_lds */

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
using std::max;
using std::min;

typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }  // "device" memory is host memory
enum { hipDeviceMallocUncached = 3, hipDeviceMallocFinegrained = 1 };
static inline hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : (hipError_t)2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s_, size_t n, int, hipStream_t) { memcpy(d, s_, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// ONE instance across the translation units of a library (C++17 inline variables): inline product functions such as mrs::lane_id() are
// merged by the linker and must see the same thread coordinates as the launch that runs them
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
static const int warpSize = 64;

static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __mul24(int a, int b) { return (int)((uint32_t)((a << 8) >> 8) * (uint32_t)((b << 8) >> 8)); }  // v_mul_i32_i24
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
template <class T> static inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }  // fibers run one at a time
template <class T> static inline T atomicMax(T *p, T v) { const T o = *p; if (v > o) *p = v; return o; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }  // v_rsq_f32 is a ~1 ulp approximation: tolerance-level agreement only
#define __expf(x) expf(x) /* glibc declares a non-static __expf */
static inline const char *hipGetErrorString(hipError_t) { return "hip error (host emulation)"; }

namespace hiphost {
constexpr int WAVE = 64, MAX_THREADS = 1024;
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = false; };
inline std::vector<Fiber> fibers;
inline ucontext_t main_ctx;
inline int cur = 0, live = 0, bar_count = 0, bar_gen = 0;
inline int wbar_count[MAX_THREADS / WAVE], wbar_gen[MAX_THREADS / WAVE];
inline uint64_t slot[MAX_THREADS];
inline std::function<void()> kernel;
inline char dyn_lds[160 * 1024] __attribute__((aligned(16)));
inline size_t dyn_lds_bytes = 0;  // size of the current launch (bounds are not enforced; launch() rejects > 160 KiB)

static inline int linear_tid() { return (int)((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x); }
static inline int nthreads() { return (int)(blockDim.x * blockDim.y * blockDim.z); }
static void yield() { swapcontext(&fibers[cur].ctx, &main_ctx); }
inline int wave_live[MAX_THREADS / WAVE];  // fibers of each wave that have not returned yet: a barrier / exchange waits for those only
static void entry() { kernel(); fibers[cur].done = true; --live; --wave_live[cur / WAVE]; }
static void block_barrier() {  // threads that already returned do not take part (as on the device)
  const int gen = bar_gen;
  ++bar_count;
  while (bar_gen == gen) {
    if (bar_count >= live) { bar_count = 0; ++bar_gen; break; }
    yield();
  }
}
static void wave_barrier(int w, int /*lanes*/) {
  const int gen = wbar_gen[w];
  ++wbar_count[w];
  while (wbar_gen[w] == gen) {
    if (wbar_count[w] >= wave_live[w]) { wbar_count[w] = 0; ++wbar_gen[w]; break; }
    yield();
  }
}
static inline int wave_lanes(int w) { const int n = nthreads() - w * WAVE; return n < WAVE ? n : WAVE; }
// every lane of the wave (convergent code) calls this together: returns the value lane `src_lane` passed
template <class T> static T exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "lane exchange payload");
  const int tid = linear_tid(), w = tid / WAVE, lanes = wave_lanes(w);
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  slot[tid] = bits;
  wave_barrier(w, lanes);
  T r;
  memcpy(&r, &slot[w * WAVE + (src_lane & (WAVE - 1))], sizeof(T));
  wave_barrier(w, lanes);
  return r;
}
static unsigned long long ballot(bool pred) {
  const int tid = linear_tid(), w = tid / WAVE, lanes = wave_lanes(w);
  slot[tid] = pred ? 1 : 0;
  wave_barrier(w, lanes);
  unsigned long long m = 0;
  for (int l = 0; l < lanes; ++l) m |= (unsigned long long)(slot[w * WAVE + l] & 1) << l;
  wave_barrier(w, lanes);
  return m;
}
// v_mov_b32_dpp source lane for the controls this repo uses (all of them permutations inside a row of 16 lanes, row_mask = bank_mask = 0xf)
static inline int dpp_src_lane(int lane, int ctrl) {
  const int row = lane & ~15, i = lane & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);  // quad_perm
  if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((i - (ctrl - 0x120)) & 15);                // row_ror:n  (lane i reads lane i - n)
  if (ctrl == 0x140) return row | (15 - i);                                                     // row_mirror
  if (ctrl == 0x141) return row | (i & 8) | (7 - (i & 7));                                      // row_half_mirror
  abort();  // a control this shim does not model
}
static inline int dpp(int v, int ctrl) { return exchange(v, dpp_src_lane(linear_tid() & (WAVE - 1), ctrl)); }

static void run_block(const dim3 &block, const std::function<void()> &k) {
  kernel = k;
  blockDim = block;
  const unsigned threads = block.x * block.y * block.z;
  if (threads > (unsigned)MAX_THREADS) abort();
  if (fibers.size() < threads) fibers.resize(threads);
  live = (int)threads; bar_count = 0;
  for (auto &c : wbar_count) c = 0;
  for (int w = 0; w < MAX_THREADS / WAVE; ++w) { const int n = (int)threads - w * WAVE; wave_live[w] = n < 0 ? 0 : (n < WAVE ? n : WAVE); }
  for (unsigned t = 0; t < threads; ++t) {
    Fiber &f = fibers[t];
    if (!f.stack) f.stack = (char *)malloc(512 * 1024);
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = 512 * 1024;
    f.ctx.uc_link = &main_ctx;
    makecontext(&f.ctx, (void (*)())entry, 0);
  }
  while (live > 0)
    for (unsigned t = 0; t < threads; ++t)
      if (!fibers[t].done) {
        cur = (int)t;
        threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
        swapcontext(&main_ctx, &fibers[t].ctx);
      }
}
static void launch(const dim3 &grid, const dim3 &block, size_t lds, const std::function<void()> &k) {
  if (lds > sizeof(dyn_lds)) abort();  // more dynamic LDS than a gfx950 workgroup can have
  dyn_lds_bytes = lds;
  gridDim = grid;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) { blockIdx.x = x; blockIdx.y = y; blockIdx.z = z; run_block(block, k); }
}
}  // namespace hiphost

static inline void __syncthreads() { hiphost::block_barrier(); }
// A wave executes in lockstep on the device, so LDS written by some lanes is visible to the others one instruction later WITHOUT a barrier; fibers are
// not in lockstep.  build_hip_host.sh inserts this at the two places of the product sources that rely on it (paged_attention.cuh, decode_attn_wave_kernel).
namespace hiphost { static inline void wave_sync() { wave_barrier(linear_tid() / WAVE, 0); } }
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return hiphost::exchange(v, (hiphost::linear_tid() & 63) ^ mask); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return hiphost::exchange(v, src); }
static inline unsigned long long __ballot(int pred) { return hiphost::ballot(pred != 0); }
template <class T> static inline T __shfl_down(T v, int delta, int width = 64) {  /* a lane whose source falls outside its `width` segment keeps its own value */
  const int lane = hiphost::linear_tid() & 63;
  return hiphost::exchange(v, (lane % width) + delta < width ? lane + delta : lane);
}
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }


// ---------------------------------------------------------------------------------------------- matrix cores (gfx950 v_mfma_f32_32x32x16_bf16)
// D = A (32 x 16) * B (16 x 32) + C, one wave: lane l holds A[l % 32][8 * (l / 32) + j], B[8 * (l / 32) + j][l % 32] (j = 0..7) and the 16
// accumulators C[8 * (i / 4) + 4 * (l / 32) + i % 4][l % 32] (i = 0..15).  Products are exact in f32; the sum over k is taken in ascending k
// with plain f32 adds (the hardware's internal order / fusing is unspecified: tolerance-level agreement).  The layout is CALIBRATED, not assumed:
// the GPU-green prefill GEMM / prefill attention tests pass on this model (`pytest --host-emulation`).
typedef short hiphost_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hiphost_f32x16 __attribute__((ext_vector_type(16)));
namespace hiphost {
inline uint16_t mfma_a[MAX_THREADS][8], mfma_b[MAX_THREADS][8];
static inline float bf16_bits_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static hiphost_f32x16 mfma_32x32x16_bf16(hiphost_bf16x8 a, hiphost_bf16x8 b, hiphost_f32x16 c) {
  const int tid = linear_tid(), w = tid / WAVE, lane = tid & (WAVE - 1), base = w * WAVE;
  for (int j = 0; j < 8; ++j) { mfma_a[tid][j] = (uint16_t)a[j]; mfma_b[tid][j] = (uint16_t)b[j]; }
  wave_barrier(w, 0);
  const int col = lane & 31, hi = lane >> 5;
  hiphost_f32x16 d = c;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i / 4) + 4 * hi + (i % 4);
    float acc = 0.0f;
    for (int k = 0; k < 16; ++k) acc += bf16_bits_f(mfma_a[base + row + 32 * (k / 8)][k % 8]) * bf16_bits_f(mfma_b[base + col + 32 * (k / 8)][k % 8]);
    d[i] = c[i] + acc;
  }
  wave_barrier(w, 0);
  return d;
}
}  // namespace hiphost
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, CBSZ, ABID, BLGP) hiphost::mfma_32x32x16_bf16((A), (B), (C))
// v_mfma_f32_32x32x16_f16: same lane layout as the bf16 form, f16 operands.  Products of the small integers ext_gemm_qi.hip feeds it are exact and every
// partial sum stays below 2^24, so the order of the k sum cannot matter (on the device tests/test_gemm_qi.py checks exactly that on adversarial operands).
typedef _Float16 hiphost_f16x8 __attribute__((ext_vector_type(8)));
namespace hiphost {
inline _Float16 mfma_ah[MAX_THREADS][8], mfma_bh[MAX_THREADS][8];
static hiphost_f32x16 mfma_32x32x16_f16(hiphost_f16x8 a, hiphost_f16x8 b, hiphost_f32x16 c) {
  const int tid = linear_tid(), w = tid / WAVE, lane = tid & (WAVE - 1), base = w * WAVE;
  for (int j = 0; j < 8; ++j) { mfma_ah[tid][j] = a[j]; mfma_bh[tid][j] = b[j]; }
  wave_barrier(w, 0);
  const int col = lane & 31, hi = lane >> 5;
  hiphost_f32x16 d = c;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i / 4) + 4 * hi + (i % 4);
    float acc = c[i];
    for (int k = 0; k < 16; ++k) acc += (float)mfma_ah[base + row + 32 * (k / 8)][k % 8] * (float)mfma_bh[base + col + 32 * (k / 8)][k % 8];
    d[i] = acc;
  }
  wave_barrier(w, 0);
  return d;
}
}  // namespace hiphost
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, CBSZ, ABID, BLGP) hiphost::mfma_32x32x16_f16((A), (B), (C))
// v_mfma_f32_32x32x2_f32 (f32 operands, one value per lane: lane l holds A[l % 32][l / 32] and B[l / 32][l % 32]): D = C + A B as the f32 FMA chain over k --
// d = fmaf(a1, b1, fmaf(a0, b0, c)) -- which is what the MI355X computes bit for bit (profiles/experiments/mfma_f32_probe.hip, round 6; the exact prompt attention of
// ext_gemm_qi.hip relies on it to reproduce the decode kernel's fmaf chains on the matrix cores).  HIPHOST_MFMA_F32_K1_FIRST flips the order (the probe decides).
namespace hiphost {
inline float mfma_af[MAX_THREADS], mfma_bf[MAX_THREADS];
static hiphost_f32x16 mfma_32x32x2_f32(float a, float b, hiphost_f32x16 c) {
  const int tid = linear_tid(), w = tid / WAVE, lane = tid & (WAVE - 1), base = w * WAVE;
  mfma_af[tid] = a; mfma_bf[tid] = b;
  wave_barrier(w, 0);
  const int col = lane & 31, hi = lane >> 5;
  hiphost_f32x16 d = c;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i / 4) + 4 * hi + (i % 4);
#ifdef HIPHOST_MFMA_F32_K1_FIRST
    d[i] = fmaf(mfma_af[base + row], mfma_bf[base + col], fmaf(mfma_af[base + row + 32], mfma_bf[base + col + 32], c[i]));
#else
    d[i] = fmaf(mfma_af[base + row + 32], mfma_bf[base + col + 32], fmaf(mfma_af[base + row], mfma_bf[base + col], c[i]));
#endif
  }
  wave_barrier(w, 0);
  return d;
}
}  // namespace hiphost
#define __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, CBSZ, ABID, BLGP) hiphost::mfma_32x32x2_f32((A), (B), (C))
// v_mfma_i32_32x32x32_i8: D = A (32 x 32 int8) * B (32 x 32 int8) + C, exact in int32.  Lane l holds A[l % 32][16 * (l / 32) + j] and
// B[16 * (l / 32) + j][l % 32] (j = 0..15, 16 consecutive bytes); C / D as every 32 x 32 MFMA (the layout of the accumulators is dtype-independent).
// Which 16 k a lane half holds does not change the result as long as A and B agree -- they do by symmetry.
typedef int hiphost_i32x4 __attribute__((ext_vector_type(4)));
typedef int hiphost_i32x16 __attribute__((ext_vector_type(16)));
namespace hiphost {
inline int8_t mfma_a8[MAX_THREADS][16], mfma_b8[MAX_THREADS][16];
static hiphost_i32x16 mfma_32x32x32_i8(hiphost_i32x4 a, hiphost_i32x4 b, hiphost_i32x16 c) {
  const int tid = linear_tid(), w = tid / WAVE, lane = tid & (WAVE - 1), base = w * WAVE;
  memcpy(mfma_a8[tid], &a, 16);
  memcpy(mfma_b8[tid], &b, 16);
  wave_barrier(w, 0);
  const int col = lane & 31, hi = lane >> 5;
  hiphost_i32x16 d = c;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i / 4) + 4 * hi + (i % 4);
    int acc = 0;
    for (int k = 0; k < 32; ++k) acc += (int)mfma_a8[base + row + 32 * (k / 16)][k % 16] * (int)mfma_b8[base + col + 32 * (k / 16)][k % 16];
    d[i] = c[i] + acc;
  }
  wave_barrier(w, 0);
  return d;
}
}  // namespace hiphost
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, C, CBSZ, ABID, BLGP) hiphost::mfma_32x32x32_i8((A), (B), (C))
// v_mfma_i32_32x32x16_i8 (the CDNA3 form, kept on gfx950): 8 bytes per lane, k = 8 * (l / 32) + j
namespace hiphost {
static hiphost_i32x16 mfma_32x32x16_i8(long a, long b, hiphost_i32x16 c) {
  const int tid = linear_tid(), w = tid / WAVE, lane = tid & (WAVE - 1), base = w * WAVE;
  memcpy(mfma_a8[tid], &a, 8);
  memcpy(mfma_b8[tid], &b, 8);
  wave_barrier(w, 0);
  const int col = lane & 31, hi = lane >> 5;
  hiphost_i32x16 d = c;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i / 4) + 4 * hi + (i % 4);
    int acc = 0;
    for (int k = 0; k < 16; ++k) acc += (int)mfma_a8[base + row + 32 * (k / 8)][k % 8] * (int)mfma_b8[base + col + 32 * (k / 8)][k % 8];
    d[i] = c[i] + acc;
  }
  wave_barrier(w, 0);
  return d;
}
}  // namespace hiphost
#define __builtin_amdgcn_mfma_i32_32x32x16_i8(A, B, C, CBSZ, ABID, BLGP) hiphost::mfma_32x32x16_i8((A), (B), (C))
// buffer resource: base + byte range; raw_buffer_load returns zeros out of range (as the hardware's bounds check does)
struct __amdgpu_buffer_rsrc_t { const char *base; unsigned bytes; };
static inline __amdgpu_buffer_rsrc_t hiphost_make_rsrc(void *p, int bytes) { return __amdgpu_buffer_rsrc_t{(const char *)p, (unsigned)bytes}; }
#define __builtin_amdgcn_make_buffer_rsrc(P, STRIDE, NUM, FLAGS) hiphost_make_rsrc((P), (NUM))
typedef unsigned hiphost_v4u __attribute__((ext_vector_type(4)));
static inline hiphost_v4u hiphost_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  hiphost_v4u v = {0, 0, 0, 0};
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 16 <= r.bytes) memcpy(&v, r.base + o, 16);
  return v;
}
#define __builtin_amdgcn_raw_buffer_load_b128(R, VOFF, SOFF, AUX) hiphost_raw_buffer_load_b128((R), (VOFF), (SOFF))
static inline unsigned hiphost_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  unsigned v = 0;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 4 <= r.bytes) memcpy(&v, r.base + o, 4);
  return v;
}
static inline short hiphost_raw_buffer_load_b16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  short v = 0;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 2 <= r.bytes) memcpy(&v, r.base + o, 2);
  return v;
}
#define __builtin_amdgcn_raw_buffer_load_b32(R, VOFF, SOFF, AUX) hiphost_raw_buffer_load_b32((R), (VOFF), (SOFF))
typedef unsigned hiphost_v2u __attribute__((ext_vector_type(2)));
static inline hiphost_v2u hiphost_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  hiphost_v2u v = {0, 0};
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 8 <= r.bytes) memcpy(&v, r.base + o, 8);
  return v;
}
#define __builtin_amdgcn_raw_buffer_load_b64(R, VOFF, SOFF, AUX) hiphost_raw_buffer_load_b64((R), (VOFF), (SOFF))
#define __builtin_amdgcn_raw_buffer_load_b16(R, VOFF, SOFF, AUX) hiphost_raw_buffer_load_b16((R), (VOFF), (SOFF))
// f32 -> bf16 (RNE) for `__builtin_convertvector(float2, __bf16 x 2)`: x86 lowers it to this runtime call
extern "C" inline __bf16 __truncsfbf2(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint16_t h;
  if ((x & 0x7fffffffu) > 0x7f800000u) h = (uint16_t)((x >> 16) | 0x40);
  else { x += 0x7fffu + ((x >> 16) & 1u); h = (uint16_t)(x >> 16); }
  __bf16 r; memcpy(&r, &h, 2); return r;
}

#define hipLaunchKernelGGL(K, G, B, LDS, STREAM, ...) hiphost::launch(dim3(G), dim3(B), (size_t)(LDS), [&] { K(__VA_ARGS__); })
#define __builtin_amdgcn_update_dpp(OLD, SRC, CTRL, RMASK, BMASK, BOUND) hiphost::dpp((SRC), (CTRL))
namespace hiphost { inline unsigned long long fake_realtime() { static thread_local unsigned long long t = 0; return t += 1000; } }  /* advances per read: time-bounded spins end */
#define __builtin_amdgcn_s_memrealtime() hiphost::fake_realtime()
#define __builtin_amdgcn_readlane(V, L) hiphost::exchange((int)(V), (L))
#define __builtin_amdgcn_readfirstlane(V) hiphost::exchange((int)(V), 0) /* kernels use it on wave-uniform values only */
#define __builtin_amdgcn_sched_group_barrier(MASK, SIZE, SYNC) ((void)0) /* instruction-order hint only */
#define __builtin_amdgcn_sched_barrier(MASK) ((void)0)                  /* instruction-order hint only */
#define MRS_OPAQUE_TID(t) ((void)0)
// LDS-DMA (global_load_lds): LDS destination = wave-uniform base + lane * size, source address per lane; synchronous here
#define MRS_GLDS16(gptr, lbase) memcpy((char *)(lbase) + (hiphost::linear_tid() & 63) * 16, (const void *)(gptr), 16)
#define MRS_GLDS4(gptr, lbase) memcpy((char *)(lbase) + (hiphost::linear_tid() & 63) * 4, (const void *)(gptr), 4)
#define MRS_WAIT_VMCNT0() ((void)0)                                     /* product code: s_waitcnt vmcnt(0) */
#define __builtin_amdgcn_fence(ORDER, SCOPE) ((void)0)                  /* one workgroup runs at a time: nothing to order */
#define __builtin_amdgcn_s_sleep(N) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_fetch_add(P, V, ORDER, SCOPE) atomicAdd((P), (V))
static inline void __threadfence() {}
#define __hip_atomic_load(P, ORDER, SCOPE) (*(P))
#define __hip_atomic_store(P, V, ORDER, SCOPE) (*(P) = (V))
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipDeviceAttributeMultiprocessorCount = 63, hipIpcMemLazyEnablePeerAccess = 1 };
#define __HIP_MEMORY_SCOPE_SYSTEM 5
struct hipIpcMemHandle_t { char reserved[64]; };
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) { memset(h, 0, sizeof *h); memcpy(h, &p, sizeof p); return hipSuccess; }  /* one address space */
static inline hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned) { memcpy(p, &h, sizeof *p); return hipSuccess; }
static inline hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s_, size_t n, int) { memcpy(d, s_, n); return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 4; return hipSuccess; }  /* the emulated "chip": 4 CUs (persistent step: 4 workgroups) */
#define MRS_WAVE_SYNC() hiphost::wave_sync()                           /* product code: a compiler-level wave barrier (lockstep lanes) */
#define __builtin_amdgcn_sdot4(A, B, C, CLAMP) hiphost_sdot4((A), (B), (C))
static inline int hiphost_sdot4(int a, int b, int c) {
  for (int i = 0; i < 4; ++i) c += (int)(int8_t)((uint32_t)a >> (8 * i)) * (int)(int8_t)((uint32_t)b >> (8 * i));
  return c;
}
