// oracle/ref_shim/fiber_shim.h -- TEST INFRASTRUCTURE ONLY.
// Runs a CUDA thread block on the host: every thread of the block is a fiber (ucontext), __syncthreads() is a block barrier,
// __shfl_xor_sync / __shfl_sync are warp (32 lanes) exchanges, `__shared__` variables are function-local statics (one block runs at a
// time), the dynamic shared memory is one global buffer.  Used by oracle/build_ref.sh to execute the reference's own
// paged_attention_v1 / v2 / v2_reduce kernels (mistralrs-paged-attn/src/cuda/pagedattention.cuh:56-667, f32 instantiation) so that the
// oracle's attention can be pinned to them (tests/test_oracle_ref.py).  Nothing here is product code.
#pragma once
#include <assert.h>
#include <float.h>
#include <functional>
#include <stdlib.h>
#include <ucontext.h>
#include <vector>

struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static thread_local ShimDim gridDim;
#define __expf(x) expf(x)          /* CUDA fast-math intrinsics: evaluated exactly on the host */
#define __fdividef(a, b) ((a) / (b))
#define __shared__ static

namespace shim_fiber {
constexpr int WARP = 32, MAX_THREADS = 1024;
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = false; };
static std::vector<Fiber> fibers;
static ucontext_t main_ctx;
static int cur = 0, live = 0, bar_count = 0, bar_gen = 0;
static int wbar_count[MAX_THREADS / WARP], wbar_gen[MAX_THREADS / WARP];
static uint64_t slot[MAX_THREADS];
static std::function<void()> kernel;
static char dyn_smem[1 << 20] __attribute__((aligned(16)));

static void yield() { swapcontext(&fibers[cur].ctx, &main_ctx); }
static void entry() { kernel(); fibers[cur].done = true; --live; }  // uc_link returns to the scheduler
static void block_barrier() {
  const int gen = bar_gen;
  if (++bar_count >= live) { bar_count = 0; ++bar_gen; return; }
  while (bar_gen == gen) yield();
}
static void warp_barrier(int w, int lanes) {
  const int gen = wbar_gen[w];
  if (++wbar_count[w] >= lanes) { wbar_count[w] = 0; ++wbar_gen[w]; return; }
  while (wbar_gen[w] == gen) yield();
}
template <class T> static T exchange(T v, int src_lane_of) {  // all live lanes of the warp call this together
  static_assert(sizeof(T) <= 8, "shuffle payload");
  const int nthreads = (int)(blockDim.x * blockDim.y);
  const int tid = (int)(threadIdx.y * blockDim.x + threadIdx.x), w = tid / WARP;  // linear thread id: warps are 32 consecutive threads
  const int lanes = nthreads - w * WARP < WARP ? nthreads - w * WARP : WARP;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  slot[tid] = bits;
  warp_barrier(w, lanes);
  T r;
  memcpy(&r, &slot[w * WARP + src_lane_of], sizeof(T));
  warp_barrier(w, lanes);
  return r;
}
// run `k` for every thread of one block (blockIdx / gridDim are set by the caller)
static void run_block(unsigned dimx, unsigned dimy, const std::function<void()> &k) {
  kernel = k;
  const unsigned threads = dimx * dimy;
  blockDim.x = dimx; blockDim.y = dimy; blockDim.z = 1;
  if (fibers.size() < threads) fibers.resize(threads);
  live = (int)threads; bar_count = 0;
  for (auto &c : wbar_count) c = 0;
  for (unsigned t = 0; t < threads; ++t) {
    Fiber &f = fibers[t];
    if (!f.stack) f.stack = (char *)malloc(256 * 1024);
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = 256 * 1024;
    f.ctx.uc_link = &main_ctx;
    makecontext(&f.ctx, (void (*)())entry, 0);
  }
  while (live > 0)
    for (unsigned t = 0; t < threads; ++t)
      if (!fibers[t].done) { cur = (int)t; threadIdx.x = t % dimx; threadIdx.y = t / dimx; threadIdx.z = 0; swapcontext(&main_ctx, &fibers[t].ctx); }
}
static void run_block(unsigned threads, const std::function<void()> &k) { run_block(threads, 1, k); }
}  // namespace shim_fiber

static inline void __syncthreads() { shim_fiber::block_barrier(); }
template <class T> static inline T shim_shfl_xor(T v, int mask) { return shim_fiber::exchange(v, (int)((threadIdx.y * blockDim.x + threadIdx.x) % 32) ^ mask); }
template <class T> static inline T shim_shfl(T v, int src) { return shim_fiber::exchange(v, src); }
template <class T> static inline T __shfl_xor_sync(unsigned, T v, int mask, int = 32) { return shim_shfl_xor(v, mask); }
template <class T> static inline T __shfl_down_sync(unsigned, T v, int delta, int = 32) {
  const int lane = (int)((threadIdx.y * blockDim.x + threadIdx.x) % 32);
  return shim_fiber::exchange(v, lane + delta < 32 ? lane + delta : lane);  // lanes past the end keep their own value
}
// bf16 stand-in (RNE from f32, exact widening) and the remaining half helpers for kernels templated on the 16-bit types
struct __nv_bfloat16 {
  uint16_t bits;
  __nv_bfloat16() = default;
  __nv_bfloat16(float f) { uint32_t b; memcpy(&b, &f, 4); if ((b & 0x7fffffffu) > 0x7f800000u) bits = (uint16_t)((b >> 16) | 0x40); else { b += 0x7fffu + ((b >> 16) & 1u); bits = (uint16_t)(b >> 16); } }
  operator float() const { uint32_t b = (uint32_t)bits << 16; float f; memcpy(&f, &b, 4); return f; }
};
#ifdef SHIM_HALF_OPS
static inline __nv_bfloat16 operator+(__nv_bfloat16 a, __nv_bfloat16 b) { return __nv_bfloat16((float)a + (float)b); }
static inline __nv_bfloat16 operator-(__nv_bfloat16 a, __nv_bfloat16 b) { return __nv_bfloat16((float)a - (float)b); }
static inline __nv_bfloat16 operator*(__nv_bfloat16 a, __nv_bfloat16 b) { return __nv_bfloat16((float)a * (float)b); }
#endif
static inline float __bfloat162float(__nv_bfloat16 v) { return (float)v; }
static inline __nv_bfloat16 __float2bfloat16(float f) { return __nv_bfloat16(f); }
static inline __half __float2half(float f) { return __half(f); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }  /* the device intrinsic is a 2-ulp approximation: compare to f32 tolerance */
#include <algorithm>
#include <type_traits>
#define VLLM_LDG(arg) *(arg)
#define VLLM_SHFL_XOR_SYNC(var, lane_mask) shim_shfl_xor(var, lane_mask)
#define VLLM_SHFL_SYNC(var, src_lane) shim_shfl(var, src_lane)
#ifndef WARP_SIZE
#define WARP_SIZE 32
#endif
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define DIVIDE_ROUND_UP(a, b) (((a) + (b) - 1) / (b))
