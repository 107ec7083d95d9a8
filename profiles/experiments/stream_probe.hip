// Streaming-read ceiling experiment (not product code): how fast can 256 CUs read a packed Q4_K-shaped tensor
// with (a) the GEMV's access pattern and no arithmetic, (b) plain 16 B/lane contiguous loads, (c) nt loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ void __launch_bounds__(256) stream_k(const uint8_t *__restrict__ w, size_t nbytes, int *sink, int waves_total) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t per = (nbytes / waves_total) & ~(size_t)1023;  // contiguous byte range per wave
  const uint8_t *p = w + (size_t)gw * per;
  v4i acc = {0, 0, 0, 0};
  if (MODE == 3 || MODE == 4) {  // GEMV-like prologue: stage 4.6 KB from global into (dynamic) LDS behind a barrier
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const v4i *src = (const v4i *)(w + nbytes - 8192);
    v4i *dst = (v4i *)smem;
    for (int i = threadIdx.x; i < 288; i += 256) dst[i] = src[i];
    __syncthreads();
    acc = dst[(threadIdx.x * 7) % 288];
  }
  if (MODE == 4) {  // + per-row epilogue: wave reduction + single-lane store every 2 KB
    const size_t n16 = per / 16;
    for (size_t i0 = 0; i0 < n16; i0 += 128) {
      v4i q0 = __builtin_nontemporal_load((const v4i *)(p + (i0 + lane) * 16));
      v4i q1 = __builtin_nontemporal_load((const v4i *)(p + (i0 + 64 + lane) * 16));
      int v = q0.x ^ q0.y ^ q0.z ^ q0.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w ^ acc.x;
      for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) sink[8 + gw * 64 + (int)(i0 / 128) % 64] = v;
    }
    return;
  }
  if (MODE == 0) {  // GEMV pattern: 144-byte blocks, 8 lanes per block: header (16 B, shared) + 16 B of qs
    const size_t nblk = per / 144;
    for (size_t b0 = 0; b0 < nblk; b0 += 8 * U) {
      v4i h[U], q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t b = b0 + u * 8 + (lane >> 3);
        if (b >= nblk) b = nblk - 1;
        const uint8_t *blk = p + b * 144;
        h[u] = *(const v4i *)blk;
        q[u] = *(const v4i *)(blk + 16 + (lane & 7) * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= h[u] ^ q[u];
    }
  } else {
    const size_t n16 = per / 16;
    for (size_t i0 = 0; i0 < n16; i0 += 64 * U) {
      v4i q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t i = i0 + u * 64 + lane;
        if (i >= n16) i = n16 - 1;
        if (MODE == 1) q[u] = *(const v4i *)(p + i * 16);
        else q[u] = __builtin_nontemporal_load((const v4i *)(p + i * 16));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= q[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
}
extern "C" void stream_launch(int mode, int u, const void *w, size_t nbytes, int *sink, int wgs, void *stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(M, UU) hipLaunchKernelGGL((stream_k<M, UU>), dim3(wgs), dim3(256), (M >= 3 ? 8192 : 0), s, (const uint8_t *)w, nbytes, sink, wgs * 4)
  if (mode == 0) { if (u == 2) L(0, 2); else if (u == 4) L(0, 4); else L(0, 8); }
  else if (mode == 1) { if (u == 2) L(1, 2); else if (u == 4) L(1, 4); else L(1, 8); }
  else if (mode == 2) { if (u == 2) L(2, 2); else if (u == 4) L(2, 4); else L(2, 8); }
  else if (mode == 3) { L(3, 4); }
  else { L(4, 4); }
}
