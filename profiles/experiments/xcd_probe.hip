// xcd_probe.hip -- round 6 measurement (not product code): the two hand-off edges an "XCD as tensor-parallel rank" persistent decode layer would be made of
// (VERDICT round 5, item 4), timed inside ONE persistent launch of 256 workgroups while the other waves of every workgroup stream weights with `nt` loads:
//   edge L (XCD-local, 3 per layer: qkv -> attention, attention -> o_proj, gate/up -> down): the 32 workgroups of an XCD each publish an 8-byte {data, tag} granule
//           (sc1 store) into their XCD's slab and wait until all 32 granules of the slab carry this round's tag (sc1 loads, one lane per granule);
//   edge G (cross-XCD, 2 per layer: the sums behind the row-parallel o_proj / down): two hops --
//           hop 1: workgroup (x, j) publishes its 128 partial values of XCD x's partial vector as 128 granules (1 KiB); workgroup (x, j) then waits for rows
//                  [128 j + 16 x, + 16) of all 8 XCDs' partials (8 x 16 granules), adds them in XCD order;
//           hop 2: it publishes the 16 reduced values (16 granules) and every workgroup sweeps all 4096 reduced granules (32 KiB): the all-gather of the guide's
//                  price list (MI355X_MICROARCH.md "allgather": 4.2 us streaming / 2.9 parked for 32 KB).
// Times: s_memrealtime (100 MHz) around R rounds on workgroup 0's sync wave, and the stream's byte rate with and without the edges.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_probe xcd_probe.hip && ./xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
  const unsigned char *w; unsigned w_bytes; unsigned per_wave;  // weight stream: every streaming wave reads per_wave bytes per pass, passes repeat until the sync wave is done
  unsigned long long *slabL;   // [8][32] granules
  unsigned long long *part;    // [8][4096] granules (hop 1)
  unsigned long long *red;     // [4096] granules (hop 2)
  unsigned long long *times;   // [256][4]: realtime at start / end, bytes streamed (per workgroup), spins
  int *xcc_count;              // [8] census: workgroups per XCD (slot inside the XCD = arrival order)
  int mode, rounds, timeout_ticks;
};

__device__ __forceinline__ void st_granule(unsigned long long *p, unsigned data, unsigned tag) {
  const unsigned long long g = ((unsigned long long)tag << 32) | data;
  __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(256) probe(const Args a) {
  __shared__ int stop, slot_s, xcc_s;
  __shared__ unsigned long long streamed[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    stop = 0;
    const int xcc = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7);  // HW_REG_XCC_ID (id 20), bits 3:0
    xcc_s = xcc;
    slot_s = atomicAdd(a.xcc_count + xcc, 1);
  }
  __syncthreads();
  const int xcc = xcc_s, slot = slot_s;
  if (wave > 0) {  // ---------------- streaming waves: 4 x 1 KiB nt loads in flight each, until the sync wave is done
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, (short)0, (int)a.w_bytes, 0x00020000);
    const unsigned base = ((blockIdx.x * 3 + (wave - 1)) * a.per_wave) % (a.w_bytes - a.per_wave);
    unsigned off = 0, acc = 0;
    unsigned long long bytes = 0;
    v4u r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off + lane * 16, 0, 2); off = (off + 1024) % a.per_wave; }
    while (!*(volatile int *)&stop) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off + lane * 16, 0, 2);
        off = (off + 1024) % a.per_wave;
      }
      bytes += 4096;
    }
    if (lane == 0) streamed[wave] = bytes + (acc == 0x1234567u);
    __syncthreads();
    return;
  }
  // ---------------- the sync wave
  const bool ok_xcd = slot < 32;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), spins = 0;
  bool timed_out = false;
  for (int r = 1; r <= a.rounds && !timed_out; ++r) {
    const unsigned long long tr = __builtin_amdgcn_s_memrealtime();
    auto expired = [&]() { return (long long)(__builtin_amdgcn_s_memrealtime() - tr) > a.timeout_ticks; };
    if (a.mode == 1 && ok_xcd) {  // edge L
      if (lane == 0) st_granule(a.slabL + xcc * 32 + slot, (unsigned)blockIdx.x, (unsigned)r);
      bool done = false;
      while (!done) {
        const unsigned long long g = lane < 32 ? ld_granule(a.slabL + xcc * 32 + lane) : ((unsigned long long)r << 32);
        done = __all((unsigned)(g >> 32) >= (unsigned)r);
        ++spins;
        if (!done) { __builtin_amdgcn_s_sleep(2); if (expired()) { timed_out = true; break; } }
      }
    } else if (a.mode == 2 && ok_xcd) {  // edge G
      // hop 1: publish my 128 partial values of XCD xcc's partial vector (rows 128 slot .. + 127): 2 granules per lane
      st_granule(a.part + (size_t)xcc * 4096 + slot * 128 + lane, (unsigned)lane, (unsigned)r);
      st_granule(a.part + (size_t)xcc * 4096 + slot * 128 + 64 + lane, (unsigned)lane, (unsigned)r);
      // wait for rows [128 slot + 16 xcc, + 16) of all 8 partials: 128 granules = 2 per lane (lane -> (x = lane / 8 [+ 4 for the second], row = lane % 16 ...))
      float sum = 0.f;
      bool done = false;
      while (!done) {
        const int x0 = lane >> 4, rr = lane & 15;
        const unsigned long long g0 = ld_granule(a.part + (size_t)x0 * 4096 + slot * 128 + 16 * xcc + rr);
        const unsigned long long g1 = ld_granule(a.part + (size_t)(x0 + 4) * 4096 + slot * 128 + 16 * xcc + rr);
        done = __all((unsigned)(g0 >> 32) >= (unsigned)r && (unsigned)(g1 >> 32) >= (unsigned)r);
        sum = __uint_as_float((unsigned)g0) + __uint_as_float((unsigned)g1);
        ++spins;
        if (!done) { __builtin_amdgcn_s_sleep(2); if (expired()) { timed_out = true; break; } }
      }
      if (timed_out) break;
      // (XCD-order sum over the 8 partials: three DPP / shuffle steps over lanes with equal rr)
      sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      // hop 2: publish the 16 reduced values, then sweep all 4096 reduced granules (64 per lane: 8 passes of 8-byte loads, 8 in flight)
      if (lane < 16) st_granule(a.red + slot * 128 + 16 * xcc + lane, __float_as_uint(sum), (unsigned)r);
      done = false;
      while (!done) {
        bool all = true;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
          unsigned long long g[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = ld_granule(a.red + (size_t)(p8 * 8 + k) * 64 + lane);
#pragma unroll
          for (int k = 0; k < 8; ++k) all = all && (unsigned)(g[k] >> 32) >= (unsigned)r;
        }
        done = __all(all);
        ++spins;
        if (!done) { __builtin_amdgcn_s_sleep(2); if (expired()) { timed_out = true; break; } }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (a.mode == 0) {  // stream only: let the other waves run for a fixed time
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < 100 * 60) __builtin_amdgcn_s_sleep(8);  // 60 us
  }
  if (lane == 0) *(volatile int *)&stop = 1;
  __syncthreads();
  if (lane == 0) {
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    a.times[blockIdx.x * 4 + 0] = t1 - t0;
    a.times[blockIdx.x * 4 + 1] = streamed[1] + streamed[2] + streamed[3];
    a.times[blockIdx.x * 4 + 2] = timed_out ? ~0ull : spins;
    a.times[blockIdx.x * 4 + 3] = t2 - t0;
  }
}

int main() {
  const size_t wbytes = (size_t)2 << 30;
  unsigned char *w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0x5a, wbytes));
  unsigned long long *slabL, *part, *red, *times; int *xc;
  CK(hipMalloc(&slabL, 8 * 32 * 8)); CK(hipMalloc(&part, 8 * 4096 * 8)); CK(hipMalloc(&red, 4096 * 8)); CK(hipMalloc(&times, 256 * 4 * 8)); CK(hipMalloc(&xc, 8 * 4));
  unsigned long long ht[256 * 4]; int hx[8];
  printf("# persistent launch, 256 workgroups x (1 sync wave + 3 streaming waves, 4 x 1 KiB nt loads in flight each); times from s_memrealtime (100 MHz)\n");
  for (int streaming = 1; streaming >= 0; --streaming)
    for (int mode = 0; mode <= 2; ++mode) {
      if (!streaming && mode == 0) continue;
      const int rounds = mode == 0 ? 0 : 200;
      CK(hipMemset(slabL, 0, 8 * 32 * 8)); CK(hipMemset(part, 0, 8 * 4096 * 8)); CK(hipMemset(red, 0, 4096 * 8)); CK(hipMemset(xc, 0, 32)); CK(hipMemset(times, 0, 256 * 32));
      Args a{w, (unsigned)(wbytes - 4096), streaming ? (2u << 20) : 4096u, slabL, part, red, times, xc, mode, rounds, 100 * 20000 /* 20 ms per round */};
      hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(ht, times, sizeof(ht), hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xc, sizeof(hx), hipMemcpyDeviceToHost));
      double tsum = 0, tmax = 0, bytes = 0, wall = 0; int bad = 0; double spins = 0;
      for (int b = 0; b < 256; ++b) {
        const double t = ht[b * 4] / 100.0; tsum += t; if (t > tmax) tmax = t;
        bytes += (double)ht[b * 4 + 1]; if (ht[b * 4 + 2] == ~0ull) ++bad; else spins += (double)ht[b * 4 + 2];
        if (ht[b * 4 + 3] / 100.0 > wall) wall = ht[b * 4 + 3] / 100.0;
      }
      printf("mode %d (%s) %s: census per XCD %d %d %d %d %d %d %d %d | ", mode, mode == 0 ? "stream only" : mode == 1 ? "edge L: XCD-local 32-granule hand-off" : "edge G: 8-partial reduce + 32 KiB all-gather",
             streaming ? "while streaming" : "parked", hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
      if (rounds) printf("%.2f us per round (mean over workgroups; slowest %.2f), %.1f polls per round, %d timed out | ", tsum / 256 / rounds, tmax / rounds, spins / 256 / rounds, bad);
      printf("stream %.2f TB/s over %.1f us\n", bytes / wall / 1e6, wall);
      fflush(stdout);
    }
  return 0;
}
