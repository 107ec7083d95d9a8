// stream_probe.hip -- round 5 measurement (not product code): what a weight-streaming launch can reach on the MI355X as a function of waves per workgroup,
// loads in flight per wave, load width mix and cache policy; the cost of out-of-range ("dead") buffer loads.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
// Every configuration streams `bytes per launch` from a fresh slice of a 3 GB buffer (nothing Infinity-Cache resident), 256 workgroups, back-to-back launches on
// one stream, HIP events around the sequence: us per launch INCLUDES the launch boundary, like scripts/bench_dec.py.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// wave w of workgroup b owns `per_wave` bytes; D loads of 1 KiB (64 lanes x 16 B) in flight; SMALL: every second pair of loads is two 4-byte loads (the tile
// mix of dec_core2.cuh: 2 x dwordx4 + 2 x dword per 2.3 KB); AUX: 0 default policy, 2 nt
template <int D, int AUX, bool SMALL>
__global__ void probe(const unsigned char *base, unsigned total, unsigned per_wave, unsigned *out, int dead_tail) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)total, 0x00020000);
  unsigned off = (blockIdx.x * nw + wave) * per_wave;
  const unsigned end = off + per_wave;
  v4u r[D];
  unsigned s[D];
  unsigned acc = 0;
  auto issue = [&](int i) {
    const unsigned o = off < end ? off : 0xF0000000u;
    r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, o + lane * 16, 0, AUX);
    if constexpr (SMALL) s[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, o + 1024 + lane * 4, 0, AUX); else s[i] = 0;
    off += SMALL ? 1280 : 1024;
  };
#pragma unroll
  for (int i = 0; i < D; ++i) issue(i);
  const unsigned stop = end + (unsigned)dead_tail * D * (SMALL ? 1280u : 1024u);  // dead_tail: extra passes of out-of-range requests after the data
  do {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w ^ s[i];
      issue(i);
    }
  } while (off < stop + D * (SMALL ? 1280u : 1024u));
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int D, int AUX, bool SMALL> static float run(const unsigned char *buf, size_t buf_bytes, size_t launch_bytes, int waves, unsigned *out, int dead_tail, int reps) {
  const unsigned per_wave = (unsigned)(launch_bytes / 256 / waves) / (SMALL ? 1280 : 1024) * (SMALL ? 1280 : 1024);
  const size_t slice = (size_t)per_wave * waves * 256;
  const int nslices = (int)(buf_bytes / slice);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    const int n = nslices < 48 ? nslices : 48;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((probe<D, AUX, SMALL>), dim3(256), dim3(64 * waves), 0, 0, buf + (size_t)i * slice, (unsigned)slice, per_wave, out, dead_tail);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / n;
    if (us < best) best = us;
  }
  return best;
}

int main() {
  const size_t buf_bytes = (size_t)3 << 30;
  unsigned char *buf; unsigned *out;
  CK(hipMalloc(&buf, buf_bytes)); CK(hipMalloc(&out, 4096 * 4));
  CK(hipMemset(buf, 0x5a, buf_bytes));
  CK(hipDeviceSynchronize());
  const size_t sizes[] = {(size_t)9 << 20, (size_t)14 << 20, (size_t)40 << 20, (size_t)66 << 20, (size_t)430 << 20};
  printf("# us per launch (boundary included), 256 workgroups; TB/s = bytes / us\n");
  printf("%8s %6s %4s %4s %6s %9s %7s\n", "MB", "waves", "D", "nt", "small", "us", "TB/s");
#define ROW(D, AUX, SMALL, W, SZ) { const float us = run<D, AUX, SMALL>(buf, buf_bytes, SZ, W, out, 0, 3); \
    printf("%8.1f %6d %4d %4d %6d %9.2f %7.2f\n", SZ / 1048576.0, W, D, AUX == 2, (int)SMALL, us, SZ / us / 1e6); fflush(stdout); }
  for (size_t sz : sizes) {
    for (int w : {4, 8, 16}) {
      ROW(2, 2, false, w, sz) ROW(4, 2, false, w, sz) ROW(8, 2, false, w, sz) ROW(16, 2, false, w, sz)
    }
    ROW(8, 0, false, 8, sz) ROW(4, 0, false, 16, sz)
    ROW(4, 2, true, 8, sz) ROW(8, 2, true, 8, sz) ROW(4, 2, true, 16, sz)
  }
  // dead requests: the same 14 MB / 66 MB launches with 1 and 2 extra passes of D out-of-range requests per wave
  printf("# dead tail: extra passes of D out-of-range requests per wave after the data\n");
  for (size_t sz : {(size_t)14 << 20, (size_t)66 << 20})
    for (int dt : {0, 1, 2, 4}) {
      const float a = run<8, 2, false>(buf, buf_bytes, sz, 8, out, dt, 3), b = run<8, 2, true>(buf, buf_bytes, sz, 8, out, dt, 3);
      printf("MB %.0f dead passes %d: D8 x4-only %.2f us, D8 tile-mix %.2f us\n", sz / 1048576.0, dt, a, b); fflush(stdout);
    }
  // an empty launch sequence: the boundary itself
  { const float us = run<2, 2, false>(buf, buf_bytes, (size_t)256 * 8 * 1024 * 2, 8, out, 0, 3); printf("# 4 MB launch (2 KB per wave): %.2f us\n", us); }
  return 0;
}
