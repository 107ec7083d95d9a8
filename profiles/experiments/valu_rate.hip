// VALU issue-rate probe: N waves per SIMD each run a loop of independent / dependent ops.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE> __global__ void __launch_bounds__(256) k(int *out, int iters, int seed) {
  int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  const int m = 0x0f0f0f0f + seed;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent and/xor per iteration (bitwise)
      a0 = (a0 & m) ^ a1; a1 = (a1 & m) ^ a2; a2 = (a2 & m) ^ a3; a3 = (a3 & m) ^ a4; a4 = (a4 & m) ^ a5; a5 = (a5 & m) ^ a6; a6 = (a6 & m) ^ a7; a7 = (a7 & m) ^ a0;
    } else if (MODE == 1) {  // 8 independent dot4
      a0 = __builtin_amdgcn_sdot4(a1, m, a0, false); a1 = __builtin_amdgcn_sdot4(a2, m, a1, false); a2 = __builtin_amdgcn_sdot4(a3, m, a2, false); a3 = __builtin_amdgcn_sdot4(a4, m, a3, false);
      a4 = __builtin_amdgcn_sdot4(a5, m, a4, false); a5 = __builtin_amdgcn_sdot4(a6, m, a5, false); a6 = __builtin_amdgcn_sdot4(a7, m, a6, false); a7 = __builtin_amdgcn_sdot4(a0, m, a7, false);
    } else if (MODE == 2) {  // dependent dot4 chain (8 per iteration, one accumulator)
      a0 = __builtin_amdgcn_sdot4(a1, m, a0, false); a0 = __builtin_amdgcn_sdot4(a2, m, a0, false); a0 = __builtin_amdgcn_sdot4(a3, m, a0, false); a0 = __builtin_amdgcn_sdot4(a4, m, a0, false);
      a0 = __builtin_amdgcn_sdot4(a5, m, a0, false); a0 = __builtin_amdgcn_sdot4(a6, m, a0, false); a0 = __builtin_amdgcn_sdot4(a7, m, a0, false); a0 = __builtin_amdgcn_sdot4(a1, m, a0, false);
    } else {  // 8 independent f32 fma
      float f0 = __int_as_float(a0), f1 = __int_as_float(a1), f2 = __int_as_float(a2), f3 = __int_as_float(a3);
      float f4 = __int_as_float(a4), f5 = __int_as_float(a5), f6 = __int_as_float(a6), f7 = __int_as_float(a7);
      const float c = __int_as_float(m);
      f0 = fmaf(f0, c, f1); f1 = fmaf(f1, c, f2); f2 = fmaf(f2, c, f3); f3 = fmaf(f3, c, f4); f4 = fmaf(f4, c, f5); f5 = fmaf(f5, c, f6); f6 = fmaf(f6, c, f7); f7 = fmaf(f7, c, f0);
      a0 = __float_as_int(f0); a1 = __float_as_int(f1); a2 = __float_as_int(f2); a3 = __float_as_int(f3); a4 = __float_as_int(f4); a5 = __float_as_int(f5); a6 = __float_as_int(f6); a7 = __float_as_int(f7);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
  int *out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    for (int mode = 0; mode < 4; ++mode) {
      const int grid = 256 * wg_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, rep);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, rep);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, rep);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: wg_per_cu waves (256 threads = 4 waves over 4 SIMDs), each iters*8 ops (mode 0: 16 ops)
      const double ops_per_simd = (double)wg_per_cu * iters * (mode == 0 ? 16 : 8);
      printf("waves/SIMD=%d mode=%d: %.3f ms -> %.2f ns/op/SIMD  (= %.2f cycles @2.4GHz)\n", wg_per_cu, mode, ms, ms * 1e6 / ops_per_simd, ms * 1e6 / ops_per_simd * 2.4);
    }
  }
  return 0;
}
