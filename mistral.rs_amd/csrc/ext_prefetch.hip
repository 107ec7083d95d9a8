// ext_prefetch.hip -- two small MI355X utilities of the decode / prompt engines.
//
// (1) mrs_l3_prefetch: pull a byte range through the memory-side Infinity Cache (256 MiB) with default-policy loads that are thrown away.  A decode
//     step is a dependency chain of short launches whose fixed costs (entry, activation prologue, attention's latency chain, drain) leave HBM idle
//     for more than half of the step; the WEIGHTS of the following launches have no dependency at all, so a side stream of the captured step graph
//     can stream them into the Infinity Cache while the chain runs.  Nothing is written; results cannot change.
// (2) mrs_mfma_f16_int_probe: the prompt GEMM of ext_gemm_qi.hip relies on v_mfma_f32_32x32x16_f16 accumulating products of small integers EXACTLY
//     (all partial sums are integers below 2^24).  The probe runs a K-deep chain of such MFMAs on caller-supplied integer-valued f16 operands so that
//     a test can hold the matrix pipe to the integer result (tests/test_gemm_qi.py).
#include "common.cuh"

namespace mrs {

typedef unsigned pf_v4u __attribute__((ext_vector_type(4)));

// every workgroup walks the range with stride gridDim * 4 KiB * UNROLL; loads are b128 per lane, default cache policy (allocate in L2 and in the
// Infinity Cache), the values are OR-ed into a register that is stored only if it equals a value the data never produces
template <int UNROLL>
__global__ void __launch_bounds__(256) l3_prefetch_kernel(const uint8_t *__restrict__ p, size_t bytes, unsigned *sink) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, (short)0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFF : (int)bytes, 0x00020000);
  const size_t chunk = (size_t)256 * 16 * UNROLL;
  pf_v4u acc = {0u, 0u, 0u, 0u};
  for (size_t base = (size_t)blockIdx.x * chunk; base < bytes; base += (size_t)gridDim.x * chunk) {
    pf_v4u v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(base + (size_t)i * 4096 + threadIdx.x * 16), 0, 0);  // past the end: zeros, no traffic
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc |= v[i];
  }
  if ((acc.x & acc.y & acc.z & acc.w) == 0xFFFFFFFFu && sink) *sink = acc.x ^ acc.y;  // keeps the loads alive; all-ones in 16 bytes of every lane's OR never gates a result
}

typedef _Float16 pf_h8 __attribute__((ext_vector_type(8)));
typedef float pf_f16v __attribute__((ext_vector_type(16)));

// one wave: C[32][32] = sum over ksteps of A_k (32 x 16) * B_k (32 x 16)^T; a / b: [ksteps][64 lanes][8] f16 in operand order (lane l holds row l % 32,
// k = 8 * (l / 32) + j), out: [64 lanes][16] accumulator registers
__global__ void __launch_bounds__(64) mfma_f16_probe_kernel(const _Float16 *a, const _Float16 *b, float *out, int ksteps) {
  const int lane = threadIdx.x;
  pf_f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < ksteps; ++s) {
    const pf_h8 av = *(const pf_h8 *)(a + ((size_t)s * 64 + lane) * 8);
    const pf_h8 bv = *(const pf_h8 *)(b + ((size_t)s * 64 + lane) * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i];
}

}  // namespace mrs

extern "C" int mrs_l3_prefetch(const void *p, size_t bytes, int workgroups, void *sink, void *stream) {
  if (!p || bytes == 0) return 0;
  if (bytes > 0xFFFFFF00ull) return 1;  // one buffer descriptor per call
  if (workgroups <= 0) workgroups = 256;
  hipLaunchKernelGGL((mrs::l3_prefetch_kernel<8>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)p, bytes, (unsigned *)sink);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int mrs_mfma_f16_int_probe(const void *a, const void *b, float *out, int ksteps, void *stream) {
  hipLaunchKernelGGL(mrs::mfma_f16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const _Float16 *)a, (const _Float16 *)b, out, ksteps);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
