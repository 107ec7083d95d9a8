// ext_isq.hip -- in-situ quantization (ISQ) of dense weights to GGML Q8_0 blocks on the GPU.
// Reference: `generate_isq!` (mistralrs-quant/src/utils/isq.rs:323-361) calls candle's QTensor::quantize(w, Q8_0) on the CPU and
// uploads the blocks; here the same per-block rule (GGML quantize_row_q8_0: d = amax/127, id = d ? 1/d : 0, q = round(x*id),
// d stored as f16) runs on the device where the dense weight already lives: 288 GB of HBM hold the bf16 source and the Q8_0
// result side by side, so there is no host round trip.  Output = standard Q8_0 bytes [N][K/32][34] consumed by the same GEMV /
// GEMM kernels as GGUF Q8_0 files.  `K % 32 != 0` is refused (the reference falls back to another dtype: isq.rs:249-287).
#include "common.cuh"

namespace mrs {

template <class T>
__global__ void __launch_bounds__(256) isq_q8_0_kernel(const T *__restrict__ w, uint8_t *__restrict__ out, size_t nblocks) {
  // one 32-lane half-wave per block: lane l holds value l of the block
  const size_t blk = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (blk >= nblocks) return;  // whole half-waves leave together
  const int l = threadIdx.x & 31;
  const float x = to_f<T>(w[blk * 32 + l]);
  float amax = fabsf(x);
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
  const float d = amax / 127.0f;
  const float id = d != 0.0f ? 1.0f / d : 0.0f;
  uint8_t *b = out + blk * 34;
  ((int8_t *)(b + 2))[l] = (int8_t)roundf(x * id);
  if (l == 0) *(uint16_t *)b = float_to_half_bits(d);
}

}  // namespace mrs

// src: dense [N*K] elements, dtype 0 = f32, 1 = f16, 30 = bf16 (ggml ids); dst: N*K/32*34 bytes.  Returns 0 / -1.
extern "C" int mrs_isq_quantize_q8_0(const void *src, int src_dtype, void *dst, long long n_elements, void *stream) {
  if (n_elements <= 0) return 0;
  if (n_elements % 32) return -1;
  const size_t nb = (size_t)n_elements / 32;
  const dim3 grid((unsigned)((nb + 7) / 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
  case 0: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<float>, grid, block, 0, s, (const float *)src, (uint8_t *)dst, nb); return 0;
  case 1: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<mrs::f16_t>, grid, block, 0, s, (const mrs::f16_t *)src, (uint8_t *)dst, nb); return 0;
  case 30: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<mrs::bf16_t>, grid, block, 0, s, (const mrs::bf16_t *)src, (uint8_t *)dst, nb); return 0;
  default: return -1;
  }
}
