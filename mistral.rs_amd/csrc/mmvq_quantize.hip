// mmvq_quantize.hip -- activation -> Q8_1 quantizer behind
//   launch_mmvq_gguf_quantize_q8_1_{bf16,f16,f32}(x, vy, kx, kx_padded, num_rows, stream)
// Reference: mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:1220-1318 (kernels), :1606-1641 (launchers).
// Semantics kept bit-for-bit: rows zero-padded to kx_padded, per 32 values d = amax/127,
// q = (amax == 0) ? 0 : (int8)roundf(x/d), ds = (half(d), half(sum x)) with the 32-lane xor-butterfly
// summation order.  (IEEE division here; the reference builds with --use_fast_math.)
#include "common.cuh"

namespace mrs {

template <class T>
__global__ void __launch_bounds__(256) quantize_q8_1_kernel(const T *__restrict__ x, uint8_t *__restrict__ y, int kx, int kx_padded) {
  const int ix = blockDim.x * blockIdx.x + threadIdx.x;
  if (ix >= kx_padded) return;  // kx_padded % 32 == 0: whole 32-lane groups leave together
  const int iy = blockIdx.y;
  const float xi = ix < kx ? to_f<T>(x[(size_t)iy * kx + ix]) : 0.0f;
  float amax = fabsf(xi), sum = xi;
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) {  // stays inside each 32-lane half of the wave64
    amax = fmaxf(amax, __shfl_xor(amax, m, 64));
    sum += __shfl_xor(sum, m, 64);
  }
  const float d = amax / 127.0f;
  const int8_t q = amax == 0.0f ? (int8_t)0 : (int8_t)roundf(xi / d);
  uint8_t *blk = y + ((size_t)iy * (kx_padded / 32) + ix / 32) * 36;
  ((int8_t *)(blk + 4))[ix & 31] = q;
  if ((ix & 31) == 0) {
    ((uint16_t *)blk)[0] = float_to_half_bits(d);
    ((uint16_t *)blk)[1] = float_to_half_bits(sum);
  }
}

template <class T> static void launch_quantize(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream, int num_blocks_x = -1) {
  if (num_rows <= 0 || kx_padded <= 0 || num_blocks_x == 0) return;
  dim3 grid(num_blocks_x > 0 ? num_blocks_x : (kx_padded + 255) / 256, num_rows, 1);
  hipLaunchKernelGGL((quantize_q8_1_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T *)x, (uint8_t *)vy, kx, kx_padded);
}

}  // namespace mrs

extern "C" void launch_mmvq_gguf_quantize_q8_1_bf16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream) {
  mrs::launch_quantize<mrs::bf16_t>(x, vy, kx, kx_padded, num_rows, stream);
}
extern "C" void launch_mmvq_gguf_quantize_q8_1_f16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream) {
  mrs::launch_quantize<mrs::f16_t>(x, vy, kx, kx_padded, num_rows, stream);
}
extern "C" void launch_mmvq_gguf_quantize_q8_1_f32(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream) {
  mrs::launch_quantize<float>(x, vy, kx, kx_padded, num_rows, stream);
}

// The MoE paths' own copies of the same quantizer (kernels/indexed_moe/indexed_moe.cu:673-808,1016-1023; Rust: src/gguf/ffi.rs:16-60;
// callers gguf/cuda.rs:514-588,1340-1640): identical arithmetic (same 32-lane butterfly), f32 variant with a caller-supplied grid width
// (`num_blocks_x` blocks of 256 columns; the callers pass ceil(kx_padded / 256)).
extern "C" void launch_quantize_q8_1(const float *x, void *vy, int kx, int kx_padded, int num_blocks_x, int num_rows, void *stream) {
  mrs::launch_quantize<float>(x, vy, kx, kx_padded, num_rows, stream, num_blocks_x > 0 ? num_blocks_x : 0);
}
extern "C" void launch_quantize_q8_1_bf16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream) {
  mrs::launch_quantize<mrs::bf16_t>(x, vy, kx, kx_padded, num_rows, stream);
}
extern "C" void launch_quantize_q8_1_f16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream) {
  mrs::launch_quantize<mrs::f16_t>(x, vy, kx, kx_padded, num_rows, stream);
}
