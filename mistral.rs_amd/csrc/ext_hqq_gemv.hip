// ext_hqq_gemv.hip -- fused HQQ dequant-GEMV for decode (b <= 8): out = x . W^T (+ bias) straight from the packed 4-bit / 8-bit HQQ tensor.
//
// Reference: HqqLayer::forward_raw = dequantize (kernels/hqq/hqq.cu:24-34,92-107: w = T(T(q - zero) * scale), group 64 along axis 0) then a dense matmul
// (mistralrs-quant/src/hqq/mod.rs:1092-1100,1163-1171); no fused kernel exists there.  Here the dequantized tensor is never materialised: per token the kernel
// reads the packed bytes once (N K / 2 or N K bytes instead of writing and re-reading 2-4 N K bytes of dense weights).
//
// Layout facts used (hqq/quantize.rs:7-70): the [N, K] weight is viewed as [64, w] with w = N K / 64; element (n, k) sits in group row
// n / (N / 64) and column j = (n % (N / 64)) * K + k, so the 64 output rows n_lo + r * (N / 64) share one K-vector of (scale, zero); 4-bit packing puts group rows r
// (high nibble) and r + 32 (low nibble) in byte [r][j].  One workgroup = one n_lo (and a slice of the packed rows): it streams 32 (16, 8) contiguous K-byte rows.
// Arithmetic = the reference's: the dequantized VALUE is bit-identical to dequantize_{4,8}bit (per column only 16 values exist in 4-bit: a 16 x 512 lookup table
// per K-chunk in LDS, laid out so that the 64 lanes of a read hit 64 banks); products and sums in f32 like a dense f32-accumulate matmul.
#include "common.cuh"
#include <hip/hip_runtime.h>

namespace mrs_host { int fail(const char *fmt, ...); }

namespace mrs {
namespace hqqv {

constexpr int NT = 512, NW = 8, CH = 512;  // threads, waves, columns per lookup-table chunk

template <class T> __device__ __forceinline__ float deq(unsigned q, float z, float s) { return to_f<T>(from_f<T>(round_to<T>((float)q - z) * s)); }

struct Args {
  const uint8_t *wq; const void *scale, *zero, *x, *bias; void *out;
  int N, K, ldx, ldo, rsplit;
};

// grid (N / 64, rsplit); LDS: xs [NCOLS][K] f32 | lut [16][CH] f32 (4-bit)
template <int BITS, class T, int NCOLS>
__global__ void __launch_bounds__(NT) hqq_gemv_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PACKED_ROWS = BITS == 4 ? 32 : 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, R = a.N / 64;          // R = output rows per group row
  const int n_lo = blockIdx.x;
  const size_t w = (size_t)R * K, j0 = (size_t)n_lo * K;
  const int pr_per_wg = PACKED_ROWS / a.rsplit, pr0 = blockIdx.y * pr_per_wg, rpw = pr_per_wg / NW;  // packed rows of this workgroup / per wave (1..8)
  float *xs = (float *)smem, *lut = xs + (size_t)NCOLS * K;
  const T *x = (const T *)a.x, *scale = (const T *)a.scale + j0, *zero = (const T *)a.zero + j0;
  for (int c = 0; c < NCOLS; ++c)
    for (int k = tid; k < K; k += NT) xs[(size_t)c * K + k] = to_f<T>(x[(size_t)c * a.ldx + k]);
  constexpr int MAXR = 8;
  float acc_hi[MAXR][NCOLS], acc_lo[MAXR][NCOLS];
#pragma unroll
  for (int r = 0; r < MAXR; ++r)
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) { acc_hi[r][c] = 0.f; acc_lo[r][c] = 0.f; }
  for (int c0 = 0; c0 < K; c0 += CH) {
    __syncthreads();  // xs staged / the previous chunk's table no longer read
    if constexpr (BITS == 4) {
      // column kk = c0 + tid of the chunk -> table position (tid % 4) * 128 + tid / 4: lane l of a reader (columns 4l' + e) then reads position e * 128 + l'
      if (c0 + tid < K) {
        const float z = to_f<T>(zero[c0 + tid]), s = to_f<T>(scale[c0 + tid]);
        const int pos = (tid & 3) * 128 + (tid >> 2);
#pragma unroll
        for (int q = 0; q < 16; ++q) lut[q * CH + pos] = deq<T>((unsigned)q, z, s);
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      if (r < rpw) {  // wave-uniform
        const uint8_t *row = a.wq + (size_t)(pr0 + wave * rpw + r) * w + j0 + c0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 512 columns = 2 dwords per lane
          const int d = half * 64 + lane, kk = 4 * d;  // dword index inside the chunk, its first column
          if (c0 + kk < K) {
            const unsigned v = *(const unsigned *)(row + kk);
            float wh[4], wl[4];
            if constexpr (BITS == 4) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned b = (v >> (8 * e)) & 0xff;
                wh[e] = lut[(b >> 4) * CH + e * 128 + d];
                wl[e] = lut[(b & 15) * CH + e * 128 + d];
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) wh[e] = deq<T>((v >> (8 * e)) & 0xff, to_f<T>(zero[c0 + kk + e]), to_f<T>(scale[c0 + kk + e]));
            }
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
              const float4 xv = *(const float4 *)(xs + (size_t)c * K + c0 + kk);
              acc_hi[r][c] = fmaf(wh[0], xv.x, acc_hi[r][c]); acc_hi[r][c] = fmaf(wh[1], xv.y, acc_hi[r][c]);
              acc_hi[r][c] = fmaf(wh[2], xv.z, acc_hi[r][c]); acc_hi[r][c] = fmaf(wh[3], xv.w, acc_hi[r][c]);
              if constexpr (BITS == 4) {
                acc_lo[r][c] = fmaf(wl[0], xv.x, acc_lo[r][c]); acc_lo[r][c] = fmaf(wl[1], xv.y, acc_lo[r][c]);
                acc_lo[r][c] = fmaf(wl[2], xv.z, acc_lo[r][c]); acc_lo[r][c] = fmaf(wl[3], xv.w, acc_lo[r][c]);
              }
            }
          }
        }
      }
    }
  }
  T *out = (T *)a.out;
  const T *bias = (const T *)a.bias;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    if (r < rpw) {
      const int gr = pr0 + wave * rpw + r;  // group row of the high nibble / of the byte
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const float sh = wave_sum(acc_hi[r][c]);
        const int n = gr * R + n_lo;
        if (lane == 0) out[(size_t)c * a.ldo + n] = from_f<T>(sh + (bias ? to_f<T>(bias[n]) : 0.f));
        if constexpr (BITS == 4) {
          const float sl = wave_sum(acc_lo[r][c]);
          const int n2 = (gr + 32) * R + n_lo;
          if (lane == 0) out[(size_t)c * a.ldo + n2] = from_f<T>(sl + (bias ? to_f<T>(bias[n2]) : 0.f));
        }
      }
    }
  }
}

template <int BITS, class T> static int launch(const Args &a, int b, hipStream_t s) {
  const dim3 grid(a.N / 64, a.rsplit), block(NT);
  auto go = [&](auto kern, int ncols) {
    const size_t lds = (size_t)ncols * a.K * 4 + (BITS == 4 ? 16 * CH * 4 : 0);
    if (lds > 158 * 1024) return -2;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
    return 0;
  };
  switch (b) {
  case 1: return go(hqq_gemv_kernel<BITS, T, 1>, 1); case 2: return go(hqq_gemv_kernel<BITS, T, 2>, 2);
  case 3: return go(hqq_gemv_kernel<BITS, T, 3>, 3); case 4: return go(hqq_gemv_kernel<BITS, T, 4>, 4);
  case 5: return go(hqq_gemv_kernel<BITS, T, 5>, 5); case 6: return go(hqq_gemv_kernel<BITS, T, 6>, 6);
  case 7: return go(hqq_gemv_kernel<BITS, T, 7>, 7); case 8: return go(hqq_gemv_kernel<BITS, T, 8>, 8);
  default: return -1;
  }
}

}  // namespace hqqv
}  // namespace mrs

// x [b][ldx], out [b][ldo], scale / zero [N K / 64], bias [N] or NULL, all of dtype 0 = f32, 1 = f16, 2 = bf16; wq = the packed tensor of HqqLayer
// ([32, N K / 64] bytes for 4 bit, [64, N K / 64] for 8 bit; group size 64, axis 0).  Returns 0; -1 = arguments outside the fused kernel (other bit widths,
// N % 64, K % 4, b > 8: the caller keeps dequantize + dense matmul); -2 = activation rows too long for LDS.
extern "C" int mrs_hqq_gemv(int bits, int dtype, const void *wq, const void *scale, const void *zero, const void *bias, const void *x, int ldx, void *out, int ldo,
                            int N, int K, int b, void *stream) {
  using namespace mrs::hqqv;
  if ((bits != 4 && bits != 8) || N <= 0 || K <= 0 || N % 64 || K % 4 || b < 1 || b > 8 || !wq || !scale || !zero || !x || !out) return -1;
  Args a{(const uint8_t *)wq, scale, zero, x, bias, out, N, K, ldx, ldo, 1};
  // split the packed rows of a group over more workgroups until the chip is covered (each split re-reads the K-vectors of x / scale / zero from L2)
  const int packed_rows = bits == 4 ? 32 : 64;
  while (a.rsplit * 2 * NW <= packed_rows && (N / 64) * a.rsplit < 256) a.rsplit *= 2;
  hipStream_t s = (hipStream_t)stream;
  if (bits == 4) {
    if (dtype == 0) return launch<4, float>(a, b, s);
    if (dtype == 1) return launch<4, mrs::f16_t>(a, b, s);
    if (dtype == 2) return launch<4, mrs::bf16_t>(a, b, s);
  } else {
    if (dtype == 0) return launch<8, float>(a, b, s);
    if (dtype == 1) return launch<8, mrs::f16_t>(a, b, s);
    if (dtype == 2) return launch<8, mrs::bf16_t>(a, b, s);
  }
  return -1;
}
