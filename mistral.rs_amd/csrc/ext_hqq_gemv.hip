// ext_hqq_gemv.hip -- fused HQQ dequant-GEMV for decode (b <= 8): out = x . W^T (+ bias) straight from the packed 4-bit / 8-bit HQQ tensor.
//
// Reference: HqqLayer::forward_raw = dequantize (kernels/hqq/hqq.cu:24-34,92-107: w = T(T(q - zero) * scale), group 64 along axis 0) then a dense matmul
// (mistralrs-quant/src/hqq/mod.rs:1092-1100,1163-1171); no fused kernel exists there.  Here the dequantized tensor is never materialised: per token the kernel
// reads the packed bytes once (N K / 2 or N K bytes instead of writing and re-reading 2-4 N K bytes of dense weights).
//
// Layout facts used (hqq/quantize.rs:7-70): the [N, K] weight is viewed as [64, w] with w = N K / 64; element (n, k) sits in group row
// n / (N / 64) and column j = (n % (N / 64)) * K + k, so the 64 output rows n_lo + r * (N / 64) share one K-vector of (scale, zero); 4-bit packing puts group rows r
// (high nibble) and r + 32 (low nibble) in byte [r][j].  One workgroup = one n_lo (and a slice of the packed rows): it streams 32 (16, 8) contiguous K-byte rows.
// Arithmetic = the reference's: the dequantized VALUE is bit-identical to dequantize_{4,8}bit (the two T-roundings are applied explicitly); products and
// sums in f32 like a dense f32-accumulate matmul.
#include "common.cuh"
#include <hip/hip_runtime.h>

namespace mrs_host { int fail(const char *fmt, ...); }

namespace mrs {
namespace hqqv {

constexpr int NT = 512, NW = 8, CH = 2048;  // threads, waves, columns per staged chunk of the (zero, scale) K-vectors

// round to T and back: the reference's dequantize kernels compute in T (hqq.cu:24-34: T(q) - zero, then * scale, each rounded to T)
template <class T> __device__ __forceinline__ float rt(float x);
template <> __device__ __forceinline__ float rt<float>(float x) { return x; }
template <> __device__ __forceinline__ float rt<f16_t>(float x) { return (float)(f16_t)x; }
typedef __bf16 hq_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hq_f32x2 __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ float rt<bf16_t>(float x) {  // one v_cvt_pk_bf16_f32 (RNE) + a shift
  const hq_f32x2 v = {x, 0.f};
  return __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(v, hq_bf16x2)) << 16);
}
// (q - z) is exact in f32 (q <= 255, z has <= 11 significant bits), rounding it to T is the reference's T-subtraction; the product of two T values
// is exact in f32, rounding it is the reference's T-multiplication: bit-identical to dequantize_{4,8}bit
template <class T> __device__ __forceinline__ float deq(unsigned q, float z, float s) { return rt<T>(rt<T>((float)q - z) * s); }

struct Args {
  const uint8_t *wq; const void *scale, *zero, *x, *bias; void *out;
  int N, K, ldx, ldo, rsplit;
};

// grid (N / 64, rsplit); LDS: xs [NCOLS][K] f32 | zs [2][CH] f32 (zero, scale of the current chunk)
// A wave owns `rpw` packed rows; per chunk it first issues every 16-byte load of its rows (bytes in flight), then decodes: 16 bytes = 16 columns x
// (high nibble = group row r, low nibble = group row r + 32), ~7 VALU per weight (extract, convert, subtract, 2 roundings, multiply, fma) against a
// 16-entry lookup table per column before (LDS-bound at 0.3-0.9 TB/s: 32 KiB of table writes per 8 KiB of packed weights).
template <int BITS, class T, int NCOLS>
__global__ void __launch_bounds__(NT) hqq_gemv_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PACKED_ROWS = BITS == 4 ? 32 : 64, MAXR = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, R = a.N / 64;          // R = output rows per group row
  const int n_lo = blockIdx.x;
  const size_t w = (size_t)R * K, j0 = (size_t)n_lo * K;
  const int pr_per_wg = PACKED_ROWS / a.rsplit, pr0 = blockIdx.y * pr_per_wg, rpw = pr_per_wg / NW;  // packed rows of this workgroup / per wave (1..8)
  float *xs = (float *)smem, *zs = xs + (size_t)NCOLS * K;
  const T *x = (const T *)a.x, *scale = (const T *)a.scale + j0, *zero = (const T *)a.zero + j0;
  for (int c = 0; c < NCOLS; ++c)
    for (int k = tid; k < K; k += NT) xs[(size_t)c * K + k] = to_f<T>(x[(size_t)c * a.ldx + k]);
  float acc_hi[MAXR][NCOLS], acc_lo[MAXR][NCOLS];
#pragma unroll
  for (int r = 0; r < MAXR; ++r)
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) { acc_hi[r][c] = 0.f; acc_lo[r][c] = 0.f; }
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  for (int c0 = 0; c0 < K; c0 += CH) {
    __syncthreads();  // xs staged / the previous chunk's (zero, scale) no longer read
    for (int k = tid; k < CH && c0 + k < K; k += NT) { zs[k] = to_f<T>(zero[c0 + k]); zs[CH + k] = to_f<T>(scale[c0 + k]); }
    // the wave's packed bytes of this chunk: 16 B per lane and piece, two pieces of 1024 columns
    v4u raw[MAXR][2];
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = c0 + (h * 64 + lane) * 16;
        raw[r][h] = (r < rpw && kk < K) ? __builtin_nontemporal_load((const v4u *)(a.wq + (size_t)(pr0 + wave * rpw + r) * w + j0 + kk)) : v4u{0, 0, 0, 0};
      }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kl = (h * 64 + lane) * 16;  // first column of the piece inside the chunk
      if (c0 + kl < K) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // dword d = 4 columns
          const float4 z4 = *(const float4 *)(zs + kl + 4 * d), s4 = *(const float4 *)(zs + CH + kl + 4 * d);
          const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
          float xv[NCOLS][4];
#pragma unroll
          for (int c = 0; c < NCOLS; ++c) {
            const float4 t = *(const float4 *)(xs + (size_t)c * K + c0 + kl + 4 * d);
            xv[c][0] = t.x; xv[c][1] = t.y; xv[c][2] = t.z; xv[c][3] = t.w;
          }
#pragma unroll
          for (int r = 0; r < MAXR; ++r) {
            if (r < rpw) {  // wave-uniform
              const unsigned v = raw[r][h][d];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned bq = (v >> (8 * e)) & 0xff;
                if constexpr (BITS == 4) {
                  const float wh = deq<T>(bq >> 4, zz[e], ss[e]), wl = deq<T>(bq & 15, zz[e], ss[e]);
#pragma unroll
                  for (int c = 0; c < NCOLS; ++c) { acc_hi[r][c] = fmaf(wh, xv[c][e], acc_hi[r][c]); acc_lo[r][c] = fmaf(wl, xv[c][e], acc_lo[r][c]); }
                } else {
                  const float wh = deq<T>(bq, zz[e], ss[e]);
#pragma unroll
                  for (int c = 0; c < NCOLS; ++c) acc_hi[r][c] = fmaf(wh, xv[c][e], acc_hi[r][c]);
                }
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
    const size_t lds = (size_t)ncols * a.K * 4 + 2 * CH * 4;
    if (lds > 158 * 1024) return -2;
    lds_attr_once((const void *)kern, 158 * 1024);
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
// N % 64, K % 16, b > 8: the caller keeps dequantize + dense matmul); -2 = activation rows too long for LDS.
extern "C" int mrs_hqq_gemv(int bits, int dtype, const void *wq, const void *scale, const void *zero, const void *bias, const void *x, int ldx, void *out, int ldo,
                            int N, int K, int b, void *stream) {
  using namespace mrs::hqqv;
  if ((bits != 4 && bits != 8) || N <= 0 || K <= 0 || N % 64 || K % 16 || b < 1 || b > 8 || !wq || !scale || !zero || !x || !out) return -1;
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
