// ext_gemm.hip -- prefill GEMM: GGUF-quantized weights x f32 activations on the bf16 matrix cores (MI355X / gfx950).
//
//   out[m][n] (+)= sum_k bf16(x[m][k]) * bf16(dequant(W)[n][k])        m < M tokens, n < N out-features, f32 accumulate
//
// Role in the reference: the prefill branch of GgufMatMul::forward_raw (mistralrs-quant/src/gguf/mod.rs:298-323, b > 8)
// -> fast_mmq::{plain,fused_qkv,fused_glu,fused_ffn} (gguf/fast_mmq.rs:528-635,762-821), i.e. the llama.cpp MMQ port under
// kernels/mmq_gguf/.  The reference quantizes the activations to int8 (block_q8_1_mmq) and uses integer MMA; north_star asks for
// the MI355X formulation instead: block-dequant fused into the GEMM -- weight tiles are decoded to bf16 straight into LDS,
// activations are rounded to bf16 while they are staged, v_mfma_f32_32x32x16_bf16 accumulates in f32 -- so the boundary sits
// one level up, at the fast_mmq::* signatures (raw x pointer), see SURVEY 7 hard part 7.  Parity is judged against oracle A
// (exact dequant matmul) within the bf16 input-rounding bound (tests/test_gemm.py).
//
// Tiling: workgroup = 256 threads = 4 waves, tile 128 (tokens) x 128 (weight rows) x 64 (k); a wave owns 64 x 64 = 2 x 2 MFMA
// tiles of 32 x 32 (64 accumulator VGPRs).  Per k-step every thread (a) converts 32 activations f32 -> bf16 and (b) decodes 32
// weights (one 32-weight sub-block of one row: header + 16/32 B of quants) into LDS; both tiles are [row][64 k] with the 16-byte
// chunk index XOR-swizzled by (row & 7), so the ds_read_b128 fragment reads (8 consecutive k per lane) are conflict-free.
// LDS is double buffered: global loads of step i+1 are issued before the MFMAs of step i, decoded/converted after them.
#include "gguf_blocks.cuh"
#include <stdio.h>

namespace mrs {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 64, GT = 256;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return (unsigned)float_to_bf16_bits(a) | ((unsigned)float_to_bf16_bits(b) << 16);
}

// raw registers of one 32-weight sub-block (row r, k in [kb*64 + half*32, +32))
template <int TYPE> struct RawG;
template <> struct RawG<T_Q4_K> { int4 hdr, q0, q1; };
template <> struct RawG<T_Q5_K> { int4 hdr, q0, q1, h0, h1; };
template <> struct RawG<T_Q6_K> { int4 l0, l1, h0, h1; unsigned sc, d; };
template <> struct RawG<T_Q8_0> { int4 q0, q1; unsigned d; };

template <int TYPE> __device__ __forceinline__ RawG<TYPE> gemm_load_w(const uint8_t *__restrict__ row, int kb, int half) {
  RawG<TYPE> r;
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *blk = row + (size_t)(kb >> 2) * 144;
    r.hdr = ld16_a4(blk);
    r.q0 = ld16_a4(blk + 16 + (kb & 3) * 32);
    r.q1 = ld16_a4(blk + 32 + (kb & 3) * 32);
  } else if constexpr (TYPE == T_Q5_K) {
    const uint8_t *blk = row + (size_t)(kb >> 2) * 176;
    r.hdr = ld16_a4(blk);
    r.h0 = ld16_a4(blk + 16);
    r.h1 = ld16_a4(blk + 32);
    r.q0 = ld16_a4(blk + 48 + (kb & 3) * 32);
    r.q1 = ld16_a4(blk + 64 + (kb & 3) * 32);
  } else if constexpr (TYPE == T_Q6_K) {
    // k-range quarter c = kb & 3: half-block h = c >> 1, 32-weight groups qt = (c & 1) * 2 + half
    const uint8_t *blk = row + (size_t)(kb >> 2) * 210;
    const int c = kb & 3, h = c >> 1, qt = (c & 1) * 2 + half;
    r.l0 = ld16_a2(blk + h * 64 + (qt & 1) * 32);
    r.l1 = ld16_a2(blk + h * 64 + (qt & 1) * 32 + 16);
    r.h0 = ld16_a2(blk + 128 + h * 32);
    r.h1 = ld16_a2(blk + 144 + h * 32);
    r.sc = ld2(blk + 192 + h * 8 + qt * 2);
    r.d = ld2(blk + 208);
  } else {
    const uint8_t *blk = row + (size_t)(kb * 2 + half) * 34;
    r.d = ld2(blk);
    r.q0 = ld16_a2(blk + 2);
    r.q1 = ld16_a2(blk + 18);
  }
  return r;
}

// 32 weights -> 32 bf16 (16 dwords), exact GGUF decode w = scale*q - offset in f32, then RNE to bf16
template <int TYPE> __device__ __forceinline__ void gemm_decode_w(const RawG<TYPE> &w, int kb, int half, unsigned (&o)[16]) {
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const int c = kb & 3;
    const float d = half_bits_to_float((uint16_t)(w.hdr.x & 0xffff)), dmin = half_bits_to_float((uint16_t)((unsigned)w.hdr.x >> 16));
    const int sh = 16 * (c & 1);
    const unsigned A = (unsigned)w.hdr.y >> sh, B = (unsigned)w.hdr.z >> sh, C = (unsigned)w.hdr.w >> sh;
    const unsigned scH = (C & 0x0f0fu) | ((A >> 2) & 0x3030u), mH = ((C >> 4) & 0x0f0fu) | ((B >> 2) & 0x3030u);
    const unsigned sc2 = (c < 2) ? (A & 0x3f3fu) : scH, mm2 = (c < 2) ? (B & 0x3f3fu) : mH;
    const float s = d * (float)((sc2 >> (8 * half)) & 0xff), m = dmin * (float)((mm2 >> (8 * half)) & 0xff);
    const unsigned q[8] = {(unsigned)w.q0.x, (unsigned)w.q0.y, (unsigned)w.q0.z, (unsigned)w.q0.w,
                           (unsigned)w.q1.x, (unsigned)w.q1.y, (unsigned)w.q1.z, (unsigned)w.q1.w};
    unsigned hb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (TYPE == T_Q5_K) {
      const unsigned hh[8] = {(unsigned)w.h0.x, (unsigned)w.h0.y, (unsigned)w.h0.z, (unsigned)w.h0.w,
                              (unsigned)w.h1.x, (unsigned)w.h1.y, (unsigned)w.h1.z, (unsigned)w.h1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) hb[i] = ((hh[i] >> (2 * c + half)) & 0x01010101u) << 4;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned v = ((q[i] >> (4 * half)) & 0x0f0f0f0fu) | hb[i];
      o[2 * i] = pack_bf16(s * (float)(v & 0xff) - m, s * (float)((v >> 8) & 0xff) - m);
      o[2 * i + 1] = pack_bf16(s * (float)((v >> 16) & 0xff) - m, s * (float)(v >> 24) - m);
    }
  } else if constexpr (TYPE == T_Q6_K) {
    const int c = kb & 3, qt = (c & 1) * 2 + half;
    const float d = half_bits_to_float((uint16_t)w.d);
    const float s0 = d * (float)(int)(int8_t)(w.sc & 0xff), s1 = d * (float)(int)(int8_t)((w.sc >> 8) & 0xff);
    const unsigned ql[8] = {(unsigned)w.l0.x, (unsigned)w.l0.y, (unsigned)w.l0.z, (unsigned)w.l0.w,
                            (unsigned)w.l1.x, (unsigned)w.l1.y, (unsigned)w.l1.z, (unsigned)w.l1.w};
    const unsigned qh[8] = {(unsigned)w.h0.x, (unsigned)w.h0.y, (unsigned)w.h0.z, (unsigned)w.h0.w,
                            (unsigned)w.h1.x, (unsigned)w.h1.y, (unsigned)w.h1.z, (unsigned)w.h1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned lo = (qt < 2 ? ql[i] : (ql[i] >> 4)) & 0x0f0f0f0fu;
      const unsigned v = lo | (((qh[i] >> (2 * qt)) & 0x03030303u) << 4);
      const float s = i < 4 ? s0 : s1;  // 16 weights per scale
      o[2 * i] = pack_bf16(s * (float)((int)(v & 0xff) - 32), s * (float)((int)((v >> 8) & 0xff) - 32));
      o[2 * i + 1] = pack_bf16(s * (float)((int)((v >> 16) & 0xff) - 32), s * (float)((int)(v >> 24) - 32));
    }
  } else {
    const float d = half_bits_to_float((uint16_t)w.d);
    const int q[8] = {w.q0.x, w.q0.y, w.q0.z, w.q0.w, w.q1.x, w.q1.y, w.q1.z, w.q1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[2 * i] = pack_bf16(d * (float)(int)(int8_t)(q[i] & 0xff), d * (float)(int)(int8_t)((q[i] >> 8) & 0xff));
      o[2 * i + 1] = pack_bf16(d * (float)(int)(int8_t)((q[i] >> 16) & 0xff), d * (float)(int)(int8_t)((unsigned)q[i] >> 24));
    }
  }
}

struct GemmArgs {
  const uint8_t *w;
  const float *x;
  float *out;
  int M, N, K, ldx, ldo, accumulate;
  size_t row_bytes;
};

// LDS tile [128 rows][64 k] bf16 = 8 chunks of 16 B per row, chunk index XOR (row & 7)
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int TYPE>
__global__ void __launch_bounds__(GT) gemm_q_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16 KB | B 16 KB]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;  // wave's 64 x 64 sub-tile
  // staging roles: A: thread -> row ar = tid / 2, 32 consecutive k (half ah); B: row br = tid / 2, sub-block half bh
  const int ar = tid >> 1, ah = tid & 1;
  const bool a_live = m0 + ar < a.M;
  const float *xrow = a.x + (size_t)min(m0 + ar, a.M - 1) * a.ldx + ah * 32;
  const uint8_t *wrow = a.w + (size_t)min(n0 + ar, a.N - 1) * a.row_bytes;
  const int nk = a.K / GK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 xa[8];
  RawG<TYPE> wb;
  auto issue = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = *(const float4 *)(xrow + (size_t)kb * GK + i * 4);
    wb = gemm_load_w<TYPE>(wrow, kb, ah);
  };
  auto commit = [&](int kb, char *buf) {
    char *A = buf, *B = buf + GM * GK * 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int4 v;
      if (a_live) {
        v.x = (int)pack_bf16(xa[2 * c].x, xa[2 * c].y); v.y = (int)pack_bf16(xa[2 * c].z, xa[2 * c].w);
        v.z = (int)pack_bf16(xa[2 * c + 1].x, xa[2 * c + 1].y); v.w = (int)pack_bf16(xa[2 * c + 1].z, xa[2 * c + 1].w);
      } else {
        v = make_int4(0, 0, 0, 0);
      }
      *(int4 *)(A + tile_off(ar, ah * 4 + c)) = v;
    }
    unsigned o[16];
    gemm_decode_w<TYPE>(wb, kb, ah, o);
#pragma unroll
    for (int c = 0; c < 4; ++c) *(int4 *)(B + tile_off(ar, ah * 4 + c)) = make_int4((int)o[4 * c], (int)o[4 * c + 1], (int)o[4 * c + 2], (int)o[4 * c + 3]);
  };

  issue(0);
  commit(0, smem);
  __syncthreads();
  const int frow = lane & 31, fk = lane >> 5;  // fragment: row / column index, which 8-k half of a 16-k slab
  for (int kb = 0; kb < nk; ++kb) {
    char *cur = smem + (kb & 1) * (2 * GM * GK * 2);
    char *nxt = smem + ((kb + 1) & 1) * (2 * GM * GK * 2);
    if (kb + 1 < nk) issue(kb + 1);
    const char *A = cur, *B = cur + GM * GK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // four 16-k slabs
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8 *)(A + tile_off(wm + i * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8 *)(B + tile_off(wn + j * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nk) commit(kb + 1, nxt);
    __syncthreads();
  }
  // C layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < a.N) {
          float *p = a.out + (size_t)m * a.ldo + n;
          *p = a.accumulate ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

template <int TYPE> static int gemm_launch(const GemmArgs &a, hipStream_t s) {
  auto kern = gemm_q_kernel<TYPE>;
  constexpr size_t lds = 2 * 2 * GM * GK * 2;  // 64 KiB
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(kern, dim3((a.N + GN - 1) / GN, (a.M + GM - 1) / GM), dim3(GT), lds, s, a);
  return 0;
}

}  // namespace mrs

using namespace mrs;

// out[m*ldo + n] (+)= sum_k bf16(x[m*ldx + k]) * bf16(W[n][k]);  W: raw GGUF blocks [N][K/blk] of type q4_k / q5_k / q6_k / q8_0.
// Returns 0, or -1 for an unsupported type / shape (K % 256 for the K-quants, K % 64 for Q8_0).
extern "C" int mrs_gemm_q_f32(const void *w, int ggml_type, int N, int K, const float *x, int ldx, float *out, int ldo, int M, int accumulate,
                              void *stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || K % 64 || ((ggml_type == T_Q4_K || ggml_type == T_Q5_K || ggml_type == T_Q6_K) && K % 256) || (ldx & 3)) return -1;
  GemmArgs a{(const uint8_t *)w, x, out, M, N, K, ldx, ldo, accumulate, 0};
  switch (ggml_type) {
  case T_Q4_K: a.row_bytes = (size_t)(K / 256) * 144; return gemm_launch<T_Q4_K>(a, (hipStream_t)stream);
  case T_Q5_K: a.row_bytes = (size_t)(K / 256) * 176; return gemm_launch<T_Q5_K>(a, (hipStream_t)stream);
  case T_Q6_K: a.row_bytes = (size_t)(K / 256) * 210; return gemm_launch<T_Q6_K>(a, (hipStream_t)stream);
  case T_Q8_0: a.row_bytes = (size_t)(K / 32) * 34; return gemm_launch<T_Q8_0>(a, (hipStream_t)stream);
  default: return -1;
  }
}
