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
#include <stdlib.h>
#include <algorithm>

namespace mrs {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 64, GT = 256;

// two f32 -> packed bf16 (RNE): one v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddling RNE of common.cuh costs ~6 VALU per value)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
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
      o[2 * i] = pack_bf16(fmaf(s, (float)(v & 0xff), -m), fmaf(s, (float)((v >> 8) & 0xff), -m));
      o[2 * i + 1] = pack_bf16(fmaf(s, (float)((v >> 16) & 0xff), -m), fmaf(s, (float)(v >> 24), -m));
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
      const float m = 32.0f * s;  // w = s * (q - 32) = fma(s, q, -32 s): exact in f32 (q <= 63, s has <= 19 significant bits)
      o[2 * i] = pack_bf16(fmaf(s, (float)(v & 0xff), -m), fmaf(s, (float)((v >> 8) & 0xff), -m));
      o[2 * i + 1] = pack_bf16(fmaf(s, (float)((v >> 16) & 0xff), -m), fmaf(s, (float)(v >> 24), -m));
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

// up to 3 weight matrices of the same type that share the activations (q/k/v, gate/up): one launch, more workgroups in flight
struct GemmArgs {
  const uint8_t *w[3];
  float *out[3];
  int N[3], ldo[3], tile0[3];  // tile0[s]: first n-tile of segment s
  int nseg;
  const float *x;
  int M, K, ldx, accumulate;
  size_t row_bytes;
};

// LDS tile [rows][64 k] bf16 = 8 chunks of 16 B per row, chunk index XOR ((row >> 1) & 7): LDS has 64 banks (two 128-byte rows per
// bank sweep) and the 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...; MI355X_MICROARCH LDS table) hold, per row parity, 8 rows
// with distinct (row >> 1) & 7 -- conflict-free fragment reads.  (XOR by row & 7 is 2-way conflicted: measured +15 % kernel time.)
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// TM = token rows per workgroup tile: 128 (wave = 64 x 64) or 64 (wave = 32 x 64; twice the workgroups for shapes that
// would otherwise leave CUs idle, at twice the weight-decode work per FLOP)
template <int TYPE, int PF, int TM>
__global__ void __launch_bounds__(GT) gemm_q_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A TM x 64 bf16 | B 128 x 64 bf16]
  constexpr int MI = TM / 64;  // 32-row MFMA tiles per wave along m
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int seg = 0;
  if (a.nseg > 2 && (int)blockIdx.x >= a.tile0[2]) seg = 2;
  else if (a.nseg > 1 && (int)blockIdx.x >= a.tile0[1]) seg = 1;
  const int segN = a.N[seg], segldo = a.ldo[seg];
  const uint8_t *segw = a.w[seg];
  float *segout = a.out[seg];
  const int m0 = blockIdx.y * TM, n0 = ((int)blockIdx.x - a.tile0[seg]) * GN;
  const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * 64;  // wave's (TM/2) x 64 sub-tile
  // staging roles.  A (activations): 16 consecutive lanes cover one 256-byte row segment (64 f32), so a load instruction of a
  // wave reads 4 whole segments (fully used cache lines); thread -> column chunk xc = tid % 16 (4 floats), rows xr0 + 16 i.
  // B (weights): thread -> row ar = tid / 2, 32-weight sub-block ah = tid % 2 of the 64-k slab.
  const int ar = tid >> 1, ah = tid & 1;
  const int xc = tid & 15, xr0 = tid >> 4;
  const float *xbase = a.x + xc * 4;
  const uint8_t *wrow = segw + (size_t)min(n0 + ar, segN - 1) * a.row_bytes;
  const int nk = a.K / GK;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two staging register sets: the loads of step kb+2 are issued before the MFMAs of step kb, while step kb+1 (issued one
  // iteration earlier) is converted / decoded into the other LDS buffer after them -- a full iteration of latency cover.
  // Loads are unconditional (k index clamped): a load inside a conditional block makes hipcc drain the whole vmcnt queue.
  constexpr int XR = TM / 16;  // activation rows staged per thread
  float4 xa[PF][XR];
  RawG<TYPE> wb[PF];
  auto issue = [&](int set, int kb_raw) {
    const int kb = min(kb_raw, nk - 1);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int m = min(m0 + xr0 + 16 * i, a.M - 1);
      xa[set][i] = *(const float4 *)(xbase + (size_t)m * a.ldx + (size_t)kb * GK);
    }
    wb[set] = gemm_load_w<TYPE>(wrow, kb, ah);
  };
  auto commit = [&](int set, int kb_raw, char *buf) {
    const int kb = min(kb_raw, nk - 1);
    char *A = buf, *B = buf + TM * GK * 2;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int row = xr0 + 16 * i;
      const bool live = m0 + row < a.M;
      int2 v;
      v.x = live ? (int)pack_bf16(xa[set][i].x, xa[set][i].y) : 0;
      v.y = live ? (int)pack_bf16(xa[set][i].z, xa[set][i].w) : 0;
      *(int2 *)(A + tile_off(row, xc >> 1) + (xc & 1) * 8) = v;  // 4 bf16 = half of a 16-byte chunk
    }
    unsigned o[16];
    gemm_decode_w<TYPE>(wb[set], kb, ah, o);
#pragma unroll
    for (int c = 0; c < 4; ++c) *(int4 *)(B + tile_off(ar, ah * 4 + c)) = make_int4((int)o[4 * c], (int)o[4 * c + 1], (int)o[4 * c + 2], (int)o[4 * c + 3]);
  };
  const int frow = lane & 31, fk = lane >> 5;  // fragment: row / column index, which 8-k half of a 16-k slab
  auto mfma_tile = [&](const char *buf) {
    const char *A = buf, *B = buf + TM * GK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // four 16-k slabs
      bf16x8 af[MI], bfr[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8 *)(A + tile_off(wm + i * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8 *)(B + tile_off(wn + j * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };
  char *buf0 = smem, *buf1 = smem + (TM + GN) * GK * 2;
  if constexpr (PF == 2) {
    issue(0, 0);
    issue(1, 1);
    commit(0, 0, buf0);
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      issue(0, kb + 2);
      mfma_tile(buf0);
      commit(1, kb + 1, buf1);
      __syncthreads();
      if (kb + 1 >= nk) break;
      issue(1, kb + 3);
      mfma_tile(buf1);
      commit(0, kb + 2, buf0);
      __syncthreads();
    }
  } else {  // one staging set (fewer registers: two workgroups per CU overlap each other's MFMA and staging phases)
    issue(0, 0);
    commit(0, 0, buf0);
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      issue(0, kb + 1);
      mfma_tile(buf0);
      commit(0, kb + 1, buf1);
      __syncthreads();
      if (kb + 1 >= nk) break;
      issue(0, kb + 2);
      mfma_tile(buf1);
      commit(0, kb + 2, buf0);
      __syncthreads();
    }
  }
  // C layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < segN) {
          float *p = segout + (size_t)m * segldo + n;
          *p = a.accumulate ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

template <int TYPE, int TM> static void gemm_launch_tm(const GemmArgs &a, int tiles, hipStream_t s) {
  auto kern = gemm_q_kernel<TYPE, 2, TM>;
  constexpr size_t lds = 2 * (TM + GN) * GK * 2;  // 64 KiB (TM 128) / 48 KiB (TM 64)
  hipLaunchKernelGGL(kern, dim3(tiles, (a.M + TM - 1) / TM), dim3(GT), lds, s, a);
}
template <int TYPE> static int gemm_launch(GemmArgs a, hipStream_t s) {
  int tiles = 0;
  for (int i = 0; i < a.nseg; ++i) { a.tile0[i] = tiles; tiles += (a.N[i] + GN - 1) / GN; }
  // fewer than one workgroup per CU with 128-row tiles: halve the tile height (deterministic, unlike split-K atomics)
  if (a.M > 64 && tiles * ((a.M + 127) / 128) < 256) gemm_launch_tm<TYPE, 64>(a, tiles, s);
  else gemm_launch_tm<TYPE, 128>(a, tiles, s);
  return 0;
}


// =====================================================================================================================
// Large-M kernel (M > 128): bf16 activations, 256 x 128 x 64 tiles, 512 threads = 8 waves of 64 x 64, optional split-K.
//
// Why a second kernel: with 128-row tiles every weight is decoded M/128 times and each wave re-converts the activations; at one wave
// per SIMD the decode VALU (~125 instructions per k-step) does not fit into the MFMA issue gaps (~5 per MFMA, MI355X_MICROARCH
// "one wave per SIMD") and the phases serialise.  Here (a) the activations arrive as bf16 (converted ONCE by the producer), so the A
// tile is a plain 16-byte copy; (b) a thread decodes 16 weights per k-step instead of 32; (c) two waves share each SIMD and run the
// two halves of a k-step in OPPOSITE order -- waves 0-3: MFMA(buf) then decode -> other buf; waves 4-7: decode first, then MFMA --
// both orders are legal between two barriers (MFMA only reads the current buffer, decode only writes the other one), so one wave's
// decode VALU runs under its SIMD partner's MFMAs; (d) shapes with fewer workgroups than CUs split K across blockIdx.z into f32
// partials that a second kernel sums in a fixed order (deterministic, unlike atomics).
constexpr int HM = 256, HN = 128, HK = 64, HT = 512;

template <int TYPE> struct RawH;
template <> struct RawH<T_Q4_K> { int4 hdr, q; };
template <> struct RawH<T_Q5_K> { int4 hdr, q, h; };
template <> struct RawH<T_Q6_K> { int4 l, h; unsigned sc, d; };
template <> struct RawH<T_Q8_0> { int4 q; unsigned d; };

// 16 weights: row, k in [kb*64 + aq*16, +16)
template <int TYPE> __device__ __forceinline__ RawH<TYPE> gemm_load_w16(const uint8_t *__restrict__ row, int kb, int aq) {
  RawH<TYPE> r;
  const int c = kb & 3;
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *blk = row + (size_t)(kb >> 2) * 144;
    r.hdr = ld16_a4(blk);
    r.q = ld16_a4(blk + 16 + c * 32 + (aq & 1) * 16);
  } else if constexpr (TYPE == T_Q5_K) {
    const uint8_t *blk = row + (size_t)(kb >> 2) * 176;
    r.hdr = ld16_a4(blk);
    r.h = ld16_a4(blk + 16 + (aq & 1) * 16);
    r.q = ld16_a4(blk + 48 + c * 32 + (aq & 1) * 16);
  } else if constexpr (TYPE == T_Q6_K) {
    const uint8_t *blk = row + (size_t)(kb >> 2) * 210;
    const int h = c >> 1, qt = (c & 1) * 2 + (aq >> 1), l0 = (aq & 1) * 16, si = h * 8 + (c & 1) * 4 + aq;
    r.l = ld16_a2(blk + h * 64 + (qt & 1) * 32 + l0);
    r.h = ld16_a2(blk + 128 + h * 32 + l0);
    r.sc = (unsigned)ld2(blk + 192 + (si & ~1)) >> (8 * (si & 1));
    r.d = ld2(blk + 208);
  } else {
    const uint8_t *blk = row + (size_t)(kb * 2 + (aq >> 1)) * 34;
    r.d = ld2(blk);
    r.q = ld16_a2(blk + 2 + (aq & 1) * 16);
  }
  return r;
}

template <int TYPE> __device__ __forceinline__ void gemm_decode_w16(const RawH<TYPE> &w, int kb, int aq, unsigned (&o)[8]) {
  const int c = kb & 3, half = aq >> 1;
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const float d = half_bits_to_float((uint16_t)(w.hdr.x & 0xffff)), dmin = half_bits_to_float((uint16_t)((unsigned)w.hdr.x >> 16));
    const int sh = 16 * (c & 1);
    const unsigned A = (unsigned)w.hdr.y >> sh, B = (unsigned)w.hdr.z >> sh, C = (unsigned)w.hdr.w >> sh;
    const unsigned scH = (C & 0x0f0fu) | ((A >> 2) & 0x3030u), mH = ((C >> 4) & 0x0f0fu) | ((B >> 2) & 0x3030u);
    const unsigned sc2 = (c < 2) ? (A & 0x3f3fu) : scH, mm2 = (c < 2) ? (B & 0x3f3fu) : mH;
    const float s = d * (float)((sc2 >> (8 * half)) & 0xff), m = dmin * (float)((mm2 >> (8 * half)) & 0xff);
    const unsigned q[4] = {(unsigned)w.q.x, (unsigned)w.q.y, (unsigned)w.q.z, (unsigned)w.q.w};
    unsigned hb[4] = {0, 0, 0, 0};
    if constexpr (TYPE == T_Q5_K) {
      const unsigned hh[4] = {(unsigned)w.h.x, (unsigned)w.h.y, (unsigned)w.h.z, (unsigned)w.h.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) hb[i] = ((hh[i] >> (2 * c + half)) & 0x01010101u) << 4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned v = ((q[i] >> (4 * half)) & 0x0f0f0f0fu) | hb[i];
      o[2 * i] = pack_bf16(fmaf(s, (float)(v & 0xff), -m), fmaf(s, (float)((v >> 8) & 0xff), -m));
      o[2 * i + 1] = pack_bf16(fmaf(s, (float)((v >> 16) & 0xff), -m), fmaf(s, (float)(v >> 24), -m));
    }
  } else if constexpr (TYPE == T_Q6_K) {
    const int qt = (c & 1) * 2 + half;
    const float s = half_bits_to_float((uint16_t)w.d) * (float)(int)(int8_t)(w.sc & 0xff), m = 32.0f * s;
    const unsigned ql[4] = {(unsigned)w.l.x, (unsigned)w.l.y, (unsigned)w.l.z, (unsigned)w.l.w};
    const unsigned qh[4] = {(unsigned)w.h.x, (unsigned)w.h.y, (unsigned)w.h.z, (unsigned)w.h.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = (qt < 2 ? ql[i] : (ql[i] >> 4)) & 0x0f0f0f0fu;
      const unsigned v = lo | (((qh[i] >> (2 * qt)) & 0x03030303u) << 4);
      o[2 * i] = pack_bf16(fmaf(s, (float)(v & 0xff), -m), fmaf(s, (float)((v >> 8) & 0xff), -m));
      o[2 * i + 1] = pack_bf16(fmaf(s, (float)((v >> 16) & 0xff), -m), fmaf(s, (float)(v >> 24), -m));
    }
  } else {
    const float d = half_bits_to_float((uint16_t)w.d);
    const int q[4] = {w.q.x, w.q.y, w.q.z, w.q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = pack_bf16(d * (float)(int)(int8_t)(q[i] & 0xff), d * (float)(int)(int8_t)((q[i] >> 8) & 0xff));
      o[2 * i + 1] = pack_bf16(d * (float)(int)(int8_t)((q[i] >> 16) & 0xff), d * (float)(int)(int8_t)((unsigned)q[i] >> 24));
    }
  }
}

struct GemmBArgs {
  const uint8_t *w[3];
  float *out[3];
  int N[3], ldo[3], tile0[3];
  int nseg;
  const uint16_t *x;  // bf16 k-slab-major [K/64][M][64]: the A tile of one k-step is ONE contiguous 128 B x rows block
  int M, K, accumulate, splits, ldp, tn;
  float *partial;     // [splits][M][ldp] when splits > 1 (column = global n-tile * tn + n)
  size_t row_bytes;
  // grouped MoE mode (gemm_qb_kernel<.., MOE = true>, blockIdx.z = expert): rows of expert e are the sorted positions [bounds[e], bounds[e + 1]);
  // w[0] = experts stacked along the row axis, expert_stride bytes apart; xrows = rows of the slab layout (tokens when gathering, routes otherwise)
  const int32_t *bounds, *sorted;  // launch_moe_dispatch outputs: sorted[pos] = flat route index (token * topk + slot)
  const float *route_w;            // down projection: out[token][n] += route_w[flat] * acc (f32 atomics); NULL: out[pos][n] = acc
  int topk, gather, xrows;         // gather: the A row of position pos is token sorted[pos] / topk (gate / up); else pos itself (down)
  size_t expert_stride;
  // fused gate / up mode (GLU = true): w[0] = gate, w[1] = up ([N[0]][K] each); a 128-row B tile = 64 output columns: rows [0,32) gate cols 0-31, [32,64) up
  // cols 0-31, [64,96) gate cols 32-63, [96,128) up cols 32-63, so a wave's two 32-column MFMA tiles are (gate, up) of the SAME columns and the epilogue
  // writes act(gate) * up as bf16 slabs glu_out[N/64][M][64] (the down GEMM's activation layout) -- no f32 round trip, no GLU kernel
  uint16_t *glu_out; int activation;
};

// NI = 128-column blocks per workgroup tile: 1 -> 256 x 128 (waves 4 x 2, 64 x 64 each), 2 -> 256 x 256 (waves 2 x 4, 128 x 64 each:
// half the A traffic, LDS bytes and decode work per FLOP; needs >= ~200 column tiles of 256 to fill the chip without split-K).
template <int TYPE, int NI, bool MOE = false, bool GLU = false>
__global__ void __launch_bounds__(HT) gemm_qb_kernel(const GemmBArgs a) {
  constexpr int TN = HN * NI, MI = 2 * NI, PF = NI == 1 ? 2 : 1;  // PF: staging register sets (prefetch distance in k-steps)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 256 x 64 bf16 | B TN x 64 bf16] = 96 / 128 KiB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int seg = 0;
  if (a.nseg > 2 && (int)blockIdx.x >= a.tile0[2]) seg = 2;
  else if (a.nseg > 1 && (int)blockIdx.x >= a.tile0[1]) seg = 1;
  const int segN = a.N[seg];
  const int m0 = blockIdx.y * HM, n0 = GLU ? (int)blockIdx.x * 64 : ((int)blockIdx.x - a.tile0[seg]) * TN;
  int cnt = a.M, pos0 = 0;  // valid rows of this row tile's matrix, its first sorted position
  const uint8_t *wbase = a.w[seg];
  if constexpr (MOE) {
    pos0 = a.bounds[blockIdx.z];
    cnt = a.bounds[blockIdx.z + 1] - pos0;
    if (m0 >= cnt) return;  // workgroup-uniform: most (expert, row tile) pairs of the worst-case grid are empty
    wbase += (size_t)blockIdx.z * a.expert_stride;
  }
  const int wm = NI == 1 ? (wave & 3) * 64 : (wave & 1) * 128, wn = NI == 1 ? (wave >> 2) * 64 : (wave >> 1) * 64;
  const int ar = tid >> 2, aq = tid & 3;     // B: rows ar + 128 j, 16-weight quarter of the 64-k slab
  const int xc = tid & 7, xr0 = tid >> 3;    // A: 16-byte chunk, rows xr0 + 64 i
  const uint8_t *wrow[NI];
#pragma unroll
  for (int jn = 0; jn < NI; ++jn) {
    if constexpr (GLU) wrow[jn] = a.w[(ar >> 5) & 1] + (size_t)min(n0 + (ar >> 6) * 32 + (ar & 31), segN - 1) * a.row_bytes;
    else wrow[jn] = wbase + (size_t)min(n0 + ar + 128 * jn, segN - 1) * a.row_bytes;
  }
  const int nk_all = a.K / HK;
  const int kz = MOE ? 0 : (int)blockIdx.z;
  const int k_lo = (int)((long)nk_all * kz / a.splits), k_hi = (int)((long)nk_all * (kz + 1) / a.splits);
  const int nk = k_hi - k_lo;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A tile: buffer loads with sc1 (agent scope = miss-always in the 32 KB vector L1, normal L2 hit).  The tile is used once per
  // workgroup: with default loads its 32 KB per k-step flush the weight-block lines (re-used over 4 k-steps) out of L1 -- measured
  // 2x on the whole kernel; nontemporal loads fix L1 but also drop the lines from L2, which every other column tile re-reads (-15 %).
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const int xrows = MOE ? a.xrows : a.M;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, (short)0, (int)((size_t)xrows * a.K * 2), 0x00020000);
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = min(m0 + xr0 + 64 * i, cnt - 1);
    if constexpr (MOE) { row += pos0; if (a.gather) row = a.sorted[row] / a.topk; }
    xoff[i] = (unsigned)((row * HK + xc * 8) * 2);
  }
  v4u xa[PF][4];
  RawH<TYPE> wb[PF][NI];
  auto issue = [&](int set, int kb_raw) {  // unconditional, k index clamped (a load under a branch makes hipcc drain vmcnt)
    const int kb = k_lo + min(kb_raw, nk - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[set][i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff[i] + (unsigned)kb * (unsigned)(xrows * HK * 2), 0, 16);
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) wb[set][jn] = gemm_load_w16<TYPE>(wrow[jn], kb, aq);
  };
  auto commit = [&](int set, int kb_raw, char *buf) {
    const int kb = k_lo + min(kb_raw, nk - 1);
    char *A = buf, *B = buf + HM * HK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(v4u *)(A + tile_off(xr0 + 64 * i, xc)) = xa[set][i];
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      unsigned o[8];
      gemm_decode_w16<TYPE>(wb[set][jn], kb, aq, o);
      *(int4 *)(B + tile_off(ar + 128 * jn, aq * 2)) = make_int4((int)o[0], (int)o[1], (int)o[2], (int)o[3]);
      *(int4 *)(B + tile_off(ar + 128 * jn, aq * 2 + 1)) = make_int4((int)o[4], (int)o[5], (int)o[6], (int)o[7]);
    }
  };
  const int frow = lane & 31, fk = lane >> 5;
  auto mfma_tile = [&](const char *buf) {
    const char *A = buf, *B = buf + HM * HK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[MI], bfr[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8 *)(A + tile_off(wm + i * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8 *)(B + tile_off(wn + j * 32 + frow, ks * 2 + fk));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };
  char *buf0 = smem, *buf1 = smem + (HM + TN) * HK * 2;
  if constexpr (PF == 2) {
    issue(0, 0);
    issue(1, 1);
    commit(0, 0, buf0);
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      issue(0, kb + 2);
      mfma_tile(buf0);
      commit(1, kb + 1, buf1);
      __syncthreads();
      if (kb + 1 >= nk) break;
      issue(1, kb + 3);
      mfma_tile(buf1);
      commit(0, kb + 2, buf0);
      __syncthreads();
    }
  } else {
    issue(0, 0);
    commit(0, 0, buf0);
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      issue(0, kb + 1);
      mfma_tile(buf0);
      commit(0, kb + 1, buf1);
      __syncthreads();
      if (kb + 1 >= nk) break;
      issue(0, kb + 2);
      mfma_tile(buf1);
      commit(0, kb + 2, buf0);
      __syncthreads();
    }
  }
  if constexpr (GLU) {
    const int n = n0 + (wn >> 6) * 32 + (lane & 31);
    uint16_t *y = a.glu_out + (size_t)(n >> 6) * a.M * 64 + (n & 63);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < segN) y[(size_t)m * 64] = (uint16_t)(pack_bf16(glu_act(acc[i][0][r], a.activation) * acc[i][1][r], 0.f) & 0xffffu);
      }
    return;
  }
  float *obase;
  int ldo, ncol0;
  if (a.splits > 1) { obase = a.partial + (size_t)blockIdx.z * a.M * a.ldp; ldo = a.ldp; ncol0 = (int)blockIdx.x * TN - n0; }
  else { obase = a.out[seg]; ldo = a.ldo[seg]; ncol0 = 0; }
  const bool accum = a.accumulate && a.splits == 1;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < cnt && n < segN) {
          if constexpr (MOE) {
            if (a.route_w) {
              const int flat = a.sorted[pos0 + m];
              atomicAdd(obase + (size_t)(flat / a.topk) * ldo + n, a.route_w[flat] * acc[i][j][r]);
            } else {
              obase[(size_t)(pos0 + m) * ldo + n] = acc[i][j][r];
            }
          } else {
            float *p = obase + (size_t)m * ldo + ncol0 + n;
            *p = accum ? *p + acc[i][j][r] : acc[i][j][r];
          }
        }
      }
    }
}

// Wave-specialised variant of the 256 x 128 x 64 tile (same operands, same MFMA order, identical bits): waves 0-3 are CONSUMERS (one per
// SIMD, 128 x 64 each = 4 x 2 MFMA tiles, 128 accumulator registers): ds_read_b128 fragments + MFMA and nothing else; waves 4-7 are
// PRODUCERS (one per SIMD): global loads two k-steps ahead, block decode to bf16, LDS writes.  The two kinds share each SIMD, so the decode
// VALU of a k-step runs under the MFMAs of the previous one by hardware wave interleaving instead of by compiler scheduling inside one
// instruction stream (gemm_qb_kernel: every wave alternates both phases and the SIMD partners fall into lockstep between the barriers).
// LDS per k-step: 48 KiB written + 96 KiB read (a 128 x 64 wave tile reads 24 KiB per 32 MFMAs instead of 16 KiB per 16).
// One barrier per k-step, hit by both kinds (s_barrier counts waves, not call sites).
#ifndef MRS_GEMM_ABLATE
#define MRS_GEMM_ABLATE 0  // experiment builds only (profiles/experiments/build_gemm_ablate.sh): 1 no decode arithmetic, 2 producers idle, 4 no MFMA, 8 no fragment reads, 16 no A copies
#endif
template <int TYPE, bool GLU = false>
__global__ void __launch_bounds__(HT) gemm_qc_kernel(const GemmBArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 256 x 64 bf16 | B 128 x 64 bf16] = 96 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int seg = 0;
  if (a.nseg > 2 && (int)blockIdx.x >= a.tile0[2]) seg = 2;
  else if (a.nseg > 1 && (int)blockIdx.x >= a.tile0[1]) seg = 1;
  const int segN = a.N[seg];
  const int m0 = blockIdx.y * HM, n0 = GLU ? (int)blockIdx.x * 64 : ((int)blockIdx.x - a.tile0[seg]) * HN;
  const int nk_all = a.K / HK;
  const int k_lo = (int)((long)nk_all * blockIdx.z / a.splits), k_hi = (int)((long)nk_all * (blockIdx.z + 1) / a.splits);
  const int nk = k_hi - k_lo;
  char *buf0 = smem, *buf1 = smem + (HM + HN) * HK * 2;

  if (wave >= 4) {  // ---------------- producers
    // B: one 32-weight sub-block per thread and k-step (block header decoded once per 32 weights instead of once per 16; rows 2 apart
    // on neighbouring lanes so that the 8 lanes of a ds_write_b128 group carry 8 distinct swizzle keys); A: 8 x 16 B per thread.  The
    // VALU budget beside the consumers' 32 MFMAs per k-step is ~5 single-issue instructions per MFMA and SIMD (MI355X_MICROARCH): PMC
    // showed 2 x 137 (gemm_qb_kernel) / 234 (16-weight units) VALU per SIMD and k-step against that ~160.
    const int p = tid - 256;
    const int half = p >> 7, q7 = p & 127, ar = ((q7 & 63) << 1) | (q7 >> 6);
    const int xc = p & 7, xr0 = p >> 3;  // A: 16-byte chunk, rows xr0 + 32 i
    const uint8_t *wrow = GLU ? a.w[(ar >> 5) & 1] + (size_t)min(n0 + (ar >> 6) * 32 + (ar & 31), segN - 1) * a.row_bytes
                              : a.w[seg] + (size_t)min(n0 + ar, segN - 1) * a.row_bytes;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, (short)0, (int)((size_t)a.M * a.K * 2), 0x00020000);
    unsigned xoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xoff[i] = (unsigned)((min(m0 + xr0 + 32 * i, a.M - 1) * HK + xc * 8) * 2);
    const int slab_bytes = a.M * HK * 2;
    v4u xa[2][8];
    RawG<TYPE> wb[2];
    auto issue = [&](int set, int kb_raw) {  // unconditional, k index clamped
      const int kb = k_lo + min(kb_raw, nk - 1);
      if constexpr (MRS_GEMM_ABLATE & 2) return;
      if constexpr (!(MRS_GEMM_ABLATE & 16)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[set][i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff[i], kb * slab_bytes, 16);  // slab offset in an SGPR
      }
      wb[set] = gemm_load_w<TYPE>(wrow, kb, half);
    };
    auto commit = [&](int set, int kb_raw, char *buf) {
      const int kb = k_lo + min(kb_raw, nk - 1);
      char *A = buf, *B = buf + HM * HK * 2;
      if constexpr (MRS_GEMM_ABLATE & 2) return;
      if constexpr (!(MRS_GEMM_ABLATE & 16)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(v4u *)(A + tile_off(xr0 + 32 * i, xc)) = xa[set][i];
      }
      unsigned o[16];
      if constexpr (MRS_GEMM_ABLATE & 1) {
        const int *r = (const int *)&wb[set];
#pragma unroll
        for (int c = 0; c < 16; ++c) o[c] = (unsigned)r[c % (int)(sizeof(wb[set]) / 4)];
      } else {
        gemm_decode_w<TYPE>(wb[set], kb, half, o);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) *(int4 *)(B + tile_off(ar, half * 4 + c)) = make_int4((int)o[4 * c], (int)o[4 * c + 1], (int)o[4 * c + 2], (int)o[4 * c + 3]);
    };
    issue(0, 0);
    issue(1, 1);
    commit(0, 0, buf0);
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      issue(0, kb + 2);
      commit(1, kb + 1, buf1);
      __syncthreads();
      if (kb + 1 >= nk) break;
      issue(1, kb + 3);
      commit(0, kb + 2, buf0);
      __syncthreads();
    }
    return;
  }

  // ---------------- consumers
  const int wm = (wave & 1) * 128, wn = (wave >> 1) * 64;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fk = lane >> 5;
  // fragment addresses: tile_off(row, chunk) with row = 32 * t + frow (+ wm / wn, multiples of 32): the swizzle key (row >> 1) & 7 depends on frow
  // only, so a stage needs one base per operand + one 16-byte-chunk offset per k-slab; the tile index is an immediate offset of the ds_read
  const int key = (frow >> 1) & 7;
  const int abase = (wm + frow) * 128, bbase = HM * HK * 2 + (wn + frow) * 128;
  int ko[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ko[ks] = ((ks * 2 + fk) ^ key) << 4;
  // Software pipeline over the four 16-k slabs of a stage: the 6 fragment reads of slab s + 1 are threaded between the 8 MFMAs of slab s
  // (hipcc otherwise sinks every read to just before its first use and the matrix pipe idles on LDS latency); the barrier sits in front
  // of the LAST slab's MFMAs (they only need registers), and the reads of the next stage's slab 0 go under them.
  bf16x8 af[2][4], bfr[2][2];
  auto rd = [&](int set, const char *buf, int ks) {
    if constexpr (MRS_GEMM_ABLATE & 8) { if (nk > 0) return; }
#pragma unroll
    for (int i = 0; i < 4; ++i) af[set][i] = *(const bf16x8 *)(buf + abase + ko[ks] + i * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[set][j] = *(const bf16x8 *)(buf + bbase + ko[ks] + j * 4096);
  };
  auto mm = [&](int set) {
    if constexpr (MRS_GEMM_ABLATE & 4) {  // keep the fragment reads alive without the matrix pipe
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j][0] += __builtin_bit_cast(float, (int)af[set][i][0] ^ (int)bfr[set][j][0]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bfr[set][j], acc[i][j], 0, 0, 0);
  };
  constexpr int STAGE = (HM + HN) * HK * 2;
  int st = 0;
  __syncthreads();
  rd(0, smem, 0);
  for (int kb = 0; kb < nk; ++kb) {
    const char *buf = smem + st;
    rd(1, buf, 1); mm(0);
    rd(0, buf, 2); mm(1);
    rd(1, buf, 3); mm(0);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __syncthreads();
    st ^= STAGE;
    rd(0, smem + st, 0);  // after the last stage: a harmless read of stale LDS
    mm(1);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
  }
  if constexpr (GLU) {
    const int n = n0 + (wn >> 6) * 32 + (lane & 31);
    uint16_t *y = a.glu_out + (size_t)(n >> 6) * a.M * 64 + (n & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < segN) y[(size_t)m * 64] = (uint16_t)(pack_bf16(glu_act(acc[i][0][r], a.activation) * acc[i][1][r], 0.f) & 0xffffu);
      }
    return;
  }
  float *obase;
  int ldo, ncol0;
  if (a.splits > 1) { obase = a.partial + (size_t)blockIdx.z * a.M * a.ldp; ldo = a.ldp; ncol0 = (int)blockIdx.x * HN - n0; }
  else { obase = a.out[seg]; ldo = a.ldo[seg]; ncol0 = 0; }
  const bool accum = a.accumulate && a.splits == 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < segN) {
          float *p = obase + (size_t)m * ldo + ncol0 + n;
          *p = accum ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

// out[seg][m][n] (+)= sum_z partial[z][m][tile0[seg]*128 + n], z ascending (fixed order)
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const GemmBArgs a) {
  const int total_cols = a.ldp;
  const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= (size_t)a.M * total_cols) return;
  const int m = (int)(idx / total_cols), gc = (int)(idx % total_cols);
  int seg = 0;
  if (a.nseg > 2 && gc >= a.tile0[2] * a.tn) seg = 2;
  else if (a.nseg > 1 && gc >= a.tile0[1] * a.tn) seg = 1;
  const int n = gc - a.tile0[seg] * a.tn;
  if (n >= a.N[seg]) return;  // N % 4 == 0 is checked by the launcher for split launches
  float4 s = *(const float4 *)(a.partial + (size_t)m * a.ldp + gc);
  for (int z = 1; z < a.splits; ++z) {
    const float4 t = *(const float4 *)(a.partial + ((size_t)z * a.M + m) * a.ldp + gc);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  float4 *o = (float4 *)(a.out[seg] + (size_t)m * a.ldo[seg] + n);
  if (a.accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  *o = s;
}

// x f32 [M][ldx] -> bf16 k-slab-major y[K/64][M][64] (row-major activations put the 256 rows of an A tile 2*K bytes apart -- a power of
// two for the usual K, i.e. on one or two L2 channels: measured 2.6x slower GEMM than with the tile contiguous)
__global__ void __launch_bounds__(256) convert_f32_bf16_slabs_kernel(const float *__restrict__ x, uint16_t *__restrict__ y, int ldx, int M, int K) {
  const int k8 = K / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * k8) return;
  const int m = (int)(i / k8), k = (int)(i % k8) * 8;
  const float4 a = *(const float4 *)(x + (size_t)m * ldx + k), b = *(const float4 *)(x + (size_t)m * ldx + k + 4);
  *(int4 *)(y + ((size_t)(k >> 6) * M + m) * 64 + (k & 63)) =
      make_int4((int)pack_bf16(a.x, a.y), (int)pack_bf16(a.z, a.w), (int)pack_bf16(b.x, b.y), (int)pack_bf16(b.z, b.w));
}

// Producers that write the slab layout directly (no f32 round trip + convert pass):
// act(g) * u, f32 [M][ld] x2 -> bf16 slabs [N/64][M][64]; same arithmetic as fused_glu_f32 followed by the conversion
__global__ void __launch_bounds__(256) glu_bf16_slabs_kernel(const float *__restrict__ g, const float *__restrict__ u, uint16_t *__restrict__ y,
                                                             int ld, int M, int N, int activation) {
  const int n8 = N / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * n8) return;
  const int m = (int)(i / n8), k = (int)(i % n8) * 8;
  const float4 g0 = *(const float4 *)(g + (size_t)m * ld + k), g1 = *(const float4 *)(g + (size_t)m * ld + k + 4);
  const float4 u0 = *(const float4 *)(u + (size_t)m * ld + k), u1 = *(const float4 *)(u + (size_t)m * ld + k + 4);
  const float o0 = glu_act(g0.x, activation) * u0.x, o1 = glu_act(g0.y, activation) * u0.y, o2 = glu_act(g0.z, activation) * u0.z,
              o3 = glu_act(g0.w, activation) * u0.w, o4 = glu_act(g1.x, activation) * u1.x, o5 = glu_act(g1.y, activation) * u1.y,
              o6 = glu_act(g1.z, activation) * u1.z, o7 = glu_act(g1.w, activation) * u1.w;
  *(int4 *)(y + ((size_t)(k >> 6) * M + m) * 64 + (k & 63)) = make_int4((int)pack_bf16(o0, o1), (int)pack_bf16(o2, o3), (int)pack_bf16(o4, o5), (int)pack_bf16(o6, o7));
}

// RMSNorm (x * rsqrt(mean(x^2) + eps) * w, the arithmetic and reduction order of mrs_rms_norm_f32) -> bf16 slabs; one workgroup per row
__global__ void __launch_bounds__(256) rms_norm_bf16_slabs_kernel(const float *__restrict__ x, const float *__restrict__ w, uint16_t *__restrict__ y,
                                                                  int M, int K, float eps) {
  __shared__ float red[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float *xr = x + (size_t)m * K;
  float sum = 0.f;
  for (int v = tid; v < K / 4; v += 256) {
    const float4 t = *(const float4 *)(xr + v * 4);
    sum = fmaf(t.x, t.x, sum); sum = fmaf(t.y, t.y, sum); sum = fmaf(t.z, t.z, sum); sum = fmaf(t.w, t.w, sum);
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);  // block_sum_256 order of core_ops.hip
  for (int v = tid; v < K / 8; v += 256) {
    const int k = v * 8;
    const float4 a = *(const float4 *)(xr + k), b = *(const float4 *)(xr + k + 4), wa = *(const float4 *)(w + k), wb = *(const float4 *)(w + k + 4);
    *(int4 *)(y + ((size_t)(k >> 6) * M + m) * 64 + (k & 63)) =
        make_int4((int)pack_bf16(a.x * inv * wa.x, a.y * inv * wa.y), (int)pack_bf16(a.z * inv * wa.z, a.w * inv * wa.w),
                  (int)pack_bf16(b.x * inv * wb.x, b.y * inv * wb.y), (int)pack_bf16(b.z * inv * wb.z, b.w * inv * wb.w));
  }
}

// 1 = wave-specialised gemm_qc_kernel, 0 = gemm_qb_kernel, -1 (default) = by weight type: measured on the MI355X (profiles/round2_prefill_gemm.md) the
// producer / consumer split wins for Q4_K (gate/up 692 -> 631 us, down 312 -> 288 us at T = 2048), is neutral for Q5_K / Q8_0 and loses for Q6_K
// (its 32-weight unit loads four 2-byte-aligned 16-B pieces).  MRS_GEMM_VARIANT / mrs_gemm_set_variant: tests compare the two bit for bit.
static int &gemm_variant() { static int v = [] { const char *e = getenv("MRS_GEMM_VARIANT"); return e ? atoi(e) : -1; }(); return v; }
template <int TYPE, int NI> static void gemm_b_launch_ni(GemmBArgs a, size_t ws_bytes, hipStream_t s) {
  constexpr int TN = HN * NI;
  int tiles = 0;
  bool n4 = true;
  for (int i = 0; i < a.nseg; ++i) { a.tile0[i] = tiles; tiles += (a.N[i] + TN - 1) / TN; n4 = n4 && a.N[i] % 4 == 0 && a.ldo[i] % 4 == 0; }
  const int mt = (a.M + HM - 1) / HM, nk = a.K / HK;
  int splits = 1;
  if (const char *e = getenv("MRS_GEMM_SPLITS")) splits = atoi(e);
  else if (tiles * mt < 192) splits = std::min(std::min(8, 256 / (tiles * mt)), std::max(1, nk / 8));
  a.ldp = tiles * TN;
  a.tn = TN;
  if (!n4 || !a.partial) splits = 1;
  while (splits > 1 && (size_t)splits * a.M * a.ldp * 4 > ws_bytes) --splits;
  if (splits < 1) splits = 1;
  a.splits = splits;
  constexpr size_t lds = 2 * (HM + TN) * HK * 2;
  static bool attr = false;
  auto kern = NI == 1 && (gemm_variant() == 1 || (gemm_variant() < 0 && TYPE == T_Q4_K)) ? gemm_qc_kernel<TYPE> : gemm_qb_kernel<TYPE, NI>;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)gemm_qb_kernel<TYPE, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)gemm_qc_kernel<TYPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, mt, splits), dim3(HT), lds, s, a);
  if (splits > 1) {
    const size_t n4s = (size_t)a.M * a.ldp / 4;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((n4s + 255) / 256)), dim3(256), 0, s, a);
  }
}
template <int TYPE> static int gemm_b_launch(const GemmBArgs &a, size_t ws_bytes, hipStream_t s) {
  // NI = 2 (256 x 256 tiles, 128 x 64 per wave) needs ~280 registers per lane with this staging scheme and spills under the 256 budget
  // of a 512-thread workgroup (measured 3x slower); it stays un-instantiated until the A tile moves to LDS-DMA (round 2).
  gemm_b_launch_ni<TYPE, 1>(a, ws_bytes, s);
  return 0;
}

}  // namespace mrs

using namespace mrs;

// x f32 [M][ldx] (ldx % 4 == 0, K % 64 == 0) -> bf16 (RNE) in the k-slab-major layout y[K/64][M][64] that mrs_gemm_q_bf16_multi reads.
extern "C" int mrs_convert_f32_bf16_slabs(const float *x, int ldx, int M, int K, void *y, void *stream) {
  if (K <= 0 || K % 64 || (ldx & 3)) return -1;
  if (M <= 0) return 0;
  const size_t n = (size_t)M * (K / 8);
  hipLaunchKernelGGL(convert_f32_bf16_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t *)y, ldx, M, K);
  return 0;
}

// act(g[m][n]) * u[m][n] (f32, row stride ld, N % 64 == 0) -> bf16 slabs y[N/64][M][64]: fused_glu + mrs_convert_f32_bf16_slabs in one pass
extern "C" int mrs_glu_bf16_slabs(const float *g, const float *u, int ld, int M, int N, int activation, void *y, void *stream) {
  if (N <= 0 || N % 64 || (ld & 3)) return -1;
  if (M <= 0) return 0;
  const size_t n = (size_t)M * (N / 8);
  hipLaunchKernelGGL(glu_bf16_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, u, (uint16_t *)y, ld, M, N, activation);
  return 0;
}
// RMSNorm of x f32 [M][K] (contiguous rows, K % 64 == 0) with weight w [K] -> bf16 slabs y[K/64][M][64]: mrs_rms_norm_f32 + conversion
extern "C" int mrs_rms_norm_bf16_slabs(const float *x, const float *w, int M, int K, float eps, void *y, void *stream) {
  if (K <= 0 || K % 64) return -1;
  if (M <= 0) return 0;
  hipLaunchKernelGGL(rms_norm_bf16_slabs_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, w, (uint16_t *)y, M, K, eps);
  return 0;
}

// Large-M GEMM over bf16 activations in k-slab-major layout x[K/64][M][64] (mrs_convert_f32_bf16_slabs): out_s[m*ldo_s + n] (+)= sum_k x[m][k] * bf16(W_s[n][k]).
// workspace (may be NULL): split-K partials for shapes with fewer tiles than CUs; mrs_gemm_q_bf16_workspace_bytes() always suffices.
// Grouped MoE GEMM of a prompt on the matrix cores (role of launch_moe_grouped_gemm_<t> / moe_grouped.cu:1180-1235 for long prompts, same
// dispatch tables): for every expert e and sorted position pos in [bounds[e], bounds[e + 1]): acc[n] = sum_k bf16(x[row][k]) * bf16(dequant(W_e)[n][k]),
// row = gather ? sorted[pos] / topk : pos;  route_w ? atomicAdd(out[sorted[pos] / topk][n], route_w[sorted[pos]] * acc) : out[pos][n] = acc.
// x_slabs = bf16 slabs [K/64][x_rows][64]; w = [E * N][K] blocks; routes = number of sorted positions (grid covers the worst case: all on one expert).
// Arithmetic and k order of mrs_gemm_q_bf16_multi: per expert bit-identical to that GEMM on the expert's rows.
extern "C" int mrs_moe_gemm_q_bf16(const void *w, int ggml_type, int N, int K, int num_experts, const void *x_slabs, int x_rows, const int32_t *bounds,
                                   const int32_t *sorted, int topk, int gather, const float *route_w, float *out, int ldo, int routes, void *stream) {
  if (!w || !x_slabs || !bounds || !sorted || !out || N <= 0 || num_experts <= 0 || topk <= 0 || x_rows <= 0) return -1;
  if (routes <= 0) return 0;
  if (K <= 0 || K % 64 || ((ggml_type == T_Q4_K || ggml_type == T_Q5_K || ggml_type == T_Q6_K) && K % 256)) return -1;
  GemmBArgs a{};
  a.nseg = 1; a.w[0] = (const uint8_t *)w; a.out[0] = out; a.N[0] = N; a.ldo[0] = ldo; a.tile0[0] = 0;
  a.x = (const uint16_t *)x_slabs; a.M = routes; a.K = K; a.splits = 1; a.tn = HN;
  a.bounds = bounds; a.sorted = sorted; a.route_w = route_w; a.topk = topk; a.gather = gather; a.xrows = x_rows;
  const dim3 grid((N + HN - 1) / HN, (routes + HM - 1) / HM, num_experts);
  constexpr size_t lds = 2 * (HM + HN) * HK * 2;
  auto go = [&](auto kern, size_t row_bytes) {
    a.row_bytes = row_bytes; a.expert_stride = (size_t)N * row_bytes;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kern, grid, dim3(HT), lds, (hipStream_t)stream, a);
    return 0;
  };
  switch (ggml_type) {
  case T_Q4_K: return go(gemm_qb_kernel<T_Q4_K, 1, true>, (size_t)(K / 256) * 144);
  case T_Q5_K: return go(gemm_qb_kernel<T_Q5_K, 1, true>, (size_t)(K / 256) * 176);
  case T_Q6_K: return go(gemm_qb_kernel<T_Q6_K, 1, true>, (size_t)(K / 256) * 210);
  case T_Q8_0: return go(gemm_qb_kernel<T_Q8_0, 1, true>, (size_t)(K / 32) * 34);
  default: return -1;
  }
}
// Fused gate / up GEMM of a prompt: y = act(W_g x) * (W_u x) as bf16 slabs y[N/64][M][64] (the down GEMM's activation layout) -- the role of
// fast_mmq::fused_glu / fused_ffn's first half (gguf/fast_mmq.rs:762-821).  Same MFMA arithmetic as two mrs_gemm_q_bf16_multi launches followed by
// mrs_glu_bf16_slabs: identical bits.  y must not alias x_slabs.  N % 64 == 0.
extern "C" int mrs_gemm_q_bf16_glu(const void *w_gate, const void *w_up, int ggml_type, int N, int K, const void *x_slabs, int M, int activation, void *y_slabs,
                                   void *stream) {
  if (!w_gate || !w_up || !x_slabs || !y_slabs || y_slabs == x_slabs || N <= 0 || N % 64) return -1;
  if (M <= 0) return 0;
  if (K <= 0 || K % 64 || ((ggml_type == T_Q4_K || ggml_type == T_Q5_K || ggml_type == T_Q6_K) && K % 256)) return -1;
  GemmBArgs a{};
  a.nseg = 1; a.w[0] = (const uint8_t *)w_gate; a.w[1] = (const uint8_t *)w_up; a.N[0] = N; a.tile0[0] = 0;
  a.x = (const uint16_t *)x_slabs; a.M = M; a.K = K; a.splits = 1; a.tn = HN; a.glu_out = (uint16_t *)y_slabs; a.activation = activation;
  const dim3 grid(N / 64, (M + HM - 1) / HM, 1);
  constexpr size_t lds = 2 * (HM + HN) * HK * 2;
  auto go = [&](auto kern, size_t row_bytes) {
    a.row_bytes = row_bytes;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kern, grid, dim3(HT), lds, (hipStream_t)stream, a);
    return 0;
  };
  const bool qc = gemm_variant() == 1 || (gemm_variant() < 0 && ggml_type == T_Q4_K);
  switch (ggml_type) {
  case T_Q4_K: return qc ? go(gemm_qc_kernel<T_Q4_K, true>, (size_t)(K / 256) * 144) : go(gemm_qb_kernel<T_Q4_K, 1, false, true>, (size_t)(K / 256) * 144);
  case T_Q5_K: return qc ? go(gemm_qc_kernel<T_Q5_K, true>, (size_t)(K / 256) * 176) : go(gemm_qb_kernel<T_Q5_K, 1, false, true>, (size_t)(K / 256) * 176);
  case T_Q6_K: return qc ? go(gemm_qc_kernel<T_Q6_K, true>, (size_t)(K / 256) * 210) : go(gemm_qb_kernel<T_Q6_K, 1, false, true>, (size_t)(K / 256) * 210);
  case T_Q8_0: return qc ? go(gemm_qc_kernel<T_Q8_0, true>, (size_t)(K / 32) * 34) : go(gemm_qb_kernel<T_Q8_0, 1, false, true>, (size_t)(K / 32) * 34);
  default: return -1;
  }
}
extern "C" void mrs_gemm_set_variant(int v) { mrs::gemm_variant() = v; }
extern "C" size_t mrs_gemm_q_bf16_workspace_bytes(int M) { return (size_t)M * 65536 * 4 / (size_t)((M + HM - 1) / HM) + 65536; }
extern "C" int mrs_gemm_q_bf16_multi(int nseg, const void *const *w, const int *N, float *const *out, const int *ldo, int ggml_type, int K,
                                     const void *x_slabs, int M, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  if (nseg < 1 || nseg > 3) return -1;
  if (M <= 0) return 0;
  if (K <= 0 || K % 64 || ((ggml_type == T_Q4_K || ggml_type == T_Q5_K || ggml_type == T_Q6_K) && K % 256)) return -1;
  GemmBArgs a{};
  a.nseg = nseg; a.x = (const uint16_t *)x_slabs; a.M = M; a.K = K; a.accumulate = accumulate;
  a.partial = (float *)workspace;
  for (int i = 0; i < nseg; ++i) {
    if (N[i] <= 0) return -1;
    a.w[i] = (const uint8_t *)w[i]; a.out[i] = out[i]; a.N[i] = N[i]; a.ldo[i] = ldo[i];
  }
  switch (ggml_type) {
  case T_Q4_K: a.row_bytes = (size_t)(K / 256) * 144; return gemm_b_launch<T_Q4_K>(a, workspace_bytes, (hipStream_t)stream);
  case T_Q5_K: a.row_bytes = (size_t)(K / 256) * 176; return gemm_b_launch<T_Q5_K>(a, workspace_bytes, (hipStream_t)stream);
  case T_Q6_K: a.row_bytes = (size_t)(K / 256) * 210; return gemm_b_launch<T_Q6_K>(a, workspace_bytes, (hipStream_t)stream);
  case T_Q8_0: a.row_bytes = (size_t)(K / 32) * 34; return gemm_b_launch<T_Q8_0>(a, workspace_bytes, (hipStream_t)stream);
  default: return -1;
  }
}

// Up to 3 weight matrices of ONE type sharing the activations: out_s[m*ldo_s + n] (+)= sum_k bf16(x[m*ldx + k]) * bf16(W_s[n][k]).
extern "C" int mrs_gemm_q_f32_multi(int nseg, const void *const *w, const int *N, float *const *out, const int *ldo, int ggml_type, int K,
                                    const float *x, int ldx, int M, int accumulate, void *stream) {
  if (nseg < 1 || nseg > 3) return -1;
  if (M <= 0) return 0;
  if (K <= 0 || K % 64 || ((ggml_type == T_Q4_K || ggml_type == T_Q5_K || ggml_type == T_Q6_K) && K % 256) || (ldx & 3)) return -1;
  GemmArgs a{};
  a.nseg = nseg; a.x = x; a.M = M; a.K = K; a.ldx = ldx; a.accumulate = accumulate;
  for (int i = 0; i < nseg; ++i) {
    if (N[i] <= 0) return -1;
    a.w[i] = (const uint8_t *)w[i]; a.out[i] = out[i]; a.N[i] = N[i]; a.ldo[i] = ldo[i];
  }
  switch (ggml_type) {
  case T_Q4_K: a.row_bytes = (size_t)(K / 256) * 144; return gemm_launch<T_Q4_K>(a, (hipStream_t)stream);
  case T_Q5_K: a.row_bytes = (size_t)(K / 256) * 176; return gemm_launch<T_Q5_K>(a, (hipStream_t)stream);
  case T_Q6_K: a.row_bytes = (size_t)(K / 256) * 210; return gemm_launch<T_Q6_K>(a, (hipStream_t)stream);
  case T_Q8_0: a.row_bytes = (size_t)(K / 32) * 34; return gemm_launch<T_Q8_0>(a, (hipStream_t)stream);
  default: return -1;
  }
}

// out[m*ldo + n] (+)= sum_k bf16(x[m*ldx + k]) * bf16(W[n][k]);  W: raw GGUF blocks [N][K/blk] of type q4_k / q5_k / q6_k / q8_0.
// Returns 0, or -1 for an unsupported type / shape (K % 256 for the K-quants, K % 64 for Q8_0).
extern "C" int mrs_gemm_q_f32(const void *w, int ggml_type, int N, int K, const float *x, int ldx, float *out, int ldo, int M, int accumulate,
                              void *stream) {
  if (N <= 0) return 0;
  return mrs_gemm_q_f32_multi(1, &w, &N, &out, &ldo, ggml_type, K, x, ldx, M, accumulate, stream);
}
