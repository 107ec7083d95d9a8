// ext_gemm_qi.hip -- prompt GEMM in the arithmetic of the reference CPU path, on the MI355X matrix cores, bit-identical to the decode engine.
//
// What the reference does on a CPU device for a T-token prompt: every linear goes through the same QMatMul f32 fallback as decode
// (mistralrs-quant/src/gguf/mod.rs:465-478): each activation ROW is quantized to Q8_K (K-quants), every output is a sum over the row's superblocks of
// (d_w d_x) <integer dot> - (dmin_w d_x) <integer min term>.  Rounds 1-3 ran prompts in a different arithmetic (weights and activations rounded to bf16);
// this kernel computes the reference's integers EXACTLY on v_mfma_f32_32x32x16_f16 and combines them in the f32 order of the decode engine ("ORD-U",
// dec_core2.cuh / oracle orc_gemv_engine), so a token's logits and KV pages do not depend on whether it was part of a prompt or decoded:
//   * the 6-bit sub-block scale is folded INTO the weight operand as an exact small integer in f16: Q4_K  sc (q - 8), |.| <= 504;  Q6_K  two operands
//     sl (q - 32) and sh (q - 32) with sc = 16 sh + sl, |.| <= 480 / 256.  The activation operand is the Q8_K quant as f16.  Every product and every partial
//     sum of a superblock is an integer below 2^24, so the f32 accumulator of the matrix core holds the EXACT sum (tests/test_gemm_qi.py holds
//     v_mfma_f32_32x32x16_f16 to that on adversarial operands); no per-sub-block scale multiply is left on the vector ALU;
//   * Q4_K: isum = X + 8 Y with Y = sum_j sc_j bsum_j, msum = sum_j m_j bsum_j: two more MFMAs per superblock (K = the 16 run sums of the Q8_K block,
//     scales duplicated per run), so (float)isum = fma(8, Y, X) and (float)msum = M are single roundings of exact integers -- what (float)int gives;
//   * per superblock and output ONE f32 term T (the decode engine's expression), terms added left to right inside each of the row's four runs of
//     superblocks, the four run sums left to right (dec_core2.cuh header).
// Operands: weights in an MFMA-order copy made at load time (mrs_gemm_qi_repack: per 32-row panel and superblock, every lane's 16 bytes of a piece
// contiguous: one buffer_load_dwordx4 per lane and piece, straight into registers); activations from mrs_qi_quantize (engine-order RmsNorm + the candle
// Q8_K quantizer of dec_core2.cuh, then f16 in the operand's k order, slab-major [superblock][token][256]).
// Tiling: workgroup 256 threads = 4 waves (one per SIMD), tile 128 weight rows x 128 tokens, wave 64 x 64 = 2 x 2 MFMA tiles; per superblock the 128
// tokens' f16 quants (64 KiB) are staged in LDS (16-byte chunks XOR-swizzled by the token index: conflict-free ds_read_b128 fragments).
#include "dec_core2.cuh"
#include <stdio.h>
#include <stdlib.h>

namespace mrs {
namespace qi {
using namespace mrs::dec2;

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int REC_Q4K = 4096 + 512 + 128, REC_Q6K = 8192 + 512 + 128;
__host__ __device__ constexpr int rec_bytes_qi(int type) { return type == T_Q4_K ? REC_Q4K : REC_Q6K; }
__host__ __device__ inline bool qi_type(int t) { return t == T_Q4_K || t == T_Q6_K; }
__host__ __device__ inline size_t qi_tensor_bytes(int type, long long n, long long k) { return (size_t)((n + 31) / 32) * (size_t)(k / 256) * rec_bytes_qi(type); }

// the k order inside a group of 8 operand slots: slot jj holds element PERM[jj] (the half2 registers of the weight operand are (e0, e2), (e1, e3), (e4, e6), (e5, e7):
// what two masks and a shift take out of a dword of nibbles / bytes)
__host__ __device__ constexpr int perm8(int jj) { return jj == 1 ? 2 : jj == 2 ? 1 : jj == 5 ? 6 : jj == 6 ? 5 : jj; }

// ------------------------------------------------------------------------------------------------ weights: GGUF blocks -> MFMA-order copy
// one thread per (panel, superblock, lane)
template <int TYPE>
__global__ void __launch_bounds__(256) qi_repack_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long long n, int K, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int S = K / 256, lane = (int)(i & 63);
  const long long ps = i >> 6;
  const int sb = (int)(ps % S);
  const long long panel = ps / S, row = panel * 32 + (lane & 31);
  const int hf = lane >> 5;
  const bool have = row < n;
  uint8_t *rec = dst + (size_t)ps * rec_bytes_qi(TYPE);
  alignas(16) uint8_t buf[16];
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *b = src + ((size_t)row * S + sb) * 144, *qs = b + 16;
    for (int c = 0; c < 4; ++c) {
      for (int k = 0; k < 8; ++k) { buf[k] = have ? qs[32 * c + 8 * hf + k] : 0x88; buf[8 + k] = have ? qs[32 * c + 16 + 8 * hf + k] : 0x88; }  // 0x88: q = 8 -> operand 0
      *(v4u *)(rec + ((size_t)c * 64 + lane) * 16) = *(const v4u *)buf;
    }
    if (hf == 0) {
      const int nn = lane & 31;
      for (int g = 0; g < 8; ++g) {  // get_scale_min_k4
        uint8_t sc = 0, mn = 0;
        if (have) {
          const uint8_t *p = b + 4;
          if (g < 4) { sc = p[g] & 63; mn = p[g + 4] & 63; } else { sc = (p[g + 4] & 15) | ((p[g - 4] >> 6) << 4); mn = (p[g + 4] >> 4) | ((p[g] >> 6) << 4); }
        }
        buf[g] = sc; buf[8 + g] = mn;
      }
      *(v4u *)(rec + 4096 + (size_t)nn * 16) = *(const v4u *)buf;
      *(uint32_t *)(rec + 4608 + (size_t)nn * 4) = have ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) : 0u;
    }
  } else {  // Q6_K: the 6-bit values as bytes (0 .. 63), 16 int8 scales, d
    const uint8_t *b = src + ((size_t)row * S + sb) * 210, *ql = b, *qh = b + 128;
    auto q6 = [&](int e) { const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32, ii = hh * 64 + pos + (qt % 2) * 32;
                           const int lo = qt < 2 ? (ql[ii] & 15) : (ql[ii] >> 4), hi = (qh[hh * 32 + pos] >> (qt * 2)) & 3; return lo | (hi << 4); };
    for (int g = 0; g < 8; ++g) {  // piece g: runs 2g, 2g + 1; the lane's 8 elements 8 hf .. 8 hf + 7 of each
      for (int k = 0; k < 8; ++k) { buf[k] = have ? (uint8_t)q6((2 * g) * 16 + 8 * hf + k) : 32; buf[8 + k] = have ? (uint8_t)q6((2 * g + 1) * 16 + 8 * hf + k) : 32; }
      *(v4u *)(rec + ((size_t)g * 64 + lane) * 16) = *(const v4u *)buf;
    }
    if (hf == 0) {
      const int nn = lane & 31;
      for (int k = 0; k < 16; ++k) buf[k] = have ? b[192 + k] : 0;
      *(v4u *)(rec + 8192 + (size_t)nn * 16) = *(const v4u *)buf;
      *(uint32_t *)(rec + 8704 + (size_t)nn * 4) = have ? ((uint32_t)b[208] | ((uint32_t)b[209] << 8)) : 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------ activations: f32 rows -> operand buffers
// One workgroup (512 threads) per token row: the decode engine's prologue (act_issue_all / act_finish_all: engine-order RmsNorm, candle's Q8_K quantizer)
// builds the row's image in LDS; the image is then written out as the GEMM's operands:
//   qf [S][T][256] f16: the int8 quants, inside every group of 8 in the operand's slot order;  yd [S][T] f32;  bsf [S][T][16] f16: the 16 run sums.
// GLU = true: the row is silu_engine(g) * u (the decode engine's gate / up epilogue expression), g = x, u = x2.
struct QuantArgs {
  const float *x, *x2; int ldx; const float *norm_w; float eps; int K, T;
  _Float16 *qf; float *yd; _Float16 *bsf;
  float *xtmp;  // GLU: f32 scratch [T][K] for the activated row (the prologue reads its input from memory)
};
template <bool GLU>
__global__ void __launch_bounds__(NT) qi_quantize_kernel(const QuantArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8];
  const int tid = tid_opaque(), t = blockIdx.x, K = a.K, S = K / 256;
  const float *xr = a.x + (size_t)t * a.ldx;
  if constexpr (GLU) {
    float *dst = a.xtmp + (size_t)t * K;
    const float *ur = a.x2 + (size_t)t * a.ldx;
    for (int e = tid * 4; e < K; e += NT * 4) {
      const float4 g = *(const float4 *)(xr + e), u = *(const float4 *)(ur + e);
      *(float4 *)(dst + e) = make_float4(silu_engine(g.x) * u.x, silu_engine(g.y) * u.y, silu_engine(g.z) * u.z, silu_engine(g.w) * u.w);
    }
    __syncthreads();  // the same threads read back what they wrote (tid * 4 + j * 2048), the barrier orders the other waves' view of nothing: cheap insurance
    xr = dst;
  }
  const ActRegs<2> pre = act_issue_all<2>(xr, a.norm_w, K);
  act_finish_all<1, 2>(smem, red, pre, xr, K, a.norm_w, a.eps, K, ACT_Q8K);
  __syncthreads();
  const Act act = act_view(smem, K, 1, ACT_Q8K);
  // thread -> (superblock, group of 8 elements): 32 groups per superblock
  for (int gi = tid; gi < S * 32; gi += NT) {
    const int sb = gi >> 5, g8 = gi & 31;
    const int8_t *q = (const int8_t *)(act.q + (size_t)sb * ACT_QS + g8 * 8);
    h8 o;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[jj] = (_Float16)(float)q[perm8(jj)];
    *(h8 *)(a.qf + ((size_t)sb * a.T + t) * 256 + g8 * 8) = o;
  }
  for (int i = tid; i < S * 16; i += NT) {
    const int sb = i >> 4, r = i & 15;
    a.bsf[((size_t)sb * a.T + t) * 16 + r] = (_Float16)(float)act.bs[(size_t)sb * ACT_BS + r];
  }
  for (int sb = tid; sb < S; sb += NT) a.yd[(size_t)sb * a.T + t] = act.d[sb];
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct GemmArgs {
  const uint8_t *w; unsigned w_bytes; int type, N, K, T;
  const _Float16 *qf; const float *yd; const _Float16 *bsf;
  float *out; int ldo; int accumulate;  // out[t * ldo + n] (+)= row sum
};
constexpr int TN = 128, TT = 128;                   // workgroup tile: weight rows x tokens
constexpr int LDS_ACT = TT * 512, LDS_YD = TT * 4, LDS_BS = TT * 48;
constexpr int LDS_TOTAL = LDS_ACT + LDS_YD + LDS_BS;

__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ unsigned as_u(h2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ h8 mk_h8(h2 a, h2 b, h2 c, h2 d) { h8 r; r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1]; return r; }
__device__ __forceinline__ h2 pkfma(h2 x, h2 s, h2 c) { return __builtin_elementwise_fma(x, s, c); }
// two bytes of a small unsigned integer -> (1024 + b0, 1024 + b1) as f16 (0x6400 = 1024: one unit per mantissa step up to 2047)
__device__ __forceinline__ h2 magic(unsigned v, unsigned mask) { return as_h2((v & mask) | 0x64006400u); }

template <int TYPE>
__global__ void __launch_bounds__(256) gemm_qi_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *act_s = smem;
  float *yd_s = (float *)(smem + LDS_ACT);
  char *bs_s = smem + LDS_ACT + LDS_YD;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * TN, t0 = blockIdx.y * TT;
  const int wn = wave & 1, wt = wave >> 1;  // the wave's 64-row / 64-token half of the tile
  const int S = a.K / 256, Cs = (S + 3) / 4;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, (short)0, (int)a.w_bytes, 0x00020000);
  constexpr int REC = rec_bytes_qi(TYPE), NPC = TYPE == T_Q4_K ? 4 : 8;
  struct WRegs { v4u q[NPC]; v4u hs; unsigned hd; };
  WRegs wr[2];
  auto load_w = [&](int sb) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned panel = (unsigned)((n0 + wn * 64 + nt * 32) >> 5);
      const unsigned rec = (panel * (unsigned)S + (unsigned)sb) * (unsigned)REC;  // panels past N: out of range -> zeros
#pragma unroll
      for (int c = 0; c < NPC; ++c) wr[nt].q[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + (unsigned)(c * 64 + lane) * 16u, 0, 0);
      wr[nt].hs = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + (unsigned)(NPC * 1024) + (unsigned)nn * 16u, 0, 0);
      wr[nt].hd = __builtin_amdgcn_raw_buffer_load_b32(rw, rec + (unsigned)(NPC * 1024 + 512) + (unsigned)nn * 4u, 0, 0);
    }
  };
  f16v run[2][2], pend[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 16; ++v) { run[nt][tt][v] = 0.f; pend[nt][tt][v] = 0.f; }
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qf, (short)0, (int)((size_t)S * a.T * 512), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.yd, (short)0, (int)((size_t)S * a.T * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)a.bsf, (short)0, (int)((size_t)S * a.T * 32), 0x00020000);
  const int tvalid = min(TT, a.T - t0);  // tokens of this tile that exist
  for (int sb = 0; sb < S; ++sb) {
    load_w(sb);
    __syncthreads();  // every wave has finished reading the previous superblock's tokens
    // stage the tile's tokens: 128 rows x 32 chunks of 16 B, chunk ch of token row tr at ch ^ (tr & 31)
    {
      const unsigned base = (unsigned)(((size_t)sb * a.T + t0) * 512);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = tid + i * 256, tr = ci >> 5, ch = ci & 31;
        v4u v = v4u{0u, 0u, 0u, 0u};
        if (tr < tvalid) v = __builtin_amdgcn_raw_buffer_load_b128(rq, base + (unsigned)ci * 16u, 0, 0);
        *(v4u *)(act_s + tr * 512 + ((ch ^ (tr & 31)) << 4)) = v;
      }
      if (tid < TT) yd_s[tid] = tid < tvalid ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, (unsigned)(((size_t)sb * a.T + t0 + tid) * 4), 0, 0)) : 0.f;
      {
        const int tr = tid >> 1, part = tid & 1;  // 128 rows x 2 halves of 16 B
        v4u v = v4u{0u, 0u, 0u, 0u};
        if (tr < tvalid) v = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(((size_t)sb * a.T + t0 + tr) * 32 + part * 16), 0, 0);
        *(v4u *)(bs_s + tr * 48 + part * 16) = v;
      }
    }
    __syncthreads();
    f16v X[2][2], X2[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { X[nt][tt][v] = 0.f; X2[nt][tt][v] = 0.f; }
    const int trow = wt * 64 + nn;  // + 32 tt: the token row whose fragment this lane supplies
    auto act_frag = [&](int tt, int run_i) -> h8 {
      const int tr = trow + 32 * tt, ch = 2 * run_i + hf;
      return *(const h8 *)(act_s + tr * 512 + ((ch ^ (tr & 31)) << 4));
    };
    if constexpr (TYPE == T_Q4_K) {
      // scales as f16 pairs.  (1024 + q) - 1032 = q - 8 and (q - 8) sc are both exact in f16; a single fma with the addend -1032 sc would not be:
      // 1032 sc is not an f16 value for odd sc >= 16
      h2 sc2[2][8];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned s01[2] = {wr[nt].hs.x, wr[nt].hs.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 sj = (_Float16)(float)byte_of(s01[j >> 2], j & 3);
          sc2[nt][j] = h2{sj, sj};
        }
      }
      const h2 m1032 = h2{(_Float16)(-1032.0f), (_Float16)(-1032.0f)};
      auto wop = [&](unsigned v, h2 sc) -> h2 { return (magic(v, 0x000F000Fu) + m1032) * sc; };
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          h8 wlo[2], whi[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const unsigned d0 = h == 0 ? wr[nt].q[c].x : wr[nt].q[c].z, d1 = h == 0 ? wr[nt].q[c].y : wr[nt].q[c].w;
            const h2 sl = sc2[nt][2 * c], sh = sc2[nt][2 * c + 1];
            wlo[nt] = mk_h8(wop(d0, sl), wop(d0 >> 8, sl), wop(d1, sl), wop(d1 >> 8, sl));
            whi[nt] = mk_h8(wop(d0 >> 4, sh), wop(d0 >> 12, sh), wop(d1 >> 4, sh), wop(d1 >> 12, sh));
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const h8 alo = act_frag(tt, 4 * c + h), ahi = act_frag(tt, 4 * c + 2 + h);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              X[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, wlo[nt], X[nt][tt], 0, 0, 0);
              X[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, whi[nt], X[nt][tt], 0, 0, 0);
            }
          }
        }
      }
      // Y = sum_run sc_{run / 2} bsum_run, M = sum_run m_{run / 2} bsum_run: operand slot (hf, jj) <-> run 8 hf + jj -> sub-block 4 hf + jj / 2
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned sw = hf ? wr[nt].hs.y : wr[nt].hs.x, mw = hf ? wr[nt].hs.w : wr[nt].hs.z;
        h8 ws, wm;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const _Float16 sk = (_Float16)(float)byte_of(sw, k), mk = (_Float16)(float)byte_of(mw, k);
          ws[2 * k] = sk; ws[2 * k + 1] = sk; wm[2 * k] = mk; wm[2 * k + 1] = mk;
        }
        const float d = half_bits_to_float((uint16_t)(wr[nt].hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(wr[nt].hd >> 16));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const h8 bsf = *(const h8 *)(bs_s + (trow + 32 * tt) * 48 + hf * 16);
          f16v zero;
#pragma unroll
          for (int v = 0; v < 16; ++v) zero[v] = 0.f;
          const f16v Y = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, ws, zero, 0, 0, 0);
          const f16v M = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, wm, zero, 0, 0, 0);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 y4 = *(const float4 *)(yd_s + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
            const float ydv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int v = 4 * q4 + k;
              const float If = fmaf(8.0f, Y[v], X[nt][tt][v]);  // (float)isum: one rounding of the exact integer X + 8 Y
              const float t = fmaf(d * ydv[k], If, -((dmin * ydv[k]) * M[v]));
              run[nt][tt][v] = (sb % Cs) == 0 ? t : run[nt][tt][v] + t;
            }
          }
        }
      }
    } else {  // Q6_K
#pragma unroll
      for (int g = 0; g < 8; ++g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // run 2 g + h
          h8 wl[2], wh[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const unsigned d0 = h == 0 ? wr[nt].q[g].x : wr[nt].q[g].z, d1 = h == 0 ? wr[nt].q[g].y : wr[nt].q[g].w;
            const int r = 2 * g + h;
            const unsigned scw = r < 4 ? wr[nt].hs.x : r < 8 ? wr[nt].hs.y : r < 12 ? wr[nt].hs.z : wr[nt].hs.w;
            const int sc = sbyte_of(scw, r & 3), sli = sc & 15, shi = sc >> 4;  // sc = 16 shi + sli
            const _Float16 slf = (_Float16)(float)sli, shf = (_Float16)(float)shi;
            const h2 sl = h2{slf, slf}, sh = h2{shf, shf};
            const h2 ol = h2{(_Float16)(-1056.0f), (_Float16)(-1056.0f)} * sl, oh = h2{(_Float16)(-1056.0f), (_Float16)(-1056.0f)} * sh;  // -(1024 + 32) s: exact
            const h2 p0 = magic(d0, 0x00FF00FFu), p1 = magic(d0 >> 8, 0x00FF00FFu), p2 = magic(d1, 0x00FF00FFu), p3 = magic(d1 >> 8, 0x00FF00FFu);
            wl[nt] = mk_h8(pkfma(p0, sl, ol), pkfma(p1, sl, ol), pkfma(p2, sl, ol), pkfma(p3, sl, ol));
            wh[nt] = mk_h8(pkfma(p0, sh, oh), pkfma(p1, sh, oh), pkfma(p2, sh, oh), pkfma(p3, sh, oh));
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const h8 af = act_frag(tt, 2 * g + h);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              X[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wl[nt], X[nt][tt], 0, 0, 0);
              X2[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wh[nt], X2[nt][tt], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float d = half_bits_to_float((uint16_t)(wr[nt].hd & 0xffff));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 y4 = *(const float4 *)(yd_s + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
            const float ydv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int v = 4 * q4 + k;
              const float If = fmaf(16.0f, X2[nt][tt][v], X[nt][tt][v]);  // (float)isum
              const float t = (d * ydv[k]) * If;
              run[nt][tt][v] = (sb % Cs) == 0 ? t : run[nt][tt][v] + t;
            }
          }
      }
    }
    if ((sb + 1) % Cs == 0 || sb + 1 == S) {  // a run of superblocks ends: its sum joins the row's left-to-right combination
      const bool first = sb < Cs;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int v = 0; v < 16; ++v) pend[nt][tt][v] = first ? run[nt][tt][v] : pend[nt][tt][v] + run[nt][tt][v];
    }
  }
  // runs that do not exist (S < 4 chunks) add +0: nothing to do.  Store: lane holds column n, rows t = 8 (v / 4) + 4 hf + v % 4 of each MFMA tile
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + wn * 64 + nt * 32 + nn;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int t = t0 + wt * 64 + tt * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
        if (n < a.N && t < a.T) {
          float *o = a.out + (size_t)t * a.ldo + n;
          *o = a.accumulate ? *o + pend[nt][tt][v] : pend[nt][tt][v];
        }
      }
  }
}

}  // namespace qi
}  // namespace mrs

using namespace mrs;

extern "C" size_t mrs_gemm_qi_repack_bytes(int type, long long n, long long k) {
  if (!qi::qi_type(type) || n <= 0 || k <= 0 || k % 256) return 0;
  return qi::qi_tensor_bytes(type, n, k);
}
// GGUF blocks [n][k / 256] (q4_k / q6_k) -> the MFMA-order copy mrs_gemm_qi reads
extern "C" int mrs_gemm_qi_repack(const void *gguf_blocks, int type, long long n, long long k, void *dst, void *stream) {
  if (!mrs_gemm_qi_repack_bytes(type, n, k) || !gguf_blocks || !dst) return -1;
  const long long total = ((n + 31) / 32) * (k / 256) * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (type == T_Q4_K) hipLaunchKernelGGL(qi::qi_repack_kernel<T_Q4_K>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gguf_blocks, (uint8_t *)dst, n, (int)k, total);
  else hipLaunchKernelGGL(qi::qi_repack_kernel<T_Q6_K>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gguf_blocks, (uint8_t *)dst, n, (int)k, total);
  return 0;
}
// bytes of the three operand buffers of T rows of K values: qf (f16 quants), yd (f32 block scales), bsf (f16 run sums), each 256-byte aligned inside ONE buffer
extern "C" size_t mrs_qi_act_bytes(int T, int K) {
  const size_t S = (size_t)(K / 256), t = (size_t)T;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  return al(S * t * 512) + al(S * t * 4) + al(S * t * 32);
}
static void qi_split(void *buf, int T, int K, _Float16 **qf, float **yd, _Float16 **bsf) {
  const size_t S = (size_t)(K / 256), t = (size_t)T;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char *p = (char *)buf;
  *qf = (_Float16 *)p; p += al(S * t * 512);
  *yd = (float *)p; p += al(S * t * 4);
  *bsf = (_Float16 *)p;
}
// x f32 [T][ldx] (-> RmsNorm with norm_w in the engine's order when norm_w != NULL) -> Q8_K per row -> operand buffers `act` (mrs_qi_act_bytes).
// x2 != NULL: the row is silu(x) * x2 (the gate / up epilogue of the decode engine), xtmp = f32 scratch [T][K].
extern "C" int mrs_qi_quantize(const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream) {
  if (!x || !act || T <= 0 || K <= 0 || K % 256 || (x2 && !xtmp)) return -1;
  qi::QuantArgs a{};
  a.x = x; a.x2 = x2; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.K = K; a.T = T; a.xtmp = xtmp;
  qi_split(act, T, K, &a.qf, &a.yd, &a.bsf);
  const size_t lds = (dec2::act_bytes(K, 1) + 15) & ~(size_t)15;
  if (lds > 158 * 1024) return -2;
  if (x2) { auto kern = qi::qi_quantize_kernel<true>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(T), dim3(dec2::NT), lds, (hipStream_t)stream, a); }
  else { auto kern = qi::qi_quantize_kernel<false>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(T), dim3(dec2::NT), lds, (hipStream_t)stream, a); }
  return 0;
}
// out[t * ldo + n] (+)= W[n] . act[t] for t < T, n < N in the decode engine's arithmetic and f32 order.  w_qi: mrs_gemm_qi_repack output; act: mrs_qi_quantize output.
extern "C" int mrs_gemm_qi(const void *w_qi, int type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *stream) {
  if (!w_qi || !act || !out || !qi::qi_type(type) || N <= 0 || T <= 0 || K <= 0 || K % 256) return -1;
  qi::GemmArgs a{};
  a.w = (const uint8_t *)w_qi; a.w_bytes = (unsigned)qi::qi_tensor_bytes(type, N, K); a.type = type; a.N = N; a.K = K; a.T = T;
  _Float16 *qf, *bsf; float *yd;
  qi_split((void *)act, T, K, &qf, &yd, &bsf);
  if ((size_t)(K / 256) * T * 512 >= 0x7fffffffull || qi::qi_tensor_bytes(type, N, K) >= 0xffffff00ull) return -2;
  a.qf = qf; a.yd = yd; a.bsf = bsf; a.out = out; a.ldo = ldo; a.accumulate = accumulate;
  const dim3 grid((N + qi::TN - 1) / qi::TN, (T + qi::TT - 1) / qi::TT);
  if (type == T_Q4_K) { auto kern = qi::gemm_qi_kernel<T_Q4_K>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(256), qi::LDS_TOTAL, (hipStream_t)stream, a); }
  else { auto kern = qi::gemm_qi_kernel<T_Q6_K>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(256), qi::LDS_TOTAL, (hipStream_t)stream, a); }
  return 0;
}

// ------------------------------------------------------------------------------------------------ prompt attention in the decode engine's arithmetic
// Query token t of the prompt sees exactly what a decode step at its position sees: the per-32-token-block partials of attn_split_core (dec_attn.cuh: f32
// online softmax through the reference's fast_exp, split length bpw from the model's maximum context like mrs_dec_attention) merged in the order of
// single_q.rs run_barrier (attn_merge_core).  One workgroup = one (token, kv head): wave w takes splits w, w + 4, ..., the partials (<= 64 per head) stay in
// LDS, waves then merge one query head each.  The result equals mrs_dec_attention at that position bit for bit (tests/test_prefill_exact.py), so the
// o_proj input, every later layer and the KV pages of a prompt are the ones a token-by-token decode would have produced.
#include "dec_attn.cuh"
namespace mrs {
namespace qi {
template <int G, class CT>
__global__ void __launch_bounds__(256) prefill_attn_exact_kernel(const dec::AttnArgs a, const int start_pos) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 128;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kvh = blockIdx.x, t = blockIdx.y;
  const int ms = a.max_splits;
  float *po = (float *)smem;                       // [G][ms][128]
  float *pm = po + (size_t)G * ms * HD, *pl = pm + G * ms;  // [G][ms]
  float *q_s = pl + G * ms + wave * (G * HD + G * 32), *p_s = q_s + G * HD;
  const int ctx = (int)a.context_lens[t];
  const int nblk = (ctx + 31) / 32, ns = (nblk + a.bpw - 1) / a.bpw;
  const int lo_w = a.window > 0 && ctx > a.window ? ctx - a.window : 0;
  for (int sp = wave; sp < ns; sp += 4) {
    const int b0 = sp * a.bpw, b1 = min(b0 + a.bpw, nblk);
    auto keep = [&](int g, float o0, float o1, float m, float l) {
      float *o = po + ((size_t)g * ms + sp) * HD;
      o[lane] = o0; o[lane + 64] = o1;
      if (lane == 0) { pm[g * ms + sp] = m; pl[g * ms + sp] = l; }
    };
    if (b1 * 32 <= lo_w) {  // the split lies before the sliding window: what a fully masked pass gives (mrs_dec_attention publishes the same)
#pragma unroll
      for (int g = 0; g < G; ++g) keep(g, 0.f, 0.f, -FLT_MAX, 0.f);
    } else {
      dec::attn_split_core<G, CT>(a, kvh, kvh * G, t, b0, b1, q_s, p_s, keep);
    }
  }
  __syncthreads();
  for (int g = wave; g < G; g += 4) {
    float v0, v1;
    dec::attn_merge_core(ns, pm + g * ms, pl + g * ms, po + (size_t)g * ms * HD, v0, v1);
    float *o = a.out + ((size_t)t * a.num_heads + kvh * G + g) * HD;
    o[lane] = v0; o[lane + 64] = v1;
  }
}
}  // namespace qi
}  // namespace mrs

// q f32 [T][q_stride] (RoPE applied), pages already hold the prompt's K / V; context_lens [T] = position + 1 of every prompt token (device), block_table = the
// sequence's row; out f32 [T][num_heads * 128].  max_context_len = the model's (it fixes the split length exactly as in mrs_dec_attention).
extern "C" int mrs_prefill_attention_exact(const float *q, const void *k_cache, const void *v_cache, const uint32_t *block_table, const uint32_t *context_lens, float *out,
                                           int T, int num_heads, int num_kv_heads, int head_size, int block_size, int q_stride, int kv_block_stride, int kv_head_stride,
                                           float scale, int max_context_len, int kv_dtype, int sliding_window, void *stream) {
  if (!q || !out || head_size != 128 || block_size != 32 || T <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads || (kv_dtype != 0 && kv_dtype != 1)) return -1;
  const int G = num_heads / num_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8) return -1;
  mrs::dec::AttnArgs a{};
  a.q = q; a.k_cache = (const uint16_t *)k_cache; a.v_cache = (const uint16_t *)v_cache; a.block_tables = block_table; a.context_lens = context_lens;
  a.out = out; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads; a.max_blocks_per_seq = 0 /* every token reads the same row */; a.q_stride = q_stride;
  a.kv_block_stride = kv_block_stride; a.kv_head_stride = kv_head_stride; a.num_seqs = T; a.scale = scale; a.window = sliding_window > 0 ? sliding_window : 0;
  const int nblk = (max_context_len + 31) / 32;
  a.bpw = nblk <= 64 ? 1 : (nblk + 63) / 64;  // == mrs_dec_attention
  a.max_splits = 64;
  const size_t lds = ((size_t)G * 64 * 128 + 2 * (size_t)G * 64 + 4 * ((size_t)G * 128 + (size_t)G * 32)) * 4;
  if (lds > 158 * 1024) return -2;
  const dim3 grid(num_kv_heads, T);
  hipStream_t s = (hipStream_t)stream;
#define MRS_PA(GG, CT) { auto kern = mrs::qi::prefill_attn_exact_kernel<GG, CT>; mrs::lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, 0); }
#define MRS_PAG(CT) switch (G) { case 1: MRS_PA(1, CT) break; case 2: MRS_PA(2, CT) break; case 4: MRS_PA(4, CT) break; default: MRS_PA(8, CT) break; }
  if (kv_dtype == 1) { MRS_PAG(mrs::bf16_t) } else { MRS_PAG(mrs::f16_t) }
#undef MRS_PAG
#undef MRS_PA
  return 0;
}
