// ext_gemm_qi.hip -- prompt GEMM in the arithmetic of the reference CPU path, on the MI355X matrix cores, bit-identical to the decode engine.
//
// What the reference does on a CPU device for a T-token prompt: every linear goes through the same QMatMul f32 fallback as decode
// (mistralrs-quant/src/gguf/mod.rs:465-478): each activation ROW is quantized to Q8_K (K-quants), every output is a sum over the row's superblocks of
// (d_w d_x) <integer dot> - (dmin_w d_x) <integer min term>.  Rounds 1-3 ran prompts in a different arithmetic (weights and activations rounded to bf16);
// this kernel computes the reference's integers EXACTLY on v_mfma_f32_32x32x16_f16 and combines them in the f32 order of the decode engine ("ORD-U",
// dec_core2.cuh / oracle orc_gemv_engine), so a token's logits and KV pages do not depend on whether it was part of a prompt or decoded:
//   * the 6-bit sub-block scale is folded INTO the weight operand as an exact small integer in f16: Q4_K  sc q <= 945;  Q6_K  two operands
//     sl (q - 32) and sh (q - 32) with sc = 16 sh + sl, |.| <= 480 / 256.  The activation operand is the Q8_K quant as f16.  Every product and every partial
//     sum of a superblock is an integer below 2^24, so the f32 accumulator of the matrix core holds the EXACT sum (tests/test_gemm_qi.py holds
//     v_mfma_f32_32x32x16_f16 to that on adversarial operands); no per-sub-block scale multiply is left on the vector ALU;
//   * Q4_K: the operand is sc q (<= 945); the first and the second eight runs of a superblock go to two accumulators (<= 15.5 M each), (float)isum = XA + XB is
//     one rounding of the exact integer -- what (float)int gives; msum = sum_j m_j bsum_j is one more MFMA per superblock (K = the 16 run sums of the Q8_K
//     block, mins duplicated per run);
//   * Q5_K (round 5): the operand is sc (q - 16) (|.| <= 1008: one packed fma for sc q, one packed add for - 16 sc), two accumulators like Q4_K (<= 16.5 M each), and the
//     16 sum_j sc_j bsum_j the offset leaves out is one more MFMA (S); isum = XA + XB + 16 S is formed in int32 (a superblock's sum reaches 64 M: two f32 additions
//     would round twice) and converted once; the mins' M as for Q4_K;
//   * per superblock and output ONE f32 term T (the decode engine's expression), terms added left to right inside each of the row's four runs of
//     superblocks, the four run sums left to right (dec_core2.cuh header).
// Operands: weights in an MFMA-order copy made at load time (mrs_gemm_qi_repack: per 32-row panel and superblock, every lane's 16 bytes of a piece
// contiguous: one buffer_load_dwordx4 per lane and piece, straight into registers); activations from mrs_qi_quantize (engine-order RmsNorm + the candle
// Q8_K quantizer of dec_core2.cuh, then f16 in the operand's k order, slab-major [superblock][token][256]).
// Tiling: workgroup 512 threads = 8 waves (two per SIMD: one wave's MFMAs run under the other's operand build and fix-up), tile 128 weight rows x 128
// tokens, wave 32 x 64 = 1 x 2 MFMA tiles; per superblock the 128 tokens' f16 quants (64 KiB) are staged in LDS (16-byte chunks XOR-swizzled by the token
// index: conflict-free ds_read_b128 fragments), double buffered: superblock s + 1 arrives by LDS-DMA (global_load_lds) while s is multiplied.
#include "dec_core2.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace mrs {
namespace qi {
using namespace mrs::dec2;

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int REC_Q4K = 4096 + 512 + 128, REC_Q6K = 8192 + 512 + 128, REC_Q5K = REC_Q6K;  // Q5_K: the 5-bit values as bytes, like Q6_K's 6-bit ones
__host__ __device__ constexpr int rec_bytes_qi(int type) { return type == T_Q4_K ? REC_Q4K : REC_Q6K; }
// Q8_0 (round 6): a "superblock" = 8 consecutive blocks of 32; the record holds the int8 quants as they are (8 pieces) + the 8 f16 block scales per row
__host__ __device__ inline bool qi_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }
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
      for (int k = 0; k < 8; ++k) { buf[k] = have ? qs[32 * c + 8 * hf + k] : 0; buf[8 + k] = have ? qs[32 * c + 16 + 8 * hf + k] : 0; }
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
  } else if constexpr (TYPE == T_Q5_K) {  // block_q5_K (176 B): d, dmin, scales[12], qh[32], qs[128]; the 5-bit values as bytes (0 .. 31) in Q6_K's piece order
    const uint8_t *b = src + ((size_t)row * S + sb) * 176, *qh = b + 16, *qs = b + 48;
    // dequantize_row_q5_K: 64 values per j: low nibbles of qs[32 j ..] with bit 2 j of qh, then the high nibbles with bit 2 j + 1
    auto q5 = [&](int e) { const int j = e >> 6, w = e & 63, l = w & 31, up = w >> 5;
                           const int lo = up ? (qs[32 * j + l] >> 4) : (qs[32 * j + l] & 15), hi = (qh[l] >> (2 * j + up)) & 1; return lo | (hi << 4); };
    for (int g = 0; g < 8; ++g) {  // piece g: runs 2g, 2g + 1 (= sub-block g); the lane's 8 elements 8 hf .. 8 hf + 7 of each
      for (int k = 0; k < 8; ++k) { buf[k] = have ? (uint8_t)q5((2 * g) * 16 + 8 * hf + k) : 0; buf[8 + k] = have ? (uint8_t)q5((2 * g + 1) * 16 + 8 * hf + k) : 0; }
      *(v4u *)(rec + ((size_t)g * 64 + lane) * 16) = *(const v4u *)buf;
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
      *(v4u *)(rec + 8192 + (size_t)nn * 16) = *(const v4u *)buf;
      *(uint32_t *)(rec + 8704 + (size_t)nn * 4) = have ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) : 0u;
    }
  } else if constexpr (TYPE == T_Q8_0) {  // 8 x block_q8_0 (34 B: f16 d, 32 int8): piece g = block g, the lane's bytes 16 hf .. 16 hf + 15 (the i8 MFMA's k order)
    const uint8_t *b = src + ((size_t)row * S + sb) * 272;
    for (int g = 0; g < 8; ++g) {
      for (int k = 0; k < 16; ++k) buf[k] = have ? b[34 * g + 2 + 16 * hf + k] : 0;
      *(v4u *)(rec + ((size_t)g * 64 + lane) * 16) = *(const v4u *)buf;
    }
    if (hf == 0) {
      const int nn = lane & 31;
      for (int g = 0; g < 8; ++g) { buf[2 * g] = have ? b[34 * g] : 0; buf[2 * g + 1] = have ? b[34 * g + 1] : 0; }
      *(v4u *)(rec + 8192 + (size_t)nn * 16) = *(const v4u *)buf;
      *(uint32_t *)(rec + 8704 + (size_t)nn * 4) = 0u;
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
  int mode;     // ACT_Q8K (K-quant weights) / ACT_Q80 (Q8_0 weights: quantize_row_q8_0 per 32 values; outputs q8 [S][T][256] int8 in element order at `qf`, the block
                // scales f32(f16(d)) block-major [S][8][T] at `bsf`)
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
  const int qw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ActRegs<1> pre = act_issue_all<1>(xr, (unsigned)K * 4u, a.norm_w, K, qw);
  act_finish_all<1>(smem, red, pre, xr, K, a.norm_w, a.eps, K, a.mode, qw);
  __syncthreads();
  const Act act = act_view(smem, K, 1, a.mode);
  if (a.mode == ACT_Q80) {
    int8_t *q8 = (int8_t *)a.qf;
    float *yd8 = (float *)a.bsf;
    for (int gi = tid; gi < S * 16; gi += NT) {  // 16-byte chunks of the superblock's 256 quants
      const int sb = gi >> 4, ch = gi & 15;
      *(v4u *)(q8 + ((size_t)sb * a.T + t) * 256 + ch * 16) = *(const v4u *)(act.q + act.qoff(sb) + ch * 16);
    }
    for (int i = tid; i < S * 8; i += NT) {
      const int sb = i >> 3, b = i & 7;
      yd8[((size_t)sb * 8 + b) * a.T + t] = act.d[(size_t)sb * act.dw + b];
    }
    return;
  }
  // thread -> (superblock, group of 8 elements): 32 groups per superblock
  for (int gi = tid; gi < S * 32; gi += NT) {
    const int sb = gi >> 5, g8 = gi & 31;
    const int8_t *q = (const int8_t *)(act.q + act.qoff(sb) + g8 * 8);
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
  float *part; int ksplit;              // ksplit = 4: workgroup z multiplies run z of the row's superblocks only and writes its sum to part[z][t][n]; the reduce kernel combines
  const int *win;                       // device {begin, end}: only the operand rows [begin, end) take part (one expert's routes of a grouped MoE prompt GEMM: the bounds
                                        // come from launch_moe_dispatch on the device, the grid is sized for the worst case and tiles past `end` leave); nullptr = all T rows
};
constexpr int TN = 128, TT = 128;                   // workgroup tile: weight rows x tokens
constexpr int GT = 512;                             // 8 waves: 4 (32-row panels) x 2 (64-token halves); two waves per SIMD overlap each other's MFMA and VALU phases
constexpr int LDS_ACT = TT * 512, LDS_YD = TT * 4, LDS_BS = TT * 32;
constexpr int LDS_BUF = LDS_ACT + LDS_YD + LDS_BS;  // one superblock of the tile's tokens; two buffers: the next superblock arrives by LDS-DMA during the MFMAs
constexpr int LDS_TOTAL = 2 * LDS_BUF;

// async global -> LDS copy, 16 / 4 bytes per lane: LDS destination = wave-uniform base + lane * size (the hardware's rule), source address per lane
#ifndef MRS_GLDS16
#define MRS_GLDS16(gptr, lbase) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lbase), 16, 0, 0)
#define MRS_GLDS4(gptr, lbase) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lbase), 4, 0, 0)
#endif
#ifndef MRS_WAIT_VMCNT0
#define MRS_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ h8 mk_h8(h2 a, h2 b, h2 c, h2 d) { h8 r; r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1]; return r; }
__device__ __forceinline__ h2 pkfma(h2 x, h2 s, h2 c) { return __builtin_elementwise_fma(x, s, c); }
// two bytes of a small unsigned integer -> (1024 + b0, 1024 + b1) as f16 (0x6400 = 1024: one unit per mantissa step up to 2047)
__device__ __forceinline__ h2 magic(unsigned v, unsigned mask) { return as_h2((v & mask) | 0x64006400u); }

template <int TYPE>
__global__ void __launch_bounds__(GT) gemm_qi_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * TN;
  const int tbeg = a.win ? __builtin_amdgcn_readfirstlane(a.win[0]) : 0, tend = a.win ? __builtin_amdgcn_readfirstlane(a.win[1]) : a.T;
  const int t0 = tbeg + blockIdx.y * TT;
  if (t0 >= tend) return;  // workgroup-uniform
  const int wn = wave & 3, wt = wave >> 2;  // the wave's 32-row panel / 64-token half of the tile
  const int S = a.K / 256, Cs = (S + 3) / 4;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, (short)0, (int)a.w_bytes, 0x00020000);
  constexpr int REC = rec_bytes_qi(TYPE), NPC = TYPE == T_Q4_K ? 4 : 8;
  struct WRegs { v4u q[NPC]; v4u hs; unsigned hd; };
  const unsigned panel = (unsigned)((n0 + wn * 32) >> 5);
  auto load_w = [&](WRegs &wr, int sb) {
    const unsigned rec = (panel * (unsigned)S + (unsigned)sb) * (unsigned)REC;  // panels past N: out of range -> zeros
#pragma unroll
    for (int c = 0; c < NPC; ++c) wr.q[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + (unsigned)(c * 64 + lane) * 16u, 0, 0);
    wr.hs = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + (unsigned)(NPC * 1024) + (unsigned)nn * 16u, 0, 0);
    wr.hd = __builtin_amdgcn_raw_buffer_load_b32(rw, rec + (unsigned)(NPC * 1024 + 512) + (unsigned)nn * 4u, 0, 0);
  };
  // stage superblock sb of the tile's tokens into buffer `buf` by LDS-DMA.  acts: 128 rows x 32 chunks of 16 B, chunk ch of row tr at position ch ^ (tr & 31);
  // a DMA instruction fills 1 KiB = 2 rows lane by lane, so lane l of the instruction for rows (2 i, 2 i + 1) fetches chunk (l & 31) ^ (tr & 31) of row
  // tr = 2 i + (l >> 5).  Rows past T re-read the last token (their outputs are never stored).
  const int tlast = tend - 1 - t0;  // last existing row of the tile
  auto stage = [&](int sb, int buf) {
    char *base = smem + buf * LDS_BUF;
    const char *gq = (const char *)a.qf + ((size_t)sb * a.T + t0) * 512;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pair = wave * 8 + i;  // rows 2 pair, 2 pair + 1
      const int tr = 2 * pair + hf, trc = min(tr, tlast);
      MRS_GLDS16(gq + (size_t)trc * 512 + (((lane & 31) ^ (tr & 31)) << 4), base + pair * 1024);
    }
    if (wave < 2) {  // yd: 128 floats = 2 instructions of 64 x 4 B
      const int tr = wave * 64 + lane;
      MRS_GLDS4((const char *)a.yd + ((size_t)sb * a.T + t0 + min(tr, tlast)) * 4, base + LDS_ACT + wave * 256);
    } else if (wave < 6) {  // run sums: 128 rows x 32 B = 4 instructions of 1 KiB
      const int i = wave - 2, tr = i * 32 + (lane >> 1);
      MRS_GLDS16((const char *)a.bsf + ((size_t)sb * a.T + t0 + min(tr, tlast)) * 32 + (lane & 1) * 16, base + LDS_ACT + LDS_YD + i * 1024);
    }
  };
  f16v run[2], pend[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 16; ++v) { run[tt][v] = 0.f; pend[tt][v] = 0.f; }
  WRegs wr, wnx;
  const int sb_first = a.ksplit > 1 ? (int)blockIdx.z * ((a.K / 256 + 3) / 4) : 0;
  if (sb_first < a.K / 256) { load_w(wr, sb_first); stage(sb_first, sb_first & 1); }
  const int trow = wt * 64 + nn;  // + 32 tt: the token row whose fragment this lane supplies
  // fragment of run r for this lane: chunk (2 r + hf) ^ (row & 31) of its token row; row & 31 == nn for both token tiles, so the 16 offsets are computed once
  int aoff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) aoff[r] = trow * 512 + (((2 * r + hf) ^ nn) << 4);
  const int sb_begin = a.ksplit > 1 ? (int)blockIdx.z * Cs : 0, sb_end = a.ksplit > 1 ? min(S, sb_begin + Cs) : S;
  if (sb_begin >= sb_end) {
    if (a.ksplit > 1) {  // a run without superblocks contributes +0
      const int n = n0 + wn * 32 + nn;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int t = t0 + wt * 64 + tt * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
          if (n < a.N && t < tend) a.part[((size_t)blockIdx.z * a.T + t) * a.N + n] = 0.f;
        }
    }
    return;
  }
  for (int sb = sb_begin; sb < sb_end; ++sb) {
    const int buf = sb & 1;
    MRS_WAIT_VMCNT0();  // this wave's DMA pieces of superblock sb (and the weights of sb) have landed
    __syncthreads();    // everyone's have, and every wave has finished reading the other buffer
    if (sb + 1 < sb_end) { stage(sb + 1, buf ^ 1); load_w(wnx, sb + 1); }
    const char *act_s = smem + buf * LDS_BUF;
    const float *yd_s = (const float *)(act_s + LDS_ACT);
    const char *bs_s = act_s + LDS_ACT + LDS_YD;
    f16v X[2], X2[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 16; ++v) { X[tt][v] = 0.f; X2[tt][v] = 0.f; }
    // Q5_K: the 16 fragment offsets are recomputed per superblock from an opaque copy of the lane's row (kept as loop invariants -- aoff[] -- they are what tips this
    // branch over 256 registers: 6 spilled dwords)
    constexpr bool RECOMP = TYPE == T_Q5_K;
    int nnv = nn;
    if constexpr (RECOMP) MRS_OPAQUE_TID(nnv);
    auto act_frag = [&](int tt, int run_i) -> h8 {
      if constexpr (RECOMP) return *(const h8 *)(act_s + trow * 512 + (((2 * run_i + hf) ^ nnv) << 4) + tt * (32 * 512));
      else return *(const h8 *)(act_s + aoff[run_i] + tt * (32 * 512));
    };
    const bool cfirst = (sb % Cs) == 0;
    const f2 keep2 = cfirst ? f2{0.f, 0.f} : f2{1.f, 1.f};  // the first term of a run of superblocks starts its sum (0 * run + t), the others add (1 * run + t: exact)
    if constexpr (TYPE == T_Q4_K) {
      // operand = sc q as an exact integer in f16 from ONE packed fma per pair of weights: (1024 + q) sc - 1024 sc for a nibble at bits 3:0 of a half-word,
      // (1024 + 16 q) (sc / 16) - 64 sc for one at bits 7:4 (1024 sc, 64 sc and sc / 16 are exact in f16; 1032 sc, which q - 8 would need, is not).  With the
      // unsigned q a superblock's sum can reach 30.9 M > 2^24, so runs 0 .. 7 and 8 .. 15 go to two accumulators (<= 15.5 M each, exact) and
      // (float)isum = XA + XB is one rounding of the exact integer.
      unsigned mg = 0x64006400u;
      MRS_OPAQUE_TID(mg);  // keeps the constant in a VGPR: v_and_or_b32 takes one literal
      auto lo4 = [&](unsigned v, h2 sc, h2 off) -> h2 { return pkfma(as_h2((v & 0x000F000Fu) | mg), sc, off); };
      auto hi4 = [&](unsigned v, h2 sc16, h2 off) -> h2 { return pkfma(as_h2((v & 0x00F000F0u) | mg), sc16, off); };
      const unsigned s01[2] = {wr.hs.x, wr.hs.y};
      // Activation fragments are read ONE GROUP AHEAD (round 6): the code used to ask for a group's four fragments and wait for them (s_waitcnt lgkmcnt(0)) right in front
      // of its MFMAs -- 8 exposed LDS round trips per superblock and wave with two waves per SIMD to hide them (MFMA pipes 25 % busy, profiles/round4_pmc.md).  Group (c, h)
      // = runs 4 c + h (low nibbles) and 4 c + 2 + h (high) for both token tiles; scheduling barriers keep the read of group g + 1 in front of the operand build of group g.
      h8 fr[2][4];
      auto read_group = [&](int gi, h8 (&f)[4]) {
        const int c = gi >> 1, h = gi & 1;
        f[0] = act_frag(0, 4 * c + h); f[1] = act_frag(0, 4 * c + 2 + h); f[2] = act_frag(1, 4 * c + h); f[3] = act_frag(1, 4 * c + 2 + h);
      };
      read_group(0, fr[0]);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        const int c = gi >> 1, h = gi & 1;
        if (gi + 1 < 8) read_group(gi + 1, fr[(gi + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const float sa = (float)byte_of(s01[c >> 1], 2 * (c & 1)), sb2 = (float)byte_of(s01[c >> 1], 2 * (c & 1) + 1);
        const h2 sl = h2{(_Float16)sa, (_Float16)sa}, ol = h2{(_Float16)(-1024.0f * sa), (_Float16)(-1024.0f * sa)};
        const h2 sh = h2{(_Float16)(sb2 * 0.0625f), (_Float16)(sb2 * 0.0625f)}, oh = h2{(_Float16)(-64.0f * sb2), (_Float16)(-64.0f * sb2)};
        const unsigned d0 = h == 0 ? wr.q[c].x : wr.q[c].z, d1 = h == 0 ? wr.q[c].y : wr.q[c].w;
        const unsigned e0 = d0 >> 8, e1 = d1 >> 8;
        const h8 wlo = mk_h8(lo4(d0, sl, ol), lo4(e0, sl, ol), lo4(d1, sl, ol), lo4(e1, sl, ol));
        const h8 whi = mk_h8(hi4(d0, sh, oh), hi4(e0, sh, oh), hi4(d1, sh, oh), hi4(e1, sh, oh));
        const h8 (&f)[4] = fr[gi & 1];
        if (c < 2) {
          X[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], wlo, X[0], 0, 0, 0);
          X[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], wlo, X[1], 0, 0, 0);
          X[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], whi, X[0], 0, 0, 0);
          X[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], whi, X[1], 0, 0, 0);
        } else {
          X2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], wlo, X2[0], 0, 0, 0);
          X2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], wlo, X2[1], 0, 0, 0);
          X2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], whi, X2[0], 0, 0, 0);
          X2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], whi, X2[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // M = sum_run m_{run / 2} bsum_run: operand slot (hf, jj) <-> run 8 hf + jj -> sub-block 4 hf + jj / 2
      const unsigned mw = hf ? wr.hs.w : wr.hs.z;
      h8 wm;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const _Float16 mk = (_Float16)(float)byte_of(mw, k); wm[2 * k] = mk; wm[2 * k + 1] = mk; }
      const float d = half_bits_to_float((uint16_t)(wr.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(wr.hd >> 16));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const h8 bsf = *(const h8 *)(bs_s + (trow + 32 * tt) * 32 + hf * 16);
        f16v zero;
#pragma unroll
        for (int v = 0; v < 16; ++v) zero[v] = 0.f;
        const f16v M = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, wm, zero, 0, 0, 0);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {  // two outputs per instruction (v_pk_*_f32: the same IEEE operations per lane)
          const float4 y4 = *(const float4 *)(yd_s + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int v = 4 * q4 + k;
            const f2 yd2 = k == 0 ? f2{y4.x, y4.y} : f2{y4.z, y4.w};
            const f2 If = f2{X[tt][v], X[tt][v + 1]} + f2{X2[tt][v], X2[tt][v + 1]};  // (float)isum: one rounding of the exact integer
            const f2 mm = (f2{dmin, dmin} * yd2) * f2{M[v], M[v + 1]};
            const f2 t = __builtin_elementwise_fma(f2{d, d} * yd2, If, -mm);
            const f2 r = __builtin_elementwise_fma(f2{run[tt][v], run[tt][v + 1]}, keep2, t);  // run * 1 + t, or (first superblock of a run) run * 0 + t
            run[tt][v] = r[0]; run[tt][v + 1] = r[1];
          }
        }
      }
    } else if constexpr (TYPE == T_Q5_K) {
      // operand = sc (q - 16), |.| <= 1008: (1024 + q) sc - 1024 sc = sc q exactly (one packed fma; 1024 sc <= 64512 is an f16 value), then - 16 sc (1040 sc is not an
      // f16 value for sc = 63, so the offset takes a second packed add).  Runs 0 .. 7 and 8 .. 15 go to two accumulators (<= 128 x 1008 x 128 = 16.5 M < 2^24 each,
      // exact); the 16 sc_j bsum terms the offset leaves out are one more MFMA (S, like the mins' M); isum = XA + XB + 16 S is added as int32 and converted once --
      // what the CPU path's (float)int is.
      unsigned mg = 0x64006400u;
      MRS_OPAQUE_TID(mg);
      auto b2 = [&](unsigned v) -> h2 { return as_h2((v & 0x00FF00FFu) | mg); };  // (1024 + u0, 1024 + u1)
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float scf = (float)byte_of(g < 4 ? wr.hs.x : wr.hs.y, g & 3);
        const h2 sc = h2{(_Float16)scf, (_Float16)scf}, off = h2{(_Float16)(-1024.0f * scf), (_Float16)(-1024.0f * scf)}, o16 = h2{(_Float16)(-16.0f * scf), (_Float16)(-16.0f * scf)};
        auto opnd = [&](unsigned v) -> h2 { return pkfma(b2(v), sc, off) + o16; };
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // run 2 g + h
          const unsigned d0 = h == 0 ? wr.q[g].x : wr.q[g].z, d1 = h == 0 ? wr.q[g].y : wr.q[g].w;
          const h8 w = mk_h8(opnd(d0), opnd(d0 >> 8), opnd(d1), opnd(d1 >> 8));
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const h8 af = act_frag(tt, 2 * g + h);
            if (g < 4) X[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, w, X[tt], 0, 0, 0);
            else X2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, w, X2[tt], 0, 0, 0);
          }
        }
      }
      // M = sum_run m_{run / 2} bsum_run, S = sum_run sc_{run / 2} bsum_run: operand slot (hf, jj) <-> run 8 hf + jj -> sub-block 4 hf + jj / 2
      const unsigned mw = hf ? wr.hs.w : wr.hs.z, sw = hf ? wr.hs.y : wr.hs.x;
      h8 wm, ws;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const _Float16 mk = (_Float16)(float)byte_of(mw, k), sk = (_Float16)(float)byte_of(sw, k);
        wm[2 * k] = mk; wm[2 * k + 1] = mk; ws[2 * k] = sk; ws[2 * k + 1] = sk;
      }
      const float d = half_bits_to_float((uint16_t)(wr.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(wr.hd >> 16));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const h8 bsf = *(const h8 *)(bs_s + (trow + 32 * tt) * 32 + hf * 16);
        f16v zero;
#pragma unroll
        for (int v = 0; v < 16; ++v) zero[v] = 0.f;
        const f16v M = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, wm, zero, 0, 0, 0);
        const f16v Sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, ws, zero, 0, 0, 0);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 y4 = *(const float4 *)(yd_s + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int v = 4 * q4 + k;
            const f2 yd2 = k == 0 ? f2{y4.x, y4.y} : f2{y4.z, y4.w};
            const int i0 = ((int)X[tt][v] + (int)X2[tt][v]) + 16 * (int)Sx[v], i1 = ((int)X[tt][v + 1] + (int)X2[tt][v + 1]) + 16 * (int)Sx[v + 1];
            const f2 If = f2{(float)i0, (float)i1};  // (float)isum
            const f2 mm = (f2{dmin, dmin} * yd2) * f2{M[v], M[v + 1]};
            const f2 t = __builtin_elementwise_fma(f2{d, d} * yd2, If, -mm);
            const f2 r = __builtin_elementwise_fma(f2{run[tt][v], run[tt][v + 1]}, keep2, t);
            run[tt][v] = r[0]; run[tt][v + 1] = r[1];
          }
        }
      }
    } else {  // Q6_K
      unsigned mg = 0x64006400u;
      MRS_OPAQUE_TID(mg);
      auto b2 = [&](unsigned v) -> h2 { return as_h2((v & 0x00FF00FFu) | mg); };  // (1024 + u0, 1024 + u1)
#pragma unroll
      for (int g = 0; g < 8; ++g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // run 2 g + h
          const unsigned d0 = h == 0 ? wr.q[g].x : wr.q[g].z, d1 = h == 0 ? wr.q[g].y : wr.q[g].w;
          const int r = 2 * g + h;
          const unsigned scw = r < 4 ? wr.hs.x : r < 8 ? wr.hs.y : r < 12 ? wr.hs.z : wr.hs.w;
          const int sc = sbyte_of(scw, r & 3), sli = sc & 15, shi = sc >> 4;  // sc = 16 shi + sli
          const float slf = (float)sli, shf = (float)shi;
          // (1024 + u) s - 1056 s = (u - 32) s = (q - 32) s in one fma: 1056 s is an f16 value for |s| <= 15 (a multiple of 8 below 16384)
          const h2 sl = h2{(_Float16)slf, (_Float16)slf}, sh = h2{(_Float16)shf, (_Float16)shf};
          const h2 ol = h2{(_Float16)(-1056.0f * slf), (_Float16)(-1056.0f * slf)}, oh = h2{(_Float16)(-1056.0f * shf), (_Float16)(-1056.0f * shf)};
          const h2 p0 = b2(d0), p1 = b2(d0 >> 8), p2 = b2(d1), p3 = b2(d1 >> 8);
          const h8 wl = mk_h8(pkfma(p0, sl, ol), pkfma(p1, sl, ol), pkfma(p2, sl, ol), pkfma(p3, sl, ol));
          const h8 wh = mk_h8(pkfma(p0, sh, oh), pkfma(p1, sh, oh), pkfma(p2, sh, oh), pkfma(p3, sh, oh));
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const h8 af = act_frag(tt, r);
            X[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wl, X[tt], 0, 0, 0);
            X2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wh, X2[tt], 0, 0, 0);
          }
        }
      }
      const float d = half_bits_to_float((uint16_t)(wr.hd & 0xffff));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 y4 = *(const float4 *)(yd_s + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int v = 4 * q4 + k;
            const f2 yd2 = k == 0 ? f2{y4.x, y4.y} : f2{y4.z, y4.w};
            const f2 If = __builtin_elementwise_fma(f2{16.0f, 16.0f}, f2{X2[tt][v], X2[tt][v + 1]}, f2{X[tt][v], X[tt][v + 1]});  // (float)isum
            const f2 t = (f2{d, d} * yd2) * If;
            const f2 r = __builtin_elementwise_fma(f2{run[tt][v], run[tt][v + 1]}, keep2, t);
            run[tt][v] = r[0]; run[tt][v + 1] = r[1];
          }
        }
    }
    if ((sb + 1) % Cs == 0 || sb + 1 == sb_end) {  // a run of superblocks ends: its sum joins the row's left-to-right combination
      const bool first = sb < Cs;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) pend[tt][v] = first ? run[tt][v] : pend[tt][v] + run[tt][v];
    }
    if (sb + 1 < sb_end) wr = wnx;
  }
  // store: lane holds column n, rows t = 8 (v / 4) + 4 hf + v % 4 of each MFMA tile
  const int n = n0 + wn * 32 + nn;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int t = t0 + wt * 64 + tt * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
      if (n < a.N && t < tend) {
        if (a.ksplit > 1) a.part[((size_t)blockIdx.z * a.T + t) * a.N + n] = run[tt][v];  // one run per workgroup: its sum, combined by gemm_qi_reduce_kernel
        else {
          float *o = a.out + (size_t)t * a.ldo + n;
          *o = a.accumulate ? *o + pend[tt][v] : pend[tt][v];
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------ Q8_0 weights x Q8_0 activation rows (round 6; BASELINE configs[2])
// The decode engine's Q8_0 term (dec_core2.cuh Tile<T_Q8_0>, oracle term_q8_0): per 32-value block the exact integer dot, p_b = ((float)isum_b dw_b) dx_b, the
// superblock's T = p_0 + p_1 + ... + p_7 left to right; superblock terms and runs combine as for the K-quants.  One v_mfma_i32_32x32x32_i8 IS one block of 32 tokens x 32
// weight rows (K = 32, int32 accumulate: exact); the fix-up (convert, two multiplies, one add per output and block) runs on the vector ALU beside the other wave's
// MFMAs.  Same tile, staging and split scheme as gemm_qi_kernel; LDS per superblock: 128 tokens x 256 int8 (16-byte chunks XOR-swizzled by token & 15) + the block
// scales [8][128] f32, double buffered by LDS-DMA.
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
constexpr int L8_ACT = TT * 256, L8_YD = 8 * TT * 4, L8_BUF = L8_ACT + L8_YD, L8_TOTAL = 2 * L8_BUF;
__global__ void __launch_bounds__(GT) gemm_q80_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * TN;
  const int tbeg = a.win ? a.win[0] : 0, tend = a.win ? a.win[1] : a.T;  // the launch's token rows [tbeg, tend) of the operand buffers (one expert's routes; all rows when dense)
  const int t0 = tbeg + blockIdx.y * TT;
  if (t0 >= tend) return;
  const int wn = wave & 3, wt = wave >> 2;
  const int S = a.K / 256, Cs = (S + 3) / 4;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, (short)0, (int)a.w_bytes, 0x00020000);
  constexpr int REC = REC_Q6K;
  struct WRegs { v4u q[8]; v4u hs; };
  const unsigned panel = (unsigned)((n0 + wn * 32) >> 5);
  // the weight registers are a ring in place: piece b of superblock sb + 1 is requested as soon as piece b of sb has been multiplied (a second register set for
  // the next superblock spilled 68 VGPRs)
  auto rec_of = [&](int sb) { return (panel * (unsigned)S + (unsigned)sb) * (unsigned)REC; };
  auto load_piece = [&](WRegs &wr, unsigned rec, int c) { wr.q[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + (unsigned)(c * 64 + lane) * 16u, 0, 0); };
  auto load_hs = [&](WRegs &wr, unsigned rec) { wr.hs = __builtin_amdgcn_raw_buffer_load_b128(rw, rec + 8192u + (unsigned)nn * 16u, 0, 0); };
  auto load_w = [&](WRegs &wr, int sb) {
    const unsigned rec = rec_of(sb);
#pragma unroll
    for (int c = 0; c < 8; ++c) load_piece(wr, rec, c);
    load_hs(wr, rec);
  };
  const int tlast = tend - 1 - t0;
  const int8_t *q8 = (const int8_t *)a.qf;
  const float *yd8 = (const float *)a.bsf;
  auto stage = [&](int sb, int buf) {
    char *base = smem + buf * L8_BUF;
    const char *gq = (const char *)q8 + ((size_t)sb * a.T + t0) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // one DMA instruction = 1 KiB = 4 token rows; lane l -> row 4 quad + (l >> 4), chunk position l & 15
      const int quad = wave * 4 + i;
      const int tr = 4 * quad + (lane >> 4), trc = min(tr, tlast);
      MRS_GLDS16(gq + (size_t)trc * 256 + (((lane & 15) ^ (tr & 15)) << 4), base + quad * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // block scales: wave w stages block w, 64 tokens per instruction
      const int tr = i * 64 + lane;
      MRS_GLDS4((const char *)yd8 + (((size_t)sb * 8 + wave) * a.T + t0 + min(tr, tlast)) * 4, base + L8_ACT + (wave * TT + i * 64) * 4);
    }
  };
  float run[2][16], pend[2][16];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 16; ++v) { run[tt][v] = 0.f; pend[tt][v] = 0.f; }
  WRegs wr;
  const int sb_begin = a.ksplit > 1 ? (int)blockIdx.z * Cs : 0, sb_end = a.ksplit > 1 ? min(S, sb_begin + Cs) : S;
  if (sb_begin >= sb_end) {
    if (a.ksplit > 1) {  // a run without superblocks contributes +0
      const int n = n0 + wn * 32 + nn;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int t = t0 + wt * 64 + tt * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
          if (n < a.N && t < tend) a.part[((size_t)blockIdx.z * a.T + t) * a.N + n] = 0.f;
        }
    }
    return;
  }
  load_w(wr, sb_begin); stage(sb_begin, sb_begin & 1);
  const int trow = wt * 64 + nn;
  const i16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int sb = sb_begin; sb < sb_end; ++sb) {
    const int buf = sb & 1;
    MRS_WAIT_VMCNT0();
    __syncthreads();
    const bool more = sb + 1 < sb_end;
    const unsigned rec_next = more ? rec_of(sb + 1) : 0xF0000000u;  // past the tensor: zeros, no traffic (the request stays unconditional: exact waits)
    if (more) stage(sb + 1, buf ^ 1);
    const char *act_s = smem + buf * L8_BUF;
    const float *yd_s = (const float *)(act_s + L8_ACT);
    float Tt[2][16];
    auto act_frag8 = [&](int tt, int b) { return *(const i4v *)(act_s + (trow + 32 * tt) * 256 + (((2 * b + hf) ^ (nn & 15)) << 4)); };
    i4v afn[2] = {act_frag8(0, 0), act_frag8(1, 0)};  // the fragments are read one block ahead of their MFMAs
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned hw = b < 2 ? wr.hs.x : b < 4 ? wr.hs.y : b < 6 ? wr.hs.z : wr.hs.w;
      const float dwb = half_bits_to_float((uint16_t)((b & 1) ? (hw >> 16) : (hw & 0xffffu)));
      const i4v wq = __builtin_bit_cast(i4v, wr.q[b]);
      const i4v afc[2] = {afn[0], afn[1]};
      if (b + 1 < 8) { afn[0] = act_frag8(0, b + 1); afn[1] = act_frag8(1, b + 1); }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const i4v af = afc[tt];
        const i16v is = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, wq, zero, 0, 0, 0);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 y4 = *(const float4 *)(yd_s + b * TT + wt * 64 + tt * 32 + 8 * q4 + 4 * hf);
          const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int v = 4 * q4 + k;
            const float p = ((float)is[v] * dwb) * yy[k];
            Tt[tt][v] = b == 0 ? p : Tt[tt][v] + p;
          }
        }
      }
      load_piece(wr, rec_next, b);
      if (b == 7) load_hs(wr, rec_next);
      __builtin_amdgcn_sched_barrier(0);  // one block at a time: hoisting the MFMAs of later blocks keeps 16 result registers each alive (37 spilled VGPRs)
    }
    const bool cfirst = (sb % Cs) == 0;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 16; ++v) run[tt][v] = cfirst ? Tt[tt][v] : run[tt][v] + Tt[tt][v];
    if ((sb + 1) % Cs == 0 || sb + 1 == sb_end) {
      const bool first = sb < Cs;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) pend[tt][v] = first ? run[tt][v] : pend[tt][v] + run[tt][v];
    }
  }
  const int n = n0 + wn * 32 + nn;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int t = t0 + wt * 64 + tt * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
      if (n < a.N && t < tend) {
        if (a.ksplit > 1) a.part[((size_t)blockIdx.z * a.T + t) * a.N + n] = run[tt][v];
        else {
          float *o = a.out + (size_t)t * a.ldo + n;
          *o = a.accumulate ? *o + pend[tt][v] : pend[tt][v];
        }
      }
    }
}
// out[t][n] (+)= ((c0 + c1) + c2) + c3 of the four run sums (the row combination of ORD-U)
__global__ void __launch_bounds__(256) gemm_qi_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int T, int N, int ldo, int accumulate) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4, tot = (size_t)T * N;
  if (i >= tot) return;
  const float4 c0 = *(const float4 *)(part + i), c1 = *(const float4 *)(part + tot + i), c2 = *(const float4 *)(part + 2 * tot + i), c3 = *(const float4 *)(part + 3 * tot + i);
  float4 r = make_float4(((c0.x + c1.x) + c2.x) + c3.x, ((c0.y + c1.y) + c2.y) + c3.y, ((c0.z + c1.z) + c2.z) + c3.z, ((c0.w + c1.w) + c2.w) + c3.w);
  const size_t t = i / N, n = i % N;  // N % 4 == 0: the four values share a row
  float4 *o = (float4 *)(out + t * ldo + n);
  if (accumulate) { const float4 old = *o; r.x = old.x + r.x; r.y = old.y + r.y; r.z = old.z + r.z; r.w = old.w + r.w; }
  *o = r;
}

}  // namespace qi
}  // namespace mrs

using namespace mrs;

extern "C" size_t mrs_gemm_qi_repack_bytes(int type, long long n, long long k) {
  if (!qi::qi_type(type) || n <= 0 || k <= 0 || k % 256) return 0;
  return qi::qi_tensor_bytes(type, n, k);
}
// GGUF blocks [n][k / 256] (q4_k / q5_k / q6_k) -> the MFMA-order copy mrs_gemm_qi reads
extern "C" int mrs_gemm_qi_repack(const void *gguf_blocks, int type, long long n, long long k, void *dst, void *stream) {
  if (!mrs_gemm_qi_repack_bytes(type, n, k) || !gguf_blocks || !dst) return -1;
  const long long total = ((n + 31) / 32) * (k / 256) * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (type == T_Q4_K) hipLaunchKernelGGL(qi::qi_repack_kernel<T_Q4_K>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gguf_blocks, (uint8_t *)dst, n, (int)k, total);
  else if (type == T_Q5_K) hipLaunchKernelGGL(qi::qi_repack_kernel<T_Q5_K>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gguf_blocks, (uint8_t *)dst, n, (int)k, total);
  else if (type == T_Q8_0) hipLaunchKernelGGL(qi::qi_repack_kernel<T_Q8_0>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gguf_blocks, (uint8_t *)dst, n, (int)k, total);
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
// mrs_qi_quantize_for: the image format the weight type `w_type` multiplies with (Q8_0 weights: Q8_0 blocks; K-quants: Q8_K) -- GgufMatMul::forward_raw's
// per-format vec_dot partner (gguf/mod.rs:465-478)
extern "C" int mrs_qi_quantize_for(int w_type, const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream);
extern "C" int mrs_qi_quantize(const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream) {
  return mrs_qi_quantize_for(T_Q4_K, x, x2, ldx, norm_w, eps, T, K, act, xtmp, stream);
}
extern "C" int mrs_qi_quantize_for(int w_type, const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream) {
  if (!x || !act || T <= 0 || K <= 0 || K % 256 || (x2 && !xtmp) || !qi::qi_type(w_type)) return -1;
  qi::QuantArgs a{};
  a.x = x; a.x2 = x2; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.K = K; a.T = T; a.xtmp = xtmp; a.mode = dec2::act_mode_for(w_type);
  qi_split(act, T, K, &a.qf, &a.yd, &a.bsf);
  const size_t lds = (dec2::act_bytes(K, 1) + 15) & ~(size_t)15;
  if (lds > 158 * 1024) return -2;
  if (x2) { auto kern = qi::qi_quantize_kernel<true>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(T), dim3(dec2::NT), lds, (hipStream_t)stream, a); }
  else { auto kern = qi::qi_quantize_kernel<false>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(T), dim3(dec2::NT), lds, (hipStream_t)stream, a); }
  return 0;
}
// out[t * ldo + n] (+)= W[n] . act[t] for t < T, n < N in the decode engine's arithmetic and f32 order.  w_qi: mrs_gemm_qi_repack output; act: mrs_qi_quantize output.
// workspace for the split launch of GEMMs with few workgroups (N * T small): 4 x T x N floats
extern "C" size_t mrs_gemm_qi_workspace_bytes(int T, int max_n_split) { return (size_t)4 * (size_t)T * (size_t)max_n_split * 4; }
extern "C" int mrs_gemm_qi_ws(const void *w_qi, int type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *workspace, size_t workspace_bytes,
                              void *stream);
extern "C" int mrs_gemm_qi(const void *w_qi, int type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *stream) {
  return mrs_gemm_qi_ws(w_qi, type, N, K, act, T, out, ldo, accumulate, nullptr, 0, stream);
}
// mrs_gemm_qi_win: the grouped form (MoE prompts): `act` holds T_total operand rows, only rows [win[0], win[1]) (device ints) are multiplied and only those rows of `out`
// are written; `max_rows` (host) bounds end - begin and sizes the grid.  No split launch (the expert GEMMs fill the chip).
static int gemm_qi_launch(const void *w_qi, int type, int N, int K, const void *act, int T, const int *win, int max_rows, float *out, int ldo, int accumulate, void *workspace,
                          size_t workspace_bytes, void *stream);
extern "C" int mrs_gemm_qi_win(const void *w_qi, int type, int N, int K, const void *act, int T_total, const int *win, int max_rows, float *out, int ldo, int accumulate, void *stream) {
  if (!win || max_rows <= 0 || max_rows > T_total) return -1;
  return gemm_qi_launch(w_qi, type, N, K, act, T_total, win, max_rows, out, ldo, accumulate, nullptr, 0, stream);
}
extern "C" int mrs_gemm_qi_ws(const void *w_qi, int type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *workspace, size_t workspace_bytes,
                              void *stream) {
  return gemm_qi_launch(w_qi, type, N, K, act, T, nullptr, T, out, ldo, accumulate, workspace, workspace_bytes, stream);
}
static int gemm_qi_launch(const void *w_qi, int type, int N, int K, const void *act, int T, const int *win, int max_rows, float *out, int ldo, int accumulate, void *workspace,
                          size_t workspace_bytes, void *stream) {
  if (!w_qi || !act || !out || !qi::qi_type(type) || N <= 0 || T <= 0 || K <= 0 || K % 256) return -1;
  qi::GemmArgs a{};
  a.w = (const uint8_t *)w_qi; a.w_bytes = (unsigned)qi::qi_tensor_bytes(type, N, K); a.type = type; a.N = N; a.K = K; a.T = T;
  _Float16 *qf, *bsf; float *yd;
  qi_split((void *)act, T, K, &qf, &yd, &bsf);
  if ((size_t)(K / 256) * T * 512 >= 0x7fffffffull || qi::qi_tensor_bytes(type, N, K) >= 0xffffff00ull) return -2;
  a.qf = qf; a.yd = yd; a.bsf = bsf; a.out = out; a.ldo = ldo; a.accumulate = accumulate; a.win = win;
  dim3 grid((N + qi::TN - 1) / qi::TN, (max_rows + qi::TT - 1) / qi::TT);
  // few workgroups (o_proj / down_proj of a 512-token prompt: 128): one workgroup per run of superblocks, then the reduce -- the same additions in the same order
  static const int split_max = [] { const char *e = getenv("MRS_GEMM_QI_SPLIT_BELOW"); return e ? atoi(e) : 200; }();
  a.ksplit = 1; a.part = nullptr;
  if (!win && (int)(grid.x * grid.y) < split_max && K / 256 >= 4 && N % 4 == 0 && workspace && workspace_bytes >= (size_t)4 * T * N * 4) { a.ksplit = 4; a.part = (float *)workspace; grid.z = 4; }
  if (type == T_Q8_0) { auto kern = qi::gemm_q80_kernel; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(qi::GT), qi::L8_TOTAL, (hipStream_t)stream, a); }
  else if (type == T_Q4_K) { auto kern = qi::gemm_qi_kernel<T_Q4_K>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(qi::GT), qi::LDS_TOTAL, (hipStream_t)stream, a); }
  else if (type == T_Q5_K) { auto kern = qi::gemm_qi_kernel<T_Q5_K>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(qi::GT), qi::LDS_TOTAL, (hipStream_t)stream, a); }
  else { auto kern = qi::gemm_qi_kernel<T_Q6_K>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(qi::GT), qi::LDS_TOTAL, (hipStream_t)stream, a); }
  if (a.ksplit > 1) hipLaunchKernelGGL(qi::gemm_qi_reduce_kernel, dim3((unsigned)(((size_t)T * N / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.part, out, T, N, ldo, accumulate);
  return 0;
}

// ------------------------------------------------------------------------------------------------ MoE prompts in the decode engine's arithmetic (round 6; BASELINE configs[4])
// The decode step of a sparse-MoE layer (host/runtime.cpp forward_engine; SparseMoeBlock::forward, models/mixtral.rs:280-304) multiplies ONE token's quantized row by its
// top-k experts' gate / up rows, quantizes silu(g) * u per expert slot, multiplies by that expert's down rows and folds the slots into the residual stream in slot order:
// h <- (h * rs + w_0 s_0) * 1 + w_1 s_1 ...  A prompt does the same per token through the grouped form of the exact GEMM: the tokens' operand rows are gathered into
// expert-sorted order (launch_moe_dispatch's table), every expert multiplies its window of rows (mrs_gemm_qi_win), and the fold below restores slot order per token.
namespace mrs {
namespace qi {
// operand rows of `src` (T rows) -> `dst` (R rows): dst row r = src row sorted[r] / tk (sorted == nullptr: r / tk ... unused); inv[sorted[r]] = r.  One workgroup per row r.
__global__ void __launch_bounds__(256) qi_gather_rows_kernel(const char *__restrict__ src, char *__restrict__ dst, int T, int R, int S, int mode, const int *__restrict__ sorted,
                                                             int tk, int *__restrict__ inv, size_t off1s, size_t off2s, size_t off1d, size_t off2d) {
  const int r = blockIdx.x, route = sorted[r], t = route / tk;
  if (threadIdx.x == 0 && inv) inv[route] = r;
  if (mode == ACT_Q80) {  // q8 [S][rows][256] at 0; block scales [S][8][rows] f32 at off2
    for (int i = threadIdx.x; i < S * 16; i += 256) {
      const int sb = i >> 4, ch = i & 15;
      *(v4u *)(dst + ((size_t)sb * R + r) * 256 + ch * 16) = *(const v4u *)(src + ((size_t)sb * T + t) * 256 + ch * 16);
    }
    for (int i = threadIdx.x; i < S * 8; i += 256) ((float *)(dst + off2d))[(size_t)i * R + r] = ((const float *)(src + off2s))[(size_t)i * T + t];
  } else {  // qf [S][rows][256] f16 at 0; yd [S][rows] f32 at off1; bsf [S][rows][16] f16 at off2
    for (int i = threadIdx.x; i < S * 32; i += 256) {
      const int sb = i >> 5, ch = i & 31;
      *(v4u *)(dst + ((size_t)sb * R + r) * 512 + ch * 16) = *(const v4u *)(src + ((size_t)sb * T + t) * 512 + ch * 16);
    }
    for (int i = threadIdx.x; i < S * 2; i += 256) {
      const int sb = i >> 1, ch = i & 1;
      *(v4u *)(dst + off2d + ((size_t)sb * R + r) * 32 + ch * 16) = *(const v4u *)(src + off2s + ((size_t)sb * T + t) * 32 + ch * 16);
    }
    for (int sb = threadIdx.x; sb < S; sb += 256) ((float *)(dst + off1d))[(size_t)sb * R + r] = ((const float *)(src + off1s))[(size_t)sb * T + t];
  }
}
// h[t][n] <- fold over the token's slots sl = 0 .. tk - 1 of  h * (sl == 0 ? rs : 1) + y[inv[t * tk + sl]][n] * w[t * tk + sl]  (two roundings per slot, the RESID
// epilogue's expression: dec_gemv.cuh EPI_RESID / EPI_RESID2)
__global__ void __launch_bounds__(256) moe_fold_exact_kernel(float *__restrict__ h, float rs, const float *__restrict__ y, const int *__restrict__ inv, const float *__restrict__ w,
                                                            int d, int tk) {
  const int t = blockIdx.x;
  for (int n = threadIdx.x * 4; n < d; n += 1024) {
    float4 v = *(const float4 *)(h + (size_t)t * d + n);
    for (int sl = 0; sl < tk; ++sl) {
      const float ws = w[(size_t)t * tk + sl], sc = sl == 0 ? rs : 1.0f;
      const float4 yy = *(const float4 *)(y + (size_t)inv[(size_t)t * tk + sl] * d + n);
      v.x = v.x * sc + yy.x * ws; v.y = v.y * sc + yy.y * ws; v.z = v.z * sc + yy.z * ws; v.w = v.w * sc + yy.w * ws;
    }
    *(float4 *)(h + (size_t)t * d + n) = v;
  }
}
// h <- h * rs + y  (tensor-parallel row-parallel projections of a prompt: the decode step's RESID epilogue with resid_scale = 1 / world, then ONE sum all-reduce of h)
__global__ void __launch_bounds__(256) resid_scale_add_kernel(float *__restrict__ h, float rs, const float *__restrict__ y, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      float4 a = *(float4 *)(h + i); const float4 b = *(const float4 *)(y + i);
      a.x = a.x * rs + b.x * 1.0f; a.y = a.y * rs + b.y * 1.0f; a.z = a.z * rs + b.z * 1.0f; a.w = a.w * rs + b.w * 1.0f;
      *(float4 *)(h + i) = a;
    } else for (size_t j = i; j < n; ++j) h[j] = h[j] * rs + y[j] * 1.0f;
  }
}
}  // namespace qi
}  // namespace mrs
extern "C" int mrs_qi_gather_rows(int w_type, const void *act_src, int T, void *act_dst, int R, int K, const int *sorted_routes, int top_k, int *inv, void *stream) {
  if (!act_src || !act_dst || !sorted_routes || T <= 0 || R <= 0 || K <= 0 || K % 256 || top_k <= 0 || !qi::qi_type(w_type)) return -1;
  _Float16 *qs, *bs, *qd, *bd; float *ys, *yd;
  qi_split((void *)act_src, T, K, &qs, &ys, &bs);
  qi_split(act_dst, R, K, &qd, &yd, &bd);
  hipLaunchKernelGGL(qi::qi_gather_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const char *)act_src, (char *)act_dst, T, R, K / 256, dec2::act_mode_for(w_type),
                     sorted_routes, top_k, inv, (size_t)((char *)ys - (char *)qs), (size_t)((char *)bs - (char *)qs), (size_t)((char *)yd - (char *)qd), (size_t)((char *)bd - (char *)qd));
  return 0;
}
extern "C" int mrs_moe_fold_exact(float *h, float resid_scale, const float *y_sorted, const int *inv, const float *weights, int T, int d, int top_k, void *stream) {
  if (!h || !y_sorted || !inv || !weights || T <= 0 || d <= 0 || d % 4 || top_k <= 0) return -1;
  hipLaunchKernelGGL(qi::moe_fold_exact_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, h, resid_scale, y_sorted, inv, weights, d, top_k);
  return 0;
}
extern "C" int mrs_resid_scale_add_f32(float *h, float resid_scale, const float *y, size_t n, void *stream) {
  if (!n) return 0;
  size_t g = (n / 4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  hipLaunchKernelGGL(qi::resid_scale_add_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, h, resid_scale, y, n);
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
// One wave owns `qw` consecutive prompt tokens of one kv head (a workgroup = 4 waves = 4 * qw tokens) and walks the splits of their contexts in ascending order; with one
// block per split (max_context_len <= 2048, ONE) the K / V block sits in registers once for all `qw` queries -- the L2 traffic of a prompt drops by 4 * qw / (1.5) against one
// workgroup per token.  Per QUERY the arithmetic is the decode kernel's: every split's (m, l, o) comes from dec::attn_block_update (the function dec::attn_split_core runs),
// and the merge is dec::attn_merge_core's (weights fast_exp(m_j - max m), sums in ascending split order, multiply and add separate).  Because the merge weights need the
// maximum over ALL splits, the walk runs twice: pass A computes (m_j, l_j) from K alone, pass B recomputes the same scores (same inputs, same instructions: same bits), forms
// o_j and accumulates o_j * w_j at once -- no partial is ever stored.
template <int G, class CT, bool ONE>
__global__ void __launch_bounds__(256) prefill_attn_exact_kernel(const dec::AttnArgs a, const int qw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 128;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kvh = blockIdx.x, tile = (int)gridDim.y - 1 - (int)blockIdx.y;  // longest contexts first
  const int T = a.num_seqs, ms = a.max_splits, bpw = a.bpw;
  const int t0 = (tile * 4 + wave) * qw, nq = min(qw, T - t0);
  if (nq <= 0) return;  // no workgroup barrier below: a wave may leave
  const size_t per_wave = (size_t)qw * G * HD * 2 + G * 32 + (size_t)qw * G * ms * 2 + (size_t)qw * G * 64;
  float *q_s = (float *)smem + wave * per_wave, *p_s = q_s + (size_t)qw * G * HD, *acc = p_s + G * 32, *ml = acc + (size_t)qw * G * HD, *sall = ml + (size_t)qw * G * ms * 2;
  int ctx_max = 0;
  for (int i = 0; i < nq; ++i) {
    const float *qg = a.q + (size_t)(t0 + i) * a.q_stride + (size_t)kvh * G * HD;
    for (int k = lane * 4; k < G * HD; k += 256) *(float4 *)(q_s + (size_t)i * G * HD + k) = *(const float4 *)(qg + k);
    for (int k = lane; k < G * HD; k += 64) acc[(size_t)i * G * HD + k] = 0.f;
    ctx_max = max(ctx_max, (int)a.context_lens[t0 + i]);
  }
  for (int k = 0; k < nq * G; ++k) sall[k * 64 + lane] = 0.f;  // the running sum of l_j * w_j: one private copy per lane (wave-uniform value)
  MRS_WAVE_SYNC();
  const uint32_t *bt = a.block_tables;  // one sequence: every token reads the same row
  const int ns_w = (((ctx_max + 31) / 32) + bpw - 1) / bpw;
  auto base_of = [&](int b) { return (size_t)bt[b] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride; };
  auto walk = [&](auto with_v) {
    constexpr bool WITH_V = decltype(with_v)::value;
    for (int sp = 0; sp < ns_w; ++sp) {
      const int b0 = sp * bpw;
      int4 kr[8], vr[8];
      if (ONE) dec::attn_load_block<WITH_V>(a, base_of(b0), kr, vr);
      for (int i = 0; i < nq; ++i) {
        const int ctx = (int)a.context_lens[t0 + i], nblk = (ctx + 31) / 32;
        if (sp >= (nblk + bpw - 1) / bpw) continue;  // this token's decode step has no such split
        const int b1 = min(b0 + bpw, nblk), lo = a.window > 0 && ctx > a.window ? ctx - a.window : 0;
        float m[G], l[G], o0[G], o1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { m[g] = -FLT_MAX; l[g] = 0.f; o0[g] = 0.f; o1[g] = 0.f; }
        if (b1 * 32 > lo) {  // else: the split lies before the sliding window -- what a fully masked pass gives (mrs_dec_attention publishes the same)
          for (int b = b0; b < b1; ++b) {
            if (!ONE) dec::attn_load_block<WITH_V>(a, base_of(b), kr, vr);
            dec::attn_block_update<G, CT, WITH_V>(a, kr, vr, q_s + (size_t)i * G * HD, p_s, b, b == b0, ctx, lo, m, l, o0, o1);
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float *e = ml + ((size_t)(i * G + g) * ms + sp) * 2;
          if (!WITH_V) {
            e[0] = m[g]; e[1] = l[g];  // every lane writes the same value
          } else {  // dec::attn_merge_core, one split: s += l_j * w_j; acc += o_j * w_j
            const float wj = e[0], lw = l[g] * wj;
            sall[(i * G + g) * 64 + lane] = sall[(i * G + g) * 64 + lane] + lw;
            float *ao = acc + (size_t)(i * G + g) * HD;
            const float u0 = o0[g] * wj, u1 = o1[g] * wj;
            ao[lane] = ao[lane] + u0;
            ao[lane + 64] = ao[lane + 64] + u1;
          }
        }
      }
    }
  };
  walk(std::false_type{});
  MRS_WAVE_SYNC();
  for (int i = 0; i < nq; ++i) {  // merge weights: w_j = fast_exp(m_j - max_j m_j), lane j <-> split j
    const int nblk = ((int)a.context_lens[t0 + i] + 31) / 32, ns = (nblk + bpw - 1) / bpw;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float *e = ml + (size_t)(i * G + g) * ms * 2;
      const float mj = lane < ns ? e[lane * 2] : -FLT_MAX;
      const float mx = wave_max(mj);
      const float w = fast_exp_ref(mj - mx);
      if (lane < ns) e[lane * 2] = w;
    }
  }
  MRS_WAVE_SYNC();
  walk(std::true_type{});
  MRS_WAVE_SYNC();
  for (int i = 0; i < nq; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float inv = 1.0f / sall[(i * G + g) * 64 + lane];
      const float *ao = acc + (size_t)(i * G + g) * HD;
      float *o = a.out + ((size_t)(t0 + i) * a.num_heads + kvh * G + g) * HD;
      o[lane] = ao[lane] * inv; o[lane + 64] = ao[lane + 64] * inv;
    }
}
// ------------------------------------------------------------------------------------------------ the same attention on the matrix cores (round 6)
// prefill_attn_exact_kernel spends a prompt's attention on the vector ALU (224 us per layer at 512 tokens: 30 % of the default TTFT; quadratic: most of a 2048-token
// prompt).  v_mfma_f32_32x32x2_f32 takes f32 operands and IS the f32 FMA chain over k, bit for bit (d = fma(a1, b1, fma(a0, b0, c)): profiles/experiments/
// mfma_f32_probe.hip on the MI355X; MI355X_MICROARCH.md, matrix cores), so the decode kernel's chains run on it unchanged:
//   scores   s = chain over dims 0 .. 63 (acc0) + chain over dims 64 .. 127 (acc1): 32 + 32 MFMAs per 32 tokens x 32 queries (the decode wave's two halves)
//   P . V    o = chain over the block's 32 tokens in ascending order, started from the split's running o * alpha (or 0): 16 MFMAs per 32 dims x 32 queries
// and everything between them is the decode kernel's per-element arithmetic (scale, mask, fast_exp_ref, the pairwise 32-token sum tree of its DPP reduction).  Per QUERY
// the splits (bpw blocks each), their online-softmax recurrence and the merge over splits are attn_split_core's / attn_merge_core's, evaluated for 32 queries at once.
// Workgroup = 4 waves = one (query head, tile of 32 consecutive prompt tokens); K / V blocks in groups of four:
//   phase 1: wave w computes block 4 g + w's scores, probabilities (-> LDS), block sum, alpha and (last block of a split) the merge weight fast_exp(m_split - M);
//   phase 2: wave w owns dims 32 w .. 32 w + 31 of the output: P . V of the group's blocks in ascending order, the split recurrences and the merge sums --
//            the f32 additions of one query happen in the decode order although four waves share the work.
// M (the maximum over the query's splits, needed by every merge weight) comes from a first pass over the scores (pass A: block maxima only).
template <class CT> __device__ __forceinline__ float cvt16_lo(unsigned w);
template <class CT> __device__ __forceinline__ float cvt16_hi(unsigned w);
template <> __device__ __forceinline__ float cvt16_lo<bf16_t>(unsigned w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float cvt16_hi<bf16_t>(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ float cvt16_lo<f16_t>(unsigned w) { return half_bits_to_float((uint16_t)(w & 0xffffu)); }
template <> __device__ __forceinline__ float cvt16_hi<f16_t>(unsigned w) { return half_bits_to_float((uint16_t)(w >> 16)); }
#ifndef MRS_MFMA_F32_K1_FIRST
#define MRS_MFMA_F32_K1_FIRST 0  // 1: the instruction adds k = 1 before k = 0 (then the lane halves swap dims / tokens); the probe says 0 on gfx950
#endif
constexpr int PM_NW = 4;                                   // waves per workgroup
constexpr int PM_Q = 64 * 64 * 4;                          // Q operand [64 steps][64 lanes] f32
constexpr int PM_P = PM_NW * 32 * 32 * 4;                  // probabilities of the group's blocks [slot][token][query] f32
constexpr int PM_SC = PM_NW * 3 * 32 * 4;                  // per slot and query: block sum, alpha, merge weight
// + per block and query the block maximum of pass A: 128 bytes per 32-token block of the longest context (sized by the launcher)
template <class CT>
__global__ void __launch_bounds__(64 * PM_NW, 2) prefill_attn_mfma_kernel(const dec::AttnArgs a, int max_blocks_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const int kp = MRS_MFMA_F32_K1_FIRST ? 1 - hf : hf;  // which element of a pair (dims 2 s, 2 s + 1 / tokens 2 s, 2 s + 1) this lane half supplies so that the chain ascends
  const int head = blockIdx.x, G = a.num_heads / a.num_kv_heads, kvh = head / G;
  // longest contexts first; when the whole grid is resident at once (two workgroups per CU: heads x tiles <= 512) the second half of the dispatch order runs the SHORT
  // tiles in ascending order, so that the two workgroups sharing a CU's matrix pipes are (longest, shortest), (second longest, second shortest), ...: every CU gets the
  // same number of blocks (a 512-token prompt: 17 per pair instead of 24 .. 10)
  const int nty = (int)gridDim.y, hy = (nty + 1) / 2;
  const bool paired = (int)gridDim.x * nty <= 512 && nty > 1;
  const int tile = !paired ? nty - 1 - (int)blockIdx.y : ((int)blockIdx.y < hy ? nty - 1 - (int)blockIdx.y : (int)blockIdx.y - hy);
  const int T = a.num_seqs, bpw = a.bpw;
  const int t_q = min(tile * 32 + j, T - 1);
  const bool q_live = tile * 32 + j < T;
  float *q_s = (float *)smem;
  float *p_s = (float *)(smem + PM_Q);
  float *sc_s = (float *)(smem + PM_Q + PM_P);
  float *mx_s = (float *)(smem + PM_Q + PM_P + PM_SC);  // [block][query], max_blocks_lds blocks
  (void)max_blocks_lds;
  // ---- the tile's queries as the B operand of the score MFMAs: step s <-> dims 2 s, 2 s + 1; lane (query j, half) holds q[j][2 s + kp]
  {
    const float *qg = a.q + (size_t)t_q * a.q_stride + (size_t)head * 128;
    for (int m = wave; m < 32; m += PM_NW) {  // float4 = dims 4 m .. 4 m + 3 -> steps 2 m, 2 m + 1
      const float4 v = *(const float4 *)(qg + 4 * m);
      q_s[(2 * m) * 64 + lane] = kp ? v.y : v.x;
      q_s[(2 * m + 1) * 64 + lane] = kp ? v.w : v.z;
    }
  }
  const int ctx = (int)a.context_lens[t_q];
  const int lo = a.window > 0 && ctx > a.window ? ctx - a.window : 0;
  const int t_last = min(tile * 32 + 31, T - 1);
  const int ctx_max = (int)a.context_lens[t_last];  // consecutive prompt positions: the tile's last token has the longest context
  const int nblk = (ctx_max + 31) / 32;
  const uint32_t *bt = a.block_tables;
  __syncthreads();
  f16v zero;
#pragma unroll
  for (int v = 0; v < 16; ++v) zero[v] = 0.f;
  // K of a block as the A operand of the score MFMAs: lane (token j, half) supplies K[token j][2 s + kp]; chunk c = dims 8 c .. 8 c + 7 (16 bytes).  The 16 chunk
  // registers are a ring in place: chunk c of the NEXT block the wave will score is requested as soon as chunk c of the current one has been converted (a block's
  // loads used to sit exposed in front of its MFMAs; a second register set costs the occupancy: 220 + 48 registers)
  auto k_ptr = [&](int b) { return a.k_cache + (size_t)bt[b] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride + (size_t)j * 8; };
  auto load_k = [&](int b, v4u (&kr)[16]) {
    const uint16_t *kb = k_ptr(b);
#pragma unroll
    for (int c = 0; c < 16; ++c) kr[c] = *(const v4u *)(kb + (size_t)c * 32 * 8);
  };
  // scores of block b for the 32 queries: lane (query j, half hf) gets tokens i = 8 (v / 4) + 4 hf + v % 4 (the accumulator layout); masked + scaled; returns the block max.
  // b_next: the block whose K replaces the registers (b itself when there is none: the reload is then unused, the code stays straight-line)
  auto scores = [&](v4u (&kr)[16], int b, int b_next, float (&val)[16], bool (&ok)[16]) -> float {
    const uint16_t *kn = k_ptr(b_next);
    f16v acc0 = zero, acc1 = zero;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const unsigned w4[4] = {kr[c].x, kr[c].y, kr[c].z, kr[c].w};
      float kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] = kp ? cvt16_hi<CT>(w4[u]) : cvt16_lo<CT>(w4[u]);
      kr[c] = *(const v4u *)(kn + (size_t)c * 32 * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float qv = q_s[(c * 4 + u) * 64 + lane];
        if (c < 8) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[u], qv, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[u], qv, acc1, 0, 0, 0);
      }
    }
    float mx = -FLT_MAX;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int pos = b * 32 + 8 * (v >> 2) + 4 * hf + (v & 3);
      ok[v] = pos < ctx && pos >= lo;
      const float sv = acc0[v] + acc1[v];
      val[v] = ok[v] ? sv * a.scale : -FLT_MAX;
      mx = fmaxf(mx, val[v]);
    }
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
  };
  v4u kcur[16];
  // ---- pass A: block maxima (wave w: blocks w, w + 4, ...); its last call reloads the wave's first block for pass B
  if (wave < nblk) load_k(wave, kcur);
  for (int b = wave; b < nblk; b += PM_NW) {
    float val[16]; bool ok[16];
    const float mx = scores(kcur, b, b + PM_NW < nblk ? b + PM_NW : wave, val, ok);
    if (hf == 0) mx_s[b * 32 + j] = mx;
  }
  __syncthreads();
  // per query: the running maximum inside every split and M = the maximum over the splits -- all from the block maxima (a maximum is exact in any order)
  float M = -FLT_MAX;
  for (int b = 0; b < nblk; ++b) M = fmaxf(M, mx_s[b * 32 + j]);
  // ---- pass B
  f16v acc = zero, osp = zero;  // merge accumulator and the current split's running output: dims 32 wave + 8 (v / 4) + 4 hf + v % 4 of query j
  float s_all = 0.f, l_run = 0.f;
  // V of block b, dims 32 wave .. + 31, as the A operand of the P . V MFMAs: lane (dim row, half) supplies V[dim][token 2 s + kp] = dword s of the dim's 32-token row
  auto load_v = [&](int b, v4u (&vr)[4]) {
    const size_t base = (size_t)bt[b] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride;
    const uint16_t *vb = a.v_cache + base + (size_t)(32 * wave + j) * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) vr[c] = *(const v4u *)(vb + c * 8);
  };
  for (int g0 = 0; g0 < nblk; g0 += PM_NW) {
    v4u vcur[4], vnext[4];
    load_v(g0, vcur);  // phase 2's first block: in flight during phase 1
    {  // phase 1
      const int b = g0 + wave;
      if (b < nblk) {
        float val[16]; bool ok[16];
        const float mx = scores(kcur, b, b + PM_NW < nblk ? b + PM_NW : b, val, ok);  // the next group's block arrives during phase 2
        const int b0 = (b / bpw) * bpw;  // first block of the split
        float m_before = -FLT_MAX;
        for (int bb = b0; bb < b; ++bb) m_before = fmaxf(m_before, mx_s[bb * 32 + j]);
        const float mn = fmaxf(m_before, mx);
        float pv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) pv[v] = ok[v] ? fast_exp_ref(val[v] - mn) : 0.f;
        // the decode wave's sum over the block's 32 tokens: pairwise tree in token order (xor 1, xor 2, half-row mirror, row mirror, xor 16 of its DPP reduction)
        float q4[4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) q4[aa] = (pv[4 * aa] + pv[4 * aa + 1]) + (pv[4 * aa + 2] + pv[4 * aa + 3]);  // tokens 8 aa + 4 hf + 0 .. 3
        float o8[4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) o8[aa] = q4[aa] + __shfl_xor(q4[aa], 32, 64);  // tokens 8 aa .. 8 aa + 7
        const float ps = (o8[0] + o8[1]) + (o8[2] + o8[3]);
        float *pw = p_s + (size_t)wave * 1024;
#pragma unroll
        for (int v = 0; v < 16; ++v) pw[(8 * (v >> 2) + 4 * hf + (v & 3)) * 32 + j] = pv[v];
        if (hf == 0) {
          float *sw = sc_s + (size_t)wave * 96;
          sw[j] = ps;
          sw[32 + j] = b == b0 ? 0.f : fast_exp_ref(m_before - mn);  // alpha (unused for the first block of a split)
          const bool last = (b + 1) % bpw == 0 || b + 1 == nblk;
          sw[64 + j] = last ? fast_exp_ref(mn - M) : 0.f;              // the split's merge weight: its final maximum is mn
        }
      }
    }
    __syncthreads();
    {  // phase 2: dims 32 wave .. + 31
      const int nb = min(PM_NW, nblk - g0);
      for (int sl = 0; sl < nb; ++sl) {
        const int b = g0 + sl;
        if (sl + 1 < nb) load_v(b + 1, vnext);
        const bool first = b % bpw == 0, last = (b + 1) % bpw == 0 || b + 1 == nblk;
        const float *sw = sc_s + (size_t)sl * 96;
        const float ps = sw[j], alpha = sw[32 + j], wsp = sw[64 + j];
        f16v o;
        if (first) { o = zero; l_run = ps; }
        else {
#pragma unroll
          for (int v = 0; v < 16; ++v) o[v] = osp[v] * alpha;
          l_run = l_run * alpha + ps;
        }
        const float *pr = p_s + (size_t)sl * 1024;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned w4[4] = {vcur[c].x, vcur[c].y, vcur[c].z, vcur[c].w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = 4 * c + u, tok = 2 * s + kp;
            float vv = kp ? cvt16_hi<CT>(w4[u]) : cvt16_lo<CT>(w4[u]);
            vv = b * 32 + tok < ctx_max ? vv : 0.f;  // slots past the tile's longest context may hold anything (the decode kernel zeroes past ITS context: those p are 0)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, pr[tok * 32 + j], o, 0, 0, 0);
          }
        }
        osp = o;
        if (last) {  // attn_merge_core: s += l_j w_j;  acc += o_j w_j  (multiply and add separate)
          const float lw = l_run * wsp;
          s_all = s_all + lw;
#pragma unroll
          for (int v = 0; v < 16; ++v) { const float t0 = o[v] * wsp; acc[v] = acc[v] + t0; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) vcur[c] = vnext[c];
      }
    }
    __syncthreads();
  }
  if (!q_live) return;
  const float inv = 1.0f / s_all;
  float *og = a.out + ((size_t)(tile * 32 + j) * a.num_heads + head) * 128 + 32 * wave;
#pragma unroll
  for (int aa = 0; aa < 4; ++aa)
    *(float4 *)(og + 8 * aa + 4 * hf) = make_float4(acc[4 * aa] * inv, acc[4 * aa + 1] * inv, acc[4 * aa + 2] * inv, acc[4 * aa + 3] * inv);
}

}  // namespace qi
}  // namespace mrs

// q f32 [T][q_stride] (RoPE applied), pages already hold the prompt's K / V; context_lens [T] = position + 1 of every prompt token (device), block_table = the
// sequence's row; out f32 [T][num_heads * 128].  max_context_len = the model's (it fixes the split length exactly as in mrs_dec_attention).
extern "C" int mrs_prefill_attention_exact(const float *q, const void *k_cache, const void *v_cache, const uint32_t *block_table, const uint32_t *context_lens, float *out,
                                           int T, int num_heads, int num_kv_heads, int head_size, int block_size, int q_stride, int kv_block_stride, int kv_head_stride,
                                           float scale, int max_context_len, int kv_dtype, int sliding_window, int max_prompt_ctx, void *stream) {
  if (!q || !out || head_size != 128 || block_size != 32 || T <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads || (kv_dtype != 0 && kv_dtype != 1)) return -1;
  const int G = num_heads / num_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8) return -1;
  mrs::dec::AttnArgs a{};
  a.q = q; a.k_cache = (const uint16_t *)k_cache; a.v_cache = (const uint16_t *)v_cache; a.block_tables = block_table; a.context_lens = context_lens;
  a.out = out; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads; a.max_blocks_per_seq = 0 /* every token reads the same row */; a.q_stride = q_stride;
  a.kv_block_stride = kv_block_stride; a.kv_head_stride = kv_head_stride; a.num_seqs = T; a.scale = scale; a.window = sliding_window > 0 ? sliding_window : 0;
  const int nblk = (max_context_len + 31) / 32;
  a.bpw = nblk <= 64 ? 1 : (nblk + 63) / 64;  // == mrs_dec_attention
  // (w_j, l_j) per split live in LDS: as many splits as the longest context of this prompt needs (<= 64 by the rule above)
  const int need_ctx = max_prompt_ctx > 0 && max_prompt_ctx < max_context_len ? max_prompt_ctx : max_context_len;
  a.max_splits = std::max(1, std::min(64, (((need_ctx + 31) / 32) + a.bpw - 1) / a.bpw));
  // round 6: the matrix-core form (prefill_attn_mfma_kernel: the same chains on v_mfma_f32_32x32x2_f32); MRS_PREFILL_ATTN_MFMA=0 keeps the vector-ALU kernel
  static const int use_mfma = [] { const char *e = getenv("MRS_PREFILL_ATTN_MFMA"); return e ? atoi(e) : 1; }();
  {
    const int nb = (need_ctx + 31) / 32;
    const size_t lds_m = (size_t)mrs::qi::PM_Q + mrs::qi::PM_P + mrs::qi::PM_SC + (size_t)nb * 128;
    if (use_mfma && lds_m <= 158 * 1024) {
      const dim3 gm(num_heads, (T + 31) / 32);
      if (kv_dtype == 1) { auto kern = mrs::qi::prefill_attn_mfma_kernel<mrs::bf16_t>; mrs::lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, gm, dim3(64 * mrs::qi::PM_NW), lds_m, (hipStream_t)stream, a, nb); }
      else { auto kern = mrs::qi::prefill_attn_mfma_kernel<mrs::f16_t>; mrs::lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, gm, dim3(64 * mrs::qi::PM_NW), lds_m, (hipStream_t)stream, a, nb); }
      return 0;
    }
  }
  static const int qw_env = [] { const char *e = getenv("MRS_PREFILL_ATTN_QW"); return e ? atoi(e) : 0; }();
  int qw = qw_env > 0 ? qw_env : 4;  // prompt tokens per wave
  auto lds_for = [&](int w) { return 4 * ((size_t)w * G * 128 * 2 + (size_t)G * 32 + (size_t)w * G * a.max_splits * 2 + (size_t)w * G * 64) * 4; };
  while (qw > 1 && (lds_for(qw) > 76 * 1024 || (T + 4 * qw - 1) / (4 * qw) * num_kv_heads < 512)) --qw;  // two workgroups per CU, and enough workgroups to fill the chip
  const size_t lds = lds_for(qw);
  if (lds > 158 * 1024) return -2;
  const dim3 grid(num_kv_heads, (T + 4 * qw - 1) / (4 * qw));
  hipStream_t s = (hipStream_t)stream;
#define MRS_PA(GG, CT) { if (a.bpw == 1) { auto kern = mrs::qi::prefill_attn_exact_kernel<GG, CT, true>; mrs::lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, qw); } \
                         else { auto kern = mrs::qi::prefill_attn_exact_kernel<GG, CT, false>; mrs::lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, qw); } }
#define MRS_PAG(CT) switch (G) { case 1: MRS_PA(1, CT) break; case 2: MRS_PA(2, CT) break; case 4: MRS_PA(4, CT) break; default: MRS_PA(8, CT) break; }
  if (kv_dtype == 1) { MRS_PAG(mrs::bf16_t) } else { MRS_PAG(mrs::f16_t) }
#undef MRS_PAG
#undef MRS_PA
  return 0;
}
