// ext_decode.hip -- MI355X-native fused decode path (not in the reference ABI; exported as mrs_*).
//
// The reference issues ~9 launches + 2 norms per decoder layer in decode: quantize, fused
// QKV GEMV, RoPE, reshape_and_cache, paged attention, quantize, o_proj GEMV, add, norm, quantize,
// gate/up GEMV(+GLU), quantize, down GEMV, add.  At batch 1 every GEMV is a 2-12 us HBM stream, so
// the ~1.5 us kernel boundaries and the tiny elementwise launches are a large fraction of the token
// time on MI355X.  Here each layer is FIVE launches built from the same GEMV core (mmvq_core.cuh):
//
//   1. qkv      prologue: RMSNorm(h)*w -> Q8_1 in LDS      epilogue: RoPE(q,k) -> q (f32), k/v -> paged cache
//   2. attn     paged attention over the cache (ext: split-KV + merge) -> Q8_1 blocks for o_proj
//   3. o_proj   prologue: stage Q8_1                        epilogue: h += W_o . attn
//   4. gate/up  prologue: RMSNorm(h)*w -> Q8_1 in LDS      epilogue: silu(g)*u -> Q8_1 blocks
//   5. down     prologue: stage Q8_1                        epilogue: h += W_d . act
//
// Numerics are those of the reference CPU path's dataflow (f32 residual stream, f32 norm / RoPE / SiLU,
// SURVEY 3.4) combined with the reference GPU path's Q8_1 activation quantisation (mmvq_gguf.cu), and are
// bit-identical to running the unfused C-ABI kernels in sequence (tests/test_llama_runner.py::test_fused_equals_reference_sequence_and_oracle).
#include "mmvq_core.cuh"
#include <stdio.h>
#include <stdlib.h>

namespace mrs {

enum : int { PRO_Q8_1 = 0, PRO_NORM = 1 };
enum : int { EPI_STORE = 0, EPI_RESID_ADD = 1, EPI_GLU_Q8_1 = 2, EPI_QKV_ROPE = 3 };

struct DecodeGemvArgs {
  const uint8_t *w[3];
  int wtype[3];
  int nrows[3];
  int K;
  // prologue
  const uint8_t *y_q8_1;  // PRO_Q8_1: [b][stride_col_y] blocks
  int stride_col_y;
  const float *x;         // PRO_NORM: [b][K] f32 residual stream
  const float *norm_w;    // [K]
  float eps;
  // epilogue
  float *out;             // STORE / RESID_ADD: out[c*out_stride + row]
  int out_stride;
  uint8_t *y_out;         // GLU_Q8_1: [b][y_out_stride] Q8_1 blocks
  int y_out_stride;
  int activation;
  float *q_out;           // QKV_ROPE: [b][nrows[0]]
  void *k_cache, *v_cache;
  const int64_t *slot_mapping;  // [b]
  const int32_t *positions;     // [b]
  const float *cos_t, *sin_t;   // [max_pos][rot_pairs]
  int head_dim, rot_pairs, num_kv_heads, block_size, cache_x;
  int rows_per_wg;
  float resid_scale;      // RESID_ADD: out = out * resid_scale + W.y  (1 = plain residual add; 1/world under tensor parallelism)
  // mixture of experts (stacked expert weights [E][N][K/blk], models/mixtral.rs:236-304): the expert of this launch is read on the
  // device, so a captured graph replays with new routing; acc_scale (the renormalised top-k weight) multiplies W.y in RESID_ADD
  const int32_t *expert_sel;  // nullptr = dense layer
  size_t expert_stride;       // bytes between experts
  const float *acc_scale;     // nullptr = 1
};

// fixed-order block reduction shared by the fused prologue and the standalone norm+quantize kernel
template <int NT> __device__ __forceinline__ float block_sum_fixed(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) s += red[w];
  __syncthreads();
  return s;
}

// RMSNorm(x)*w -> Q8_1 activation image in LDS, hot-format layout of mmvq_core.cuh (same values the C-ABI quantizer
// would produce from the f32 normed row): q pieces (swizzled), d8 = float(half(amax/127)), S = d8 * sum(q over each
// 16-run).  xs: f32 scratch [K] in LDS.
// The sum of squares always runs on the first 256 threads in a fixed order, so kernels with different workgroup sizes
// (and the standalone mrs_rms_norm kernel) produce bit-identical norms.
template <int NCOLS, int NT>
__device__ __forceinline__ ActLds stage_norm_q8_1(char *smem, const float *__restrict__ x, const float *__restrict__ nw, int K, float eps) {
  const int runs = K / 16, nblk = K / 32;
  const ActLds v = act_view<NCOLS>(smem, K);
  int8_t *q = (int8_t *)v.q;
  float *d8 = (float *)v.d8, *S = (float *)v.S;
  float *xs = (float *)(smem + act_lds_bytes(K, NCOLS));
  float *red = xs + K;
  const int tid = threadIdx.x;
  for (int c = 0; c < NCOLS; ++c) {
    const float *xr = x + (size_t)c * K;
    float ss = 0.f;
    if (tid < 256) {
      for (int i = tid * 4; i < K; i += 256 * 4) {
        const float4 v4 = *(const float4 *)(xr + i);
        *(float4 *)(xs + i) = v4;
        ss = fmaf(v4.x, v4.x, ss); ss = fmaf(v4.y, v4.y, ss); ss = fmaf(v4.z, v4.z, ss); ss = fmaf(v4.w, v4.w, ss);
      }
      ss = wave_sum(ss);
      if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();  // xs and red visible
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);  // same order as block_sum_256 (core_ops.hip)
    // quantize: a thread owns 4 consecutive values, 8 threads own one Q8_1 block (all cross-lane steps are DPP)
    for (int e = tid * 4; e < K; e += NT * 4) {
      const float4 xv = *(const float4 *)(xs + e);
      const float4 wv = *(const float4 *)(nw + e);
      const float v0 = xv.x * inv * wv.x, v1 = xv.y * inv * wv.y, v2 = xv.z * inv * wv.z, v3 = xv.w * inv * wv.w;
      float amax = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
      amax = fmaxf(amax, dpp_mov<0xB1>(amax));
      amax = fmaxf(amax, dpp_mov<0x4E>(amax));
      amax = fmaxf(amax, dpp_mov<0x141>(amax));  // 8 lanes = one 32-value block
      const float d = amax / 127.0f;
      int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
      if (amax != 0.0f) { q0 = (int)roundf(v0 / d); q1 = (int)roundf(v1 / d); q2 = (int)roundf(v2 / d); q3 = (int)roundf(v3 / d); }
      const int piece = e >> 4;
      *(int *)(q + (size_t)c * K + (size_t)swz(piece) * 16 + (e & 15)) = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
      int s = (q0 + q1) + (q2 + q3);
      s += __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, false);
      s += __builtin_amdgcn_update_dpp(0, s, 0x4E, 0xf, 0xf, false);  // 4 lanes = one 16-run
      const float dh = half_bits_to_float(float_to_half_bits(d));
      if ((e & 15) == 0) S[c * runs + piece] = dh * (float)s;
      if ((e & 31) == 0) d8[c * nblk + (e >> 5)] = dh;
    }
    __syncthreads();
  }
  return v;
}

__host__ __device__ inline size_t norm_lds_bytes(int K, int ncols) {
  return act_lds_bytes(K, ncols) + (size_t)K * 4 + 64;
}

// wave-uniform runtime dispatch over the formats the fused path supports (K-quants + Q8_0)
#define MRS_HOT_TYPE_SWITCH(t, ...)                              \
  switch (t) {                                                   \
  case T_Q4_K: { constexpr int TT = T_Q4_K; __VA_ARGS__ } break; \
  case T_Q5_K: { constexpr int TT = T_Q5_K; __VA_ARGS__ } break; \
  case T_Q6_K: { constexpr int TT = T_Q6_K; __VA_ARGS__ } break; \
  case T_Q8_0: { constexpr int TT = T_Q8_0; __VA_ARGS__ } break; \
  default: break;                                                \
  }

__device__ __forceinline__ size_t hot_row_bytes(int t, int K) {
  switch (t) {
  case T_Q4_K: return (size_t)(K / 256) * 144;
  case T_Q5_K: return (size_t)(K / 256) * 176;
  case T_Q6_K: return (size_t)(K / 256) * 210;
  default: return (size_t)(K / 32) * 34;
  }
}

// MRS_DECODE_MIN_WAVES (build-time experiment knob, default off = the measured round-1 object code): the minimum number of waves per SIMD the
// register allocator must leave room for.  4 caps the kernel at 128 VGPRs so that TWO 512-thread workgroups share a CU (16 waves per CU instead
// of 8); pair it with MRS_PROJ_WGS=512 / MRS_GLU_PER=32 / MRS_QKV_PPW at run time.  profiles/round1_decode_kernel_resources.md has the numbers.
#ifdef MRS_DECODE_MIN_WAVES
#define MRS_DECODE_BOUNDS(NT) __launch_bounds__(NT, MRS_DECODE_MIN_WAVES)
#else
#define MRS_DECODE_BOUNDS(NT) __launch_bounds__(NT)
#endif
template <int NCOLS, int PRO, int EPI, int NT>
__global__ void MRS_DECODE_BOUNDS(NT) decode_gemv_kernel(const DecodeGemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NT / 64;
  const int K = a.K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t act_bytes = (PRO == PRO_NORM) ? norm_lds_bytes(K, NCOLS) : act_lds_bytes(K, NCOLS, true);
  float *out_s = (float *)(smem + ((act_bytes + 15) & ~(size_t)15));  // EPI_GLU_Q8_1: [rows_per_wg][NCOLS]
  auto pro = [&]() -> ActLds {
    if constexpr (PRO == PRO_NORM) {
      return stage_norm_q8_1<NCOLS, NT>(smem, a.x, a.norm_w, K, a.eps);
    } else {
      const ActLds act = stage_q8_1<T_Q4_K, NCOLS>(smem, a.y_q8_1, K, a.stride_col_y);
      __syncthreads();
      return act;
    }
  };
  const int total_rows = (EPI == EPI_QKV_ROPE) ? a.nrows[0] + a.nrows[1] + a.nrows[2] : a.nrows[0];
  const int row0 = blockIdx.x * a.rows_per_wg;
  const int row1 = min(row0 + a.rows_per_wg, total_rows);
  const size_t eoff = a.expert_sel ? (size_t)(*a.expert_sel) * a.expert_stride : 0;  // wave-uniform scalar load

  if constexpr (EPI == EPI_GLU_Q8_1) {
    const size_t rb = hot_row_bytes(a.wtype[0], K);
    const int rpw = a.rows_per_wg / NW;
    const int first = row0 + wave * rpw;
    const int nrows = max(0, min(rpw, row1 - first));
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = a.w[0] + eoff + (size_t)r * rb; pB = a.w[1] + eoff + (size_t)r * rb; };
    auto epi = [&](int r, float(&acc)[2][NCOLS]) {
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) out_s[(r - row0) * NCOLS + c] = glu_act(acc[0][c], a.activation) * acc[1][c];
      }
    };
    MRS_HOT_TYPE_SWITCH(a.wtype[0], (stream_rows_auto<TT, NCOLS, true>(first, nrows, 1, K, rpw, rowptr, pro, epi));)
    __syncthreads();
    // quantize this workgroup's rows (a multiple of 32) to Q8_1 blocks: lane <-> row inside a 32-block
    const int nblk = (row1 - row0) / 32;
    for (int i = threadIdx.x; i < nblk * 32 * NCOLS; i += NT) {
      const int c = i / (nblk * 32), e = i % (nblk * 32);
      const float v = out_s[e * NCOLS + c];
      float amax = fabsf(v), sum = v;
#pragma unroll
      for (int m = 16; m > 0; m >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, m, 64)); sum += __shfl_xor(sum, m, 64); }
      const float d = amax / 127.0f;
      uint8_t *blk = a.y_out + ((size_t)c * a.y_out_stride + (row0 + e) / 32) * 36;
      ((int8_t *)(blk + 4))[e & 31] = amax == 0.0f ? (int8_t)0 : (int8_t)roundf(v / d);
      if ((e & 31) == 0) { ((uint16_t *)blk)[0] = float_to_half_bits(d); ((uint16_t *)blk)[1] = float_to_half_bits(sum); }
    }
  } else if constexpr (EPI == EPI_QKV_ROPE) {
    // the host guarantees that a workgroup's rows lie inside ONE of q / k / v (rows_per_wg divides nq and nk)
    int m = 0, base = 0;
    if (row0 >= a.nrows[0] + a.nrows[1]) { m = 2; base = a.nrows[0] + a.nrows[1]; }
    else if (row0 >= a.nrows[0]) { m = 1; base = a.nrows[0]; }
    const size_t rb = hot_row_bytes(a.wtype[m], K);
    const uint8_t *wm = a.w[m];
    const int ppw = a.rows_per_wg / (2 * NW);  // interleaved RoPE pairs (2i, 2i+1) stay in one wave
    const int first = row0 + wave * 2 * ppw;
    const int npairs = max(0, min(ppw, (row1 - first) / 2));
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = wm + (size_t)(r - base) * rb; pB = pA + rb; };
    // epilogue operands are fetched up front (lane i <-> the wave's pair i) so that no dependent global load sits between the
    // last dot product of a pair and its store: rotation (cos, sin) at this token's position and the cache slot
    float pcs[NCOLS], psn[NCOLS];
    {
      const int lr_i = first - base + 2 * min(lane, max(npairs - 1, 0));
      const int pair_i = (lr_i % a.head_dim) >> 1;
      const bool rot = m < 2 && pair_i < a.rot_pairs;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const size_t ti = (size_t)a.positions[c] * a.rot_pairs + (rot ? pair_i : 0);
        const float cs = a.cos_t[ti], sn = a.sin_t[ti];
        pcs[c] = rot ? cs : 1.0f;  // identity rotation for v and for unrotated dims (x*1 - y*0 = x exactly)
        psn[c] = rot ? sn : 0.0f;
      }
    }
    int64_t slots[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) slots[c] = m == 0 ? 0 : a.slot_mapping[c];
    auto epi = [&](int r, float(&acc)[2][NCOLS]) {
      const int pi = (r - first) >> 1;  // wave-uniform
      float cs[NCOLS], sn[NCOLS];
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        cs[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pcs[c]), pi));
        sn[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, psn[c]), pi));
      }
      if (lane == 0) {
        const int lr = r - base;
        const int head = lr / a.head_dim, d = lr % a.head_dim;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          float x, y;
          rope_pair<float>(acc[0][c], acc[1][c], cs[c], sn[c], x, y);
          if (m == 0) {
            a.q_out[(size_t)c * a.nrows[0] + lr] = x;
            a.q_out[(size_t)c * a.nrows[0] + lr + 1] = y;
          } else {
            const int64_t slot = slots[c];
            if (slot >= 0) {
              const int64_t blk = slot / a.block_size, off = slot % a.block_size;
              uint16_t *kc = (uint16_t *)a.k_cache, *vc = (uint16_t *)a.v_cache;
              if (m == 1) {
                const int X = a.cache_x;
                const int64_t o = ((blk * a.num_kv_heads + head) * (a.head_dim / X) + d / X) * a.block_size * X + off * X + d % X;
                kc[o] = float_to_bf16_bits(x);
                kc[o + 1] = float_to_bf16_bits(y);  // d is even and X is even: same 16-byte group
              } else {
                const int64_t o = ((blk * a.num_kv_heads + head) * a.head_dim + d) * a.block_size + off;
                vc[o] = float_to_bf16_bits(x);
                vc[o + a.block_size] = float_to_bf16_bits(y);
              }
            }
          }
        }
      }
    };
    MRS_HOT_TYPE_SWITCH(a.wtype[m], (stream_rows_auto<TT, NCOLS, true>(first, npairs, 2, K, ppw, rowptr, pro, epi));)
  } else {
    const size_t rb = hot_row_bytes(a.wtype[0], K);
    const int rpw = a.rows_per_wg / NW;
    const int first = row0 + wave * rpw;
    const int nrows = max(0, min(rpw, row1 - first));
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = a.w[0] + eoff + (size_t)r * rb; pB = pA + rb; };
    const float ascale = a.acc_scale ? *a.acc_scale : 1.0f;
    // residual values are fetched up front (lane i <-> the wave's row i): no dependent load between a row's reduction and its store
    float hold[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) hold[c] = 0.0f;
    if constexpr (EPI == EPI_RESID_ADD) {
      if (lane < nrows) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) hold[c] = a.out[(size_t)c * a.out_stride + first + lane];
      }
    }
    auto put = [&](int r, const float(&v)[NCOLS]) {  // called by every lane (readlane is wave-wide), lane 0 stores
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        float *o = a.out + (size_t)c * a.out_stride + r;
        if constexpr (EPI == EPI_RESID_ADD) {
          const float old = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hold[c]), r - first));
          if (lane == 0) *o = old * a.resid_scale + v[c] * ascale;  // ascale == 1 outside MoE: bit-identical to the dense path
        } else {
          if (lane == 0) *o = v[c];
        }
      }
    };
    auto epi = [&](int r, float(&acc)[1][NCOLS]) { put(r, acc[0]); };
    auto epi2 = [&](int r, float(&acc)[2][NCOLS]) { put(r, acc[0]); put(r + 1, acc[1]); };
    const bool pair_ok = ((rpw | a.nrows[0]) & 1) == 0;  // the host rounds rows-per-wave up to even for the paired-row formats
    MRS_HOT_TYPE_SWITCH(a.wtype[0],
      if constexpr (PairQ<TT>::value) {
        if (pair_ok) stream_rows_auto<TT, NCOLS, true>(first, nrows / 2, 2, K, rpw / 2, rowptr, pro, epi2);
        else stream_rows_auto<TT, NCOLS, false>(first, nrows, 1, K, rpw, rowptr, pro, epi);
      } else {
        stream_rows_auto<TT, NCOLS, false>(first, nrows, 1, K, rpw, rowptr, pro, epi);
      })
  }
}

static bool hot_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }

template <int PRO, int EPI> struct DecodeLaunch {
  // kernels with the RMSNorm+Q8_1 prologue pay it once per workgroup: fat workgroups (8 waves), about one per CU (measured:
  // 224 x 512-thread workgroups for gate/up beat 448 by 0.14 ms per token); staged-Q8_1 kernels keep 4-wave workgroups
  // fat 8-wave workgroups, about one per CU: every workgroup pays the activation prologue (RMSNorm+Q8_1, or staging the Q8_1
  // blocks) once, and measured decode time falls monotonically from 1024 x 4-wave to 256 x 8-wave workgroups (2.15 -> 2.09 ms)
  static constexpr int NT = 512;
  template <int NCOLS> static int go(DecodeGemvArgs a, hipStream_t s) {
    const int total = (EPI == EPI_QKV_ROPE) ? a.nrows[0] + a.nrows[1] + a.nrows[2] : a.nrows[0];
    constexpr int NW = NT / 64;
    int per;
    if (EPI == EPI_GLU_Q8_1) {
      per = 32 * ((total / 32 + 255) / 256);  // whole Q8_1 output blocks per workgroup, <= 256 workgroups
      { static int ov = -1; if (ov < 0) { const char *e = getenv("MRS_GLU_PER"); ov = e ? atoi(e) : 0; } if (ov > 0) per = ov; }
      if (per < 32) per = 32;
    } else if (EPI == EPI_QKV_ROPE) {
      int ppw = (total / 2 + NW * 192 - 1) / (NW * 192);  // RoPE pairs per wave: <= 192 fat workgroups (measured optimum for 6144 rows)
      if (ppw < 1) ppw = 1;
      if (ppw > 64) ppw = 64;  // epilogue operands are prefetched one pair per lane
      { static int ov = -1; if (ov < 0) { const char *e = getenv("MRS_QKV_PPW"); ov = e ? atoi(e) : 0; } if (ov > 0) ppw = ov; }
      per = 2 * NW * ppw;
      while (per > 2 * NW && (a.nrows[0] % per || a.nrows[1] % per)) per -= 2 * NW;
      if (a.nrows[0] % per || a.nrows[1] % per) return -3;  // a workgroup must not straddle q/k/v
    } else {
      static int tgt = 0; if (!tgt) { const char *e = getenv("MRS_PROJ_WGS"); tgt = e ? atoi(e) : 256; }
      int rpw = (total + NW * tgt - 1) / (NW * tgt);
      if (rpw < 1) rpw = 1;
      if (rpw > 64) rpw = 64;  // epilogue operands are prefetched one row per lane
      if ((a.wtype[0] == T_Q4_K || a.wtype[0] == T_Q5_K) && (total & 1) == 0 && (rpw & 1)) ++rpw;  // paired rows
      per = NW * rpw;
    }
    a.rows_per_wg = per;
    const int grid = (total + per - 1) / per;
    size_t lds = (PRO == PRO_NORM) ? norm_lds_bytes(a.K, NCOLS) : act_lds_bytes(a.K, NCOLS, true);
    lds = (lds + 15) & ~(size_t)15;
    if (EPI == EPI_GLU_Q8_1) lds += (size_t)per * NCOLS * sizeof(float);
    if (lds > 160 * 1024) return -2;
    auto kern = decode_gemv_kernel<NCOLS, PRO, EPI, NT>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
    return 0;
  }
  static int run(const DecodeGemvArgs &a, int b, hipStream_t s) {
    switch (b) {
    case 1: return go<1>(a, s); case 2: return go<2>(a, s); case 3: return go<3>(a, s); case 4: return go<4>(a, s);
    case 5: return go<5>(a, s); case 6: return go<6>(a, s); case 7: return go<7>(a, s); case 8: return go<8>(a, s);
    default: return -1;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// small kernels of the decode step

template <int TYPE> __device__ __forceinline__ void dequant_slice(const uint8_t *row, int s, float *o) {
  const Slice sl = load_slice<TYPE>(row, s);
  int ra, rb;
  slice_runs<TYPE>(s, ra, rb);
  const int8_t *qa = (const int8_t *)&sl.qa, *qb = (const int8_t *)&sl.qb;
#pragma unroll
  for (int j = 0; j < 16; ++j) { o[ra * 16 + j] = sl.sa * (float)qa[j] - sl.oa; o[rb * 16 + j] = sl.sb * (float)qb[j] - sl.ob; }
}

// token ids -> f32 rows of the (quantized) embedding table. grid = tokens
template <int NT>
__global__ void __launch_bounds__(NT) embedding_kernel(const uint8_t *__restrict__ table, int type, const int32_t *__restrict__ ids,
                                                       float *__restrict__ out, int K) {
  const int tok = blockIdx.x;
  float *o = out + (size_t)tok * K;
  const int64_t id = ids[tok];
  if (type == 0) { const float *src = (const float *)table + id * K; for (int i = threadIdx.x; i < K; i += NT) o[i] = src[i]; return; }
  if (type == 1) { const uint16_t *src = (const uint16_t *)table + id * K; for (int i = threadIdx.x; i < K; i += NT) o[i] = half_bits_to_float(src[i]); return; }
  if (type == 30) { const uint16_t *src = (const uint16_t *)table + id * K; for (int i = threadIdx.x; i < K; i += NT) o[i] = bf16_bits_to_float(src[i]); return; }
  const uint8_t *row = table + (size_t)id * hot_row_bytes(type, K);
  for (int s = threadIdx.x; s < K / 32; s += NT) {
    MRS_HOT_TYPE_SWITCH(type, dequant_slice<TT>(row, s, o);)
  }
}

// greedy sampling: argmax over [b][vocab] logits (first max wins, like candle's argmax), then advance the
// device-resident decode state so the next graph replay needs no host work:
//   next_ids[c] = argmax ; tokens_out[c][step] = argmax ; positions[c]++ ; context_lens[c]++ ;
//   slot_mapping[c] = block_table[c][pos / bs] * bs + pos % bs
struct SampleArgs {
  const float *logits; int vocab; int b;
  int32_t *next_ids; int32_t *tokens_out; int tokens_out_stride; int32_t *step_counter;
  int32_t *positions; uint32_t *context_lens; int64_t *slot_mapping; const uint32_t *block_tables; int max_blocks; int block_size;
  unsigned long long *scratch;  // [b] packed (value, index) maxima, zeroed by the launcher
};

__device__ __forceinline__ unsigned long long pack_max(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map
  return ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - idx);  // ties -> smaller index
}

// round 5: 16-byte loads, one contiguous slice per workgroup (the round-4 kernel walked the row with 4-byte strided loads from 128 workgroups: 8 us for 513 KB);
// same packed key, same winner (largest value, lowest index), still one atomicMax per wave
__global__ void __launch_bounds__(256) argmax_partial_kernel(const SampleArgs a) {
  const int c = blockIdx.y;
  const float *l = a.logits + (size_t)c * a.vocab;
  const int n4 = a.vocab >> 2;
  const int per = (n4 + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(i0 + per, n4);
  unsigned long long best = 0;
  if ((((size_t)l) & 15) == 0) {
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
      const float4 v = *(const float4 *)(l + 4 * (size_t)i);
      unsigned long long p = pack_max(v.x, 4 * i);
      best = p > best ? p : best;
      p = pack_max(v.y, 4 * i + 1); best = p > best ? p : best;
      p = pack_max(v.z, 4 * i + 2); best = p > best ? p : best;
      p = pack_max(v.w, 4 * i + 3); best = p > best ? p : best;
    }
  } else {
    for (int i = 4 * i0 + threadIdx.x; i < 4 * i1; i += 256) { const unsigned long long p = pack_max(l[i], i); best = p > best ? p : best; }
  }
  if (blockIdx.x == gridDim.x - 1)
    for (int i = 4 * n4 + threadIdx.x; i < a.vocab; i += 256) { const unsigned long long p = pack_max(l[i], i); best = p > best ? p : best; }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) { const unsigned long long o = __shfl_xor(best, m, 64); best = o > best ? o : best; }
  __shared__ unsigned long long wbest[4];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic per workgroup (~12 ns each on one address)
    unsigned long long b2 = wbest[0];
    for (int w = 1; w < 4; ++w) b2 = wbest[w] > b2 ? wbest[w] : b2;
    atomicMax(a.scratch + c, b2);
  }
}

__global__ void __launch_bounds__(64) sample_advance_kernel(const SampleArgs a) {
  const int c = threadIdx.x;
  if (c >= a.b) return;
  const int idx = 0x7fffffff - (int)(a.scratch[c] & 0xffffffffu);
  a.scratch[c] = 0;
  a.next_ids[c] = idx;
  const int step = *a.step_counter;
  // a graph replayed past the buffers it was captured for must not write outside them: tokens beyond tokens_out are dropped, and a position
  // beyond the block table gets the pad slot (-1: reshape_and_cache / the fused epilogues skip it) instead of another sequence's page
  if (a.tokens_out && step < a.tokens_out_stride) a.tokens_out[(size_t)c * a.tokens_out_stride + step] = idx;
  const int pos = a.positions[c] + 1;
  a.positions[c] = pos;
  a.context_lens[c] = (uint32_t)pos + 1;
  const int blk = pos / a.block_size;
  a.slot_mapping[c] = blk < a.max_blocks ? (int64_t)a.block_tables[(size_t)c * a.max_blocks + blk] * a.block_size + pos % a.block_size : (int64_t)-1;
  __syncthreads();
  if (c == 0) *a.step_counter = step + 1;
}

// round 6, batch 1: sample_advance_kernel + the NEXT step's embedding gather in one launch (the captured decode graph then opens with layer 0's qkv: the step is two
// launches shorter -- argmax_partial_kernel lives in lm_head's epilogue, mrs_dec_proj_argmax, and embedding_kernel here).  a.scratch[0] holds the packed maximum.
struct EmbedNext { const uint8_t *table; int type; float *h; int K; };
__global__ void __launch_bounds__(256) sample_advance_embed_kernel(const SampleArgs a, const EmbedNext e) {
  __shared__ int s_idx;
  if (threadIdx.x == 0) {
    const int idx = 0x7fffffff - (int)(a.scratch[0] & 0xffffffffu);
    a.scratch[0] = 0;
    a.next_ids[0] = idx;
    const int step = *a.step_counter;
    if (a.tokens_out && step < a.tokens_out_stride) a.tokens_out[step] = idx;
    const int pos = a.positions[0] + 1;
    a.positions[0] = pos;
    a.context_lens[0] = (uint32_t)pos + 1;
    const int blk = pos / a.block_size;
    a.slot_mapping[0] = blk < a.max_blocks ? (int64_t)a.block_tables[blk] * a.block_size + pos % a.block_size : (int64_t)-1;
    *a.step_counter = step + 1;
    s_idx = idx;
  }
  __syncthreads();
  const int64_t id = s_idx;
  const int K = e.K;
  float *o = e.h;
  if (e.type == 0) { const float *src = (const float *)e.table + id * K; for (int i = threadIdx.x; i < K; i += 256) o[i] = src[i]; return; }
  if (e.type == 1) { const uint16_t *src = (const uint16_t *)e.table + id * K; for (int i = threadIdx.x; i < K; i += 256) o[i] = half_bits_to_float(src[i]); return; }
  if (e.type == 30) { const uint16_t *src = (const uint16_t *)e.table + id * K; for (int i = threadIdx.x; i < K; i += 256) o[i] = bf16_bits_to_float(src[i]); return; }
  const uint8_t *row = e.table + (size_t)id * hot_row_bytes(e.type, K);
  for (int s = threadIdx.x; s < K / 32; s += 256) {
    MRS_HOT_TYPE_SWITCH(e.type, dequant_slice<TT>(row, s, o);)
  }
}

// f32 [rows][K] -> Q8_1 blocks, same bytes as launch_mmvq_gguf_quantize_q8_1_f32 (used after attention)
__global__ void __launch_bounds__(256) quantize_rows_kernel(const float *__restrict__ x, uint8_t *__restrict__ y, int K, int stride_blocks) {
  const int c = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= stride_blocks * 32) return;
  const float v = e < K ? x[(size_t)c * K + e] : 0.0f;
  float amax = fabsf(v), sum = v;
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, m, 64)); sum += __shfl_xor(sum, m, 64); }
  const float d = amax / 127.0f;
  uint8_t *blk = y + ((size_t)c * stride_blocks + e / 32) * 36;
  ((int8_t *)(blk + 4))[e & 31] = amax == 0.0f ? (int8_t)0 : (int8_t)roundf(v / d);
  if ((e & 31) == 0) { ((uint16_t *)blk)[0] = float_to_half_bits(d); ((uint16_t *)blk)[1] = float_to_half_bits(sum); }
}

}  // namespace mrs

using namespace mrs;

// ---- C entry points (declared in include/mrs_hip_ext.h) -----------------------------------------
extern "C" int mrs_decode_gemv_supported(int ggml_type) { return hot_type(ggml_type) ? 1 : 0; }

// QKV: h [b][K] f32 --RMSNorm*norm_w--> Q8_1 (LDS) --> q (f32, RoPE'd), k (RoPE'd) / v -> paged bf16 cache
extern "C" int mrs_decode_qkv(const void *wq, const void *wk, const void *wv, int tq, int tk, int tv, int nq, int nk, int nv, int K,
                              const float *h, const float *norm_w, float eps, float *q_out, void *k_cache, void *v_cache,
                              const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t,
                              int head_dim, int rot_pairs, int num_kv_heads, int block_size, int b, void *stream) {
  if (!hot_type(tq) || !hot_type(tk) || !hot_type(tv) || (nq | nk | nv | head_dim) & 1) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)wq; a.w[1] = (const uint8_t *)wk; a.w[2] = (const uint8_t *)wv;
  a.wtype[0] = tq; a.wtype[1] = tk; a.wtype[2] = tv; a.nrows[0] = nq; a.nrows[1] = nk; a.nrows[2] = nv; a.K = K;
  a.x = h; a.norm_w = norm_w; a.eps = eps; a.q_out = q_out; a.k_cache = k_cache; a.v_cache = v_cache;
  a.slot_mapping = slot_mapping; a.positions = positions; a.cos_t = cos_t; a.sin_t = sin_t; a.head_dim = head_dim;
  a.rot_pairs = rot_pairs; a.num_kv_heads = num_kv_heads; a.block_size = block_size; a.cache_x = 8;
  return DecodeLaunch<PRO_NORM, EPI_QKV_ROPE>::run(a, b, (hipStream_t)stream);
}

// gate/up: h --RMSNorm--> Q8_1 (LDS) --> act(gate.x)*up.x --> Q8_1 blocks y_out [b][y_out_stride]
extern "C" int mrs_decode_gate_up(const void *wg, const void *wu, int type, int n, int K, const float *h, const float *norm_w,
                                  float eps, int activation, void *y_out, int y_out_stride, int b, void *stream) {
  if (!hot_type(type) || n % 32) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)wg; a.w[1] = (const uint8_t *)wu; a.wtype[0] = a.wtype[1] = type; a.nrows[0] = n; a.K = K;
  a.x = h; a.norm_w = norm_w; a.eps = eps; a.activation = activation; a.y_out = (uint8_t *)y_out; a.y_out_stride = y_out_stride;
  return DecodeLaunch<PRO_NORM, EPI_GLU_Q8_1>::run(a, b, (hipStream_t)stream);
}

// row-parallel projections (o_proj, down): y Q8_1 --> out[c*out_stride + r] (+)= W.y
extern "C" int mrs_decode_proj(const void *w, int type, int n, int K, const void *y_q8_1, int stride_col_y, float *out,
                               int out_stride, int accumulate, int b, void *stream) {
  if (!hot_type(type)) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)w; a.wtype[0] = type; a.nrows[0] = n; a.K = K; a.y_q8_1 = (const uint8_t *)y_q8_1;
  a.stride_col_y = stride_col_y; a.out = out; a.out_stride = out_stride; a.resid_scale = 1.0f;
  return accumulate ? DecodeLaunch<PRO_Q8_1, EPI_RESID_ADD>::run(a, b, (hipStream_t)stream)
                    : DecodeLaunch<PRO_Q8_1, EPI_STORE>::run(a, b, (hipStream_t)stream);
}

// same with out = out * resid_scale + W.y (tensor parallelism: resid_scale = 1 / world_size before the sum all-reduce of out)
extern "C" int mrs_decode_proj_scaled(const void *w, int type, int n, int K, const void *y_q8_1, int stride_col_y, float *out,
                                      int out_stride, float resid_scale, int b, void *stream) {
  if (!hot_type(type)) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)w; a.wtype[0] = type; a.nrows[0] = n; a.K = K; a.y_q8_1 = (const uint8_t *)y_q8_1;
  a.stride_col_y = stride_col_y; a.out = out; a.out_stride = out_stride; a.resid_scale = resid_scale;
  return DecodeLaunch<PRO_Q8_1, EPI_RESID_ADD>::run(a, b, (hipStream_t)stream);
}

// ---- mixture of experts, decode (b = 1 token per launch; the reference's indexed GEMV: kernels/indexed_moe/indexed_moe.cu:892-1150,
//      moe_gemv_fused_gate_up / moe_gemv_down_aggregate) -- same GEMV kernels, expert chosen on the device
extern "C" int mrs_moe_decode_gate_up(const void *wg, const void *wu, size_t expert_stride_bytes, const int32_t *expert_sel, int type, int n, int K,
                                      const float *h, const float *norm_w, float eps, int activation, void *y_out, int y_out_stride, void *stream) {
  if (!hot_type(type) || n % 32 || !expert_sel) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)wg; a.w[1] = (const uint8_t *)wu; a.wtype[0] = a.wtype[1] = type; a.nrows[0] = n; a.K = K;
  a.x = h; a.norm_w = norm_w; a.eps = eps; a.activation = activation; a.y_out = (uint8_t *)y_out; a.y_out_stride = y_out_stride;
  a.expert_sel = expert_sel; a.expert_stride = expert_stride_bytes;
  return DecodeLaunch<PRO_NORM, EPI_GLU_Q8_1>::run(a, 1, (hipStream_t)stream);
}
// out += topk_weight * (W_down[expert] . y)
extern "C" int mrs_moe_decode_down(const void *w, size_t expert_stride_bytes, const int32_t *expert_sel, const float *topk_weight, int type, int n, int K,
                                   const void *y_q8_1, int stride_col_y, float *out, void *stream) {
  if (!hot_type(type) || !expert_sel) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)w; a.wtype[0] = type; a.nrows[0] = n; a.K = K; a.y_q8_1 = (const uint8_t *)y_q8_1; a.stride_col_y = stride_col_y;
  a.out = out; a.out_stride = n; a.resid_scale = 1.0f; a.expert_sel = expert_sel; a.expert_stride = expert_stride_bytes; a.acc_scale = topk_weight;
  return DecodeLaunch<PRO_Q8_1, EPI_RESID_ADD>::run(a, 1, (hipStream_t)stream);
}

namespace mrs {
// router: logits = gate_w [E][K] (f32) . x [K]; softmax over all experts -> top-k -> renormalise (moe_router_topk, ops.rs:259-336,
// Mixtral settings: models/mixtral.rs:286-300).  One workgroup per token, wave e computes logit e (E <= 16 waves ... loops otherwise).
// One wave per expert (up to 16 waves), the loop unrolled so that a lane's loads of a row are all in flight before its first fma: the 4-wave version
// with one dependent load pair per iteration took 17-18 us per decode token (16 serialized memory round trips), 10 % of a Mixtral step.  The per-lane
// summation order (element order, then the wave sum) is unchanged.
// NORM: x is the un-normed hidden state; the first 256 threads compute RmsNorm(x) * norm_w into LDS with the arithmetic of rms_norm_kernel<float, 0, true>
// (core_ops.hip: float4 v = tid + 256 j, fmaf chain, wave sums, red[0..3] in order, x * inv * w) -- same values, one launch less per MoE layer and token.
template <bool NORM>
__global__ void __launch_bounds__(1024) moe_router_kernel(const float *__restrict__ x, const float *__restrict__ gate_w, int E, int K, int top_k,
                                                          int renormalize, int32_t *__restrict__ ids, float *__restrict__ weights, float *__restrict__ logits_out,
                                                          const float *__restrict__ norm_w, float eps) {
  __shared__ float lg[512];
  __shared__ float red[4];
  extern __shared__ __attribute__((aligned(16))) char xn_s[];
  const int tok = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const float *xr = x + (size_t)tok * K;
  if constexpr (NORM) {
    float *xn = (float *)xn_s;
    const int tid = threadIdx.x, nvec = K / 4;
    float sum = 0.f;
    if (tid < 256)
      for (int v = tid; v < nvec; v += 256) {
        const float4 t = *(const float4 *)(xr + 4 * v);
        sum = fmaf(t.x, t.x, sum); sum = fmaf(t.y, t.y, sum); sum = fmaf(t.z, t.z, sum); sum = fmaf(t.w, t.w, sum);
      }
    sum = wave_sum(sum);
    if (tid < 256 && lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
    for (int v = tid; v < nvec; v += blockDim.x) {
      const float4 t = *(const float4 *)(xr + 4 * v), w = *(const float4 *)(norm_w + 4 * v);
      *(float4 *)(xn + 4 * v) = make_float4(t.x * inv * w.x, t.y * inv * w.y, t.z * inv * w.z, t.w * inv * w.w);
    }
    __syncthreads();
    xr = xn;
  }
  for (int e = wave; e < E; e += nwaves) {
    float s = 0.f;
    const float *wr = gate_w + (size_t)e * K;
#pragma unroll 8
    for (int i = lane * 4; i < K; i += 256) {
      const float4 a4 = *(const float4 *)(xr + i), w4 = *(const float4 *)(wr + i);
      s = fmaf(a4.x, w4.x, s); s = fmaf(a4.y, w4.y, s); s = fmaf(a4.z, w4.z, s); s = fmaf(a4.w, w4.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) lg[e] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // E is tiny (8 for Mixtral): serial softmax / top-k, ties -> lower index (candle topk order)
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, lg[e]);
    float den = 0.f;
    for (int e = 0; e < E; ++e) den += expf(lg[e] - mx);
    float sel = 0.f;
    unsigned long long used = 0;
    for (int k = 0; k < top_k; ++k) {
      int best = -1; float bv = -INFINITY;
      for (int e = 0; e < E; ++e) if (!((used >> e) & 1ull) && lg[e] > bv) { bv = lg[e]; best = e; }
      if (best < 0) {  // every remaining logit is NaN or -inf (no comparison was true): take the lowest unused expert instead of shifting by -1 /
                       // storing expert -1 (advisor, rounds 1-2); its weight is NaN / 0 as the arithmetic gives it
        for (int e = 0; e < E && best < 0; ++e) if (!((used >> e) & 1ull)) best = e;
        bv = lg[best];
      }
      used |= 1ull << best;
      const float p = expf(bv - mx) / den;
      ids[(size_t)tok * top_k + k] = best;
      weights[(size_t)tok * top_k + k] = p;
      sel += p;
    }
    if (renormalize) for (int k = 0; k < top_k; ++k) weights[(size_t)tok * top_k + k] /= sel;
    if (logits_out) for (int e = 0; e < E; ++e) logits_out[(size_t)tok * E + e] = lg[e];
  }
}
// Round 6: the same router as E workgroups per token (one expert's logit each) + the last arriver's softmax / top-k.  moe_router_kernel<true> computes all E logits in ONE
// workgroup: 8 x 16 KB of router rows through one CU's memory pipe = most of its 9.4 us (12 % of a Mixtral decode layer).  Here every workgroup repeats the RmsNorm (the NORM
// variant's arithmetic: same thread <-> element mapping, same reduction order), wave 0 computes the logit of expert blockIdx.x with the one-wave-per-expert chain of the kernel
// above, publishes it write-through and draws a ticket; the workgroup that draws the last one reads the E logits past L1 and runs the serial softmax / top-k code unchanged:
// same ids, same weights, bit for bit (tests/test_moe.py).  scratch: per token E floats + one u32 ticket (zero at rest; the kernel leaves it zero).
#ifndef MRS_WAIT_VMCNT0
#define MRS_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
__global__ void __launch_bounds__(256) moe_router_split_kernel(const float *__restrict__ x, const float *__restrict__ gate_w, int E, int K, int top_k, int renormalize,
                                                               int32_t *__restrict__ ids, float *__restrict__ weights, const float *__restrict__ norm_w, float eps,
                                                               float *__restrict__ scratch) {
  __shared__ float lg[64];
  __shared__ float red[4];
  __shared__ int last_s;
  extern __shared__ __attribute__((aligned(16))) char xn_s[];
  const int e = blockIdx.x, tok = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float *xr = x + (size_t)tok * K;
  float *xn = (float *)xn_s;
  const int nvec = K / 4;
  float sum = 0.f;
  for (int v = tid; v < nvec; v += 256) {
    const float4 t = *(const float4 *)(xr + 4 * v);
    sum = fmaf(t.x, t.x, sum); sum = fmaf(t.y, t.y, sum); sum = fmaf(t.z, t.z, sum); sum = fmaf(t.w, t.w, sum);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
  for (int v = tid; v < nvec; v += 256) {
    const float4 t = *(const float4 *)(xr + 4 * v), w = *(const float4 *)(norm_w + 4 * v);
    *(float4 *)(xn + 4 * v) = make_float4(t.x * inv * w.x, t.y * inv * w.y, t.z * inv * w.z, t.w * inv * w.w);
  }
  __syncthreads();
  float *slog = scratch + (size_t)tok * (E + 1);
  unsigned *ticket = (unsigned *)(slog + E);
  if (wave == 0) {
    float s = 0.f;
    const float *wr = gate_w + (size_t)e * K;
#pragma unroll 8
    for (int i = lane * 4; i < K; i += 256) {
      const float4 a4 = *(const float4 *)(xn + i), w4 = *(const float4 *)(wr + i);
      s = fmaf(a4.x, w4.x, s); s = fmaf(a4.y, w4.y, s); s = fmaf(a4.z, w4.z, s); s = fmaf(a4.w, w4.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) {
      __hip_atomic_store(slog + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
      MRS_WAIT_VMCNT0();
      last_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(E - 1);
      if (last_s) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last_s) return;
  if (tid < E) lg[tid] = __hip_atomic_load(slog + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: past L1
  __syncthreads();
  if (tid == 0) {  // moe_router_kernel's serial softmax / top-k, unchanged
    float mx = -INFINITY;
    for (int j = 0; j < E; ++j) mx = fmaxf(mx, lg[j]);
    float den = 0.f;
    for (int j = 0; j < E; ++j) den += expf(lg[j] - mx);
    float sel = 0.f;
    unsigned long long used = 0;
    for (int k = 0; k < top_k; ++k) {
      int best = -1; float bv = -INFINITY;
      for (int j = 0; j < E; ++j) if (!((used >> j) & 1ull) && lg[j] > bv) { bv = lg[j]; best = j; }
      if (best < 0) {
        for (int j = 0; j < E && best < 0; ++j) if (!((used >> j) & 1ull)) best = j;
        bv = lg[best];
      }
      used |= 1ull << best;
      const float p = expf(bv - mx) / den;
      ids[(size_t)tok * top_k + k] = best;
      weights[(size_t)tok * top_k + k] = p;
      sel += p;
    }
    if (renormalize) for (int k = 0; k < top_k; ++k) weights[(size_t)tok * top_k + k] /= sel;
  }
}
}  // namespace mrs
extern "C" int mrs_moe_router_topk(const float *x, const float *gate_w, int tokens, int n_experts, int K, int top_k, int renormalize, int32_t *ids,
                                   float *weights, float *logits_out, void *stream) {
  if (tokens <= 0) return 0;
  if (n_experts < 1 || n_experts > 64 || top_k < 1 || top_k > n_experts || (K & 3)) return -1;
  const int waves = n_experts < 4 ? 4 : (n_experts > 16 ? 16 : n_experts);
  hipLaunchKernelGGL(mrs::moe_router_kernel<false>, dim3(tokens), dim3(64 * waves), 0, (hipStream_t)stream, x, gate_w, n_experts, K, top_k, renormalize, ids, weights,
                     logits_out, (const float *)nullptr, 0.f);
  return 0;
}
// the same on the un-normed hidden state: RmsNorm(h) * norm_w (the values mrs_rms_norm_f32 writes) computed inside the router's workgroup; -3: row too long for LDS
extern "C" int mrs_moe_router_topk_norm(const float *h, const float *norm_w, float eps, const float *gate_w, int tokens, int n_experts, int K, int top_k, int renormalize,
                                        int32_t *ids, float *weights, void *stream) {
  if (tokens <= 0) return 0;
  if (n_experts < 1 || n_experts > 64 || top_k < 1 || top_k > n_experts || (K & 3) || !norm_w) return -1;
  if ((size_t)K * 4 > 64 * 1024) return -3;
  const int waves = n_experts < 4 ? 4 : (n_experts > 16 ? 16 : n_experts);
  hipLaunchKernelGGL(mrs::moe_router_kernel<true>, dim3(tokens), dim3(64 * waves), (size_t)K * 4, (hipStream_t)stream, h, gate_w, n_experts, K, top_k, renormalize, ids,
                     weights, (float *)nullptr, norm_w, eps);
  return 0;
}

// scratch: tokens * (n_experts + 1) * 4 bytes, zero before the first call (the tickets return to zero); -3: row too long for LDS (callers fall back to mrs_moe_router_topk_norm)
extern "C" size_t mrs_moe_router_split_scratch_bytes(int tokens, int n_experts) { return (size_t)tokens * (size_t)(n_experts + 1) * 4; }
extern "C" int mrs_moe_router_topk_norm_split(const float *h, const float *norm_w, float eps, const float *gate_w, int tokens, int n_experts, int K, int top_k, int renormalize,
                                              int32_t *ids, float *weights, void *scratch, void *stream) {
  if (tokens <= 0) return 0;
  if (n_experts < 1 || n_experts > 64 || top_k < 1 || top_k > n_experts || (K & 3) || !norm_w || !scratch) return -1;
  if ((size_t)K * 4 > 64 * 1024) return -3;
  hipLaunchKernelGGL(mrs::moe_router_split_kernel, dim3(n_experts, tokens), dim3(256), (size_t)K * 4, (hipStream_t)stream, h, gate_w, n_experts, K, top_k, renormalize, ids, weights,
                     norm_w, eps, (float *)scratch);
  return 0;
}

namespace mrs {
__global__ void __launch_bounds__(256) vec_add_kernel(float *__restrict__ a, const float *__restrict__ b, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    if (i + 4 <= n) { float4 x = *(float4 *)(a + i); const float4 y = *(const float4 *)(b + i); x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; *(float4 *)(a + i) = x; }
    else for (size_t j = i; j < n; ++j) a[j] += b[j];
  }
}
}  // namespace mrs
// a += b (residual add after a row-parallel all-reduce in the prefill path)
extern "C" int mrs_vec_add_f32(float *a, const float *b, size_t n, void *stream) {
  if (!n) return 0;
  size_t g = (n / 4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  hipLaunchKernelGGL(mrs::vec_add_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, n);
  return 0;
}

// final norm + lm_head: h --RMSNorm--> Q8_1 (LDS) --> logits f32 [b][n]
extern "C" int mrs_decode_norm_proj(const void *w, int type, int n, int K, const float *h, const float *norm_w, float eps,
                                    float *out, int out_stride, int b, void *stream) {
  if (!hot_type(type)) return -1;
  DecodeGemvArgs a{};
  a.w[0] = (const uint8_t *)w; a.wtype[0] = type; a.nrows[0] = n; a.K = K; a.x = h; a.norm_w = norm_w; a.eps = eps;
  a.out = out; a.out_stride = out_stride;
  return DecodeLaunch<PRO_NORM, EPI_STORE>::run(a, b, (hipStream_t)stream);
}

extern "C" int mrs_embedding(const void *table, int type, const int32_t *ids, float *out, int K, int tokens, void *stream) {
  if (!(hot_type(type) || type == 0 || type == 1 || type == 30) || tokens <= 0) return -1;
  hipLaunchKernelGGL((embedding_kernel<256>), dim3(tokens), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)table, type, ids, out, K);
  return 0;
}

extern "C" int mrs_quantize_rows_q8_1(const float *x, void *y, int K, int stride_blocks, int rows, void *stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((stride_blocks * 32 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, x, (uint8_t *)y, K, stride_blocks);
  return 0;
}

// batch 1, the maximum already in scratch[0] (mrs_dec_proj_argmax): next id, state advance and h_next = embedding row of the next id (mrs_embedding's values)
extern "C" int mrs_sample_advance_embed(int32_t *next_ids, int32_t *tokens_out, int tokens_out_stride, int32_t *step_counter, int32_t *positions, uint32_t *context_lens,
                                        int64_t *slot_mapping, const uint32_t *block_tables, int max_blocks, int block_size, void *scratch, const void *table, int type,
                                        float *h_next, int K, void *stream) {
  if (!scratch || !table || !h_next || K <= 0 || !(type == 0 || type == 1 || type == 30 || hot_type(type))) return -1;
  SampleArgs a{nullptr, 0, 1, next_ids, tokens_out, tokens_out_stride, step_counter, positions, context_lens, slot_mapping, block_tables, max_blocks, block_size,
               (unsigned long long *)scratch};
  EmbedNext e{(const uint8_t *)table, type, h_next, K};
  hipLaunchKernelGGL(sample_advance_embed_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, e);
  return 0;
}

extern "C" int mrs_sample_greedy_advance(const float *logits, int vocab, int b, int32_t *next_ids, int32_t *tokens_out,
                                         int tokens_out_stride, int32_t *step_counter, int32_t *positions,
                                         uint32_t *context_lens, int64_t *slot_mapping, const uint32_t *block_tables,
                                         int max_blocks, int block_size, void *scratch, void *stream) {
  if (b <= 0 || b > 64) return -1;
  SampleArgs a{logits, vocab, b, next_ids, tokens_out, tokens_out_stride, step_counter, positions, context_lens, slot_mapping,
               block_tables, max_blocks, block_size, (unsigned long long *)scratch};
  int gx = (vocab + 2047) / 2048; if (gx > 256) gx = 256; if (gx < 1) gx = 1;  // two float4 per thread
  hipLaunchKernelGGL(argmax_partial_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(sample_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return 0;
}
