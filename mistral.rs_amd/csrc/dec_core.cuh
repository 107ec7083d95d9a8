// dec_core.cuh -- MI355X decode engine ("dec"): the GEMV core of the batch<=8 decode step in the arithmetic of the reference CPU path.
//
// What the reference does on this path: GgufMatMul::forward_raw -> candle QMatMul::forward with f32 activations
// (mistralrs-quant/src/gguf/mod.rs:465-478): every activation row is quantized to the vec_dot partner of the weight format -- Q8_K
// (one f32 scale per 256 values, int8 quants, per-16 sums) for the K-quants, Q8_0 (f16 scale per 32) for Q8_0 -- and every output is
// sum over blocks of (d_w * d_x) * <integer dot> (- (dmin * d_x) * <integer min term>).  north_star pins THIS arithmetic (logits within
// 1e-3 of the CPU path, same greedy ids), so the engine computes exactly these integers and combines them in f32; only the f32 summation
// order over the superblocks of a row differs from a sequential CPU loop (oracle: oracle/ggml_oracle.c dot_kquant_q8K / dot_legacy_q8).
//
// MI355X design:
//   * weights live in a DECODE LAYOUT made once at load time (mrs_dec_repack): per tensor four planes -- quant bytes, extra bits, 8-bit
//     pre-decoded sub-block scales, f16 super-scales -- each row-major, so a wave's 64 lanes read 64 consecutive 16-byte pieces (1 KiB per
//     instruction, 16-byte aligned also for Q6_K / Q8_0 whose GGUF blocks are only 2-byte aligned) and no lane decodes 6-bit scales;
//   * a wave owns whole rows; lane l takes unit t*64+l of a row (unit = 32 weights for Q4_K / Q5_K, 64 for Q6_K, 16 for Q8_0);
//   * loads are raw buffer loads (one descriptor per tensor): positions past the end of the wave's work or of a ragged row are sent out of
//     range, return zeros and cost no memory traffic, so the prefetch ring needs no branches (hipcc keeps exact vmcnt waits) and ragged
//     tails need no masks;
//   * a ring of DEPTH tiles per wave is in flight before the activation prologue starts and stays full across rows and tensors;
//   * activations: int8 in LDS in an XOR piece swizzle that is bank-conflict-free for the three access patterns, f32 block scales, int32
//     per-16 sums; integer dots with v_dot4_i32_i8, integer scale / min combination, two f32 FMAs per unit, DPP wave reduction per row.
#pragma once
#include "gguf_blocks.cuh"
#include <type_traits>

#ifndef MRS_WAVE_SYNC
#define MRS_WAVE_SYNC() __builtin_amdgcn_wave_barrier() /* lanes of a wave exchange through LDS in lockstep; the host emulation maps this to a fiber sync */
#endif

namespace mrs {
namespace dec {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#ifndef MRS_DEC_NT
#define MRS_DEC_NT 512
#endif
constexpr int NT = MRS_DEC_NT, NW = NT / 64;  // threads / waves per workgroup (one workgroup per CU): 512 or 1024
constexpr int ACT_STRIDE = NT * 4;           // values between a thread's consecutive float4 pieces of the activation row
constexpr unsigned OOB = 0xFFFFFF00u;  // buffer offset that is out of range for every tensor: the load returns zeros

// ------------------------------------------------------------------------------------------------ decode layout
// Plane offsets (bytes) inside one tensor's repacked buffer.  S = K / 256 superblocks per row.
//   Q4_K: q  [n][S][128] nibbles as in the GGUF block            hs [n][S][4][sc(2c) sc(2c+1) m(2c) m(2c+1)]   hd [n][S][d dmin] f16
//   Q5_K: q  as Q4_K; x [n][S][8 slices][u32 hi-bits, see repack] hs, hd as Q4_K
//   Q6_K: q  [n][S][128] = ql; x [n][S][4 units][16 B 2-bit fields]; hs [n][S][4 units][4 x int8 scale]; hd [n][S] f16 d
//   Q8_0: q  [n][K] int8;                                         hd [n][K/32] f16 d
struct Planes { size_t q, x, hs, hd, total; };
__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
__host__ __device__ inline Planes plane_layout(int type, long long n, long long k) {
  Planes p{};
  const size_t S = (size_t)(k / 256), N = (size_t)n;
  size_t o = 0;
  switch (type) {
  case T_Q4_K: o = align256(N * S * 128); p.hs = o; o += align256(N * S * 16); p.hd = o; o += align256(N * S * 4); break;
  case T_Q5_K: o = align256(N * S * 128); p.x = o; o += align256(N * S * 32); p.hs = o; o += align256(N * S * 16); p.hd = o; o += align256(N * S * 4); break;
  case T_Q6_K: o = align256(N * S * 128); p.x = o; o += align256(N * S * 64); p.hs = o; o += align256(N * S * 16); p.hd = o; o += align256(N * S * 2); break;
  case T_Q8_0: o = align256(N * (size_t)k); p.hd = o; o += align256(N * (size_t)(k / 32) * 2); break;
  default: break;
  }
  p.total = o;
  return p;
}
__host__ __device__ inline bool dec_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }

// one tensor in decode layout, as the kernels see it
struct Mat {
  const uint8_t *base;
  unsigned off_x, off_hs, off_hd, bytes;  // the q plane starts at 0
  int type, n, k;
};

// ------------------------------------------------------------------------------------------------ activations in LDS
// per column:  q [K] int8 (16-byte pieces, piece p of superblock sb stored at p ^ m(sb));  d [K/32] f32 (Q8_K mode: entry sb = d of the
// superblock; Q8_0 mode: entry b = f32(f16(d)) of block b);  bs [K/16] int32 sums of each 16-run (Q8_K mode)
// m(sb) = (sb & 1) * 3 | ((sb >> 1) & 1) * 4: the 16 lanes of one ds_read_b128 group then touch 16 different 16-byte bank groups for
//   Q4_K / Q5_K (lane -> run sb*16 + 4c + hp [+2]),  Q6_K (lane -> runs sb*16 + 8h + 2j [+1, +4, +5])  and  Q8_0 (lane -> piece).
enum : int { ACT_Q8K = 0, ACT_Q80 = 1 };
__host__ __device__ inline int act_mode_for(int type) { return type == T_Q8_0 ? ACT_Q80 : ACT_Q8K; }
__host__ __device__ inline size_t act_bytes(int K, int ncols) { return (size_t)ncols * ((size_t)K + (size_t)(K / 32) * 4 + (size_t)(K / 16) * 4); }
struct Act {
  const char *q;    // [ncols][K]
  const float *d;   // [ncols][K/32]
  const int *bs;    // [ncols][K/16]
  int K;
};
__device__ __forceinline__ int sb_mask(int sb) { return (sb & 1) * 3 | ((sb >> 1) & 1) * 4; }
__device__ __forceinline__ int swz_piece(int p) { return p ^ sb_mask(p >> 4); }

template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

// f32 activations [NCOLS][ldx] (optionally RMSNorm(x) * w first: RmsNorm::forward, mistralrs-core/src/layers.rs:403-414) -> the LDS image
// above.  Quantizers: candle BlockQ8K::from_float (amax with its sign, iscale = -128 / max, q = min(127, round(iscale * x)), d = 1 / iscale,
// first maximum wins) and quantize_row_q8_0 (d = amax / 127, q = round(x / d), d kept as f16) -- restated in oracle/ggml_oracle.c
// orc_quantize_q8_K / quantize_legacy.
// Two steps so that the activation loads sit IN FRONT of the weight ring in the wave's (in-order) load queue: act_issue() puts the first
// column's values (and the norm weights) into registers with buffer loads -- thread t takes the float4 at t*4 + j*2048, i.e. wave w owns the
// 256-blocks w, w+8, ... -- then the caller fills the ring, then act_finish() normalises / quantizes while the weights are in flight.
// Rows longer than 16384 values (or 8192 with a norm) take the remaining pieces after the ring (correct, just later).
#ifndef MRS_ACT_MAXV
#define MRS_ACT_MAXV (16384 / (MRS_DEC_NT * 4))
#endif
#ifndef MRS_ACT_MAXW
#define MRS_ACT_MAXW (8192 / (MRS_DEC_NT * 4))
#endif
constexpr int ACT_MAXV = MRS_ACT_MAXV, ACT_MAXW = MRS_ACT_MAXW;  // register-resident pieces: rows of <= 16384 values (8192 with a norm)
// NV / NWV register-resident float4 pieces of the activation row / of the norm weights per thread.  The loads are unconditional (exact vmcnt
// bookkeeping), and a load past the row still costs its ~16 cycles in the texture addresser: with 8 + 4 pieces per wave the out-of-range ones of a
// 4096-wide row cost ~0.5 us per launch, so rows of <= 2 pieces (4096 values at 512 threads) run a <2, 2> instantiation (round 3)
template <int NV_, int NWV_> struct ActPreT { static constexpr int NV = NV_, NWV = NWV_; v4u xv[NV_]; v4u wv[NWV_]; };
using ActPre = ActPreT<ACT_MAXV, ACT_MAXW>;
using ActPreSmall = ActPreT<2, 2>;
__device__ __forceinline__ float4 as_f4(v4u v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }

// SC1: loads at agent scope (bypass the CU's L1) -- the persistent step kernel reads vectors that other CUs wrote a moment ago
template <bool SC1> __device__ __forceinline__ v4u ld_act(__amdgpu_buffer_rsrc_t r, unsigned off) {
  if constexpr (SC1) return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
  else return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
template <bool SC1, class AP = ActPre> __device__ __forceinline__ AP act_issue(const float *x, const float *nw, int K) {
  AP p;
  const unsigned off = (unsigned)tid_opaque() * 16u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? nw : x), (short)0, nw ? K * 4 : 0, 0x00020000);
#pragma unroll
  for (int j = 0; j < AP::NV; ++j) p.xv[j] = ld_act<SC1>(rx, off + (unsigned)j * (unsigned)(ACT_STRIDE * 4));  // beyond K: out of range, zeros, no traffic
#pragma unroll
  for (int j = 0; j < AP::NWV; ++j) p.wv[j] = ld_act<false>(rw, off + (unsigned)j * (unsigned)(ACT_STRIDE * 4));
  return p;
}

__device__ __forceinline__ float wave_max_all(float v) {
  v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v)); v = fmaxf(v, dppf<0x140>(v));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ int wave_min_all(int v) {
  v = min(v, dppi<0xB1>(v)); v = min(v, dppi<0x4E>(v)); v = min(v, dppi<0x141>(v)); v = min(v, dppi<0x140>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// sum over the wave, every lane gets it (DPP inside rows of 16, readlane across)
__device__ __forceinline__ float wave_sum_all(float v) {
  v += dppf<0xB1>(v);
  v += dppf<0x4E>(v);
  v += dppf<0x141>(v);
  v += dppf<0x140>(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}


// round half away from zero (C roundf / Rust f32::round): trunc(x + copysign(0.49999997, x)) -- the largest float below one half -- equals roundf(x)
// for EVERY float with |x| <= 129 (checked exhaustively, 2.2e9 values: tests/test_oracle.py::test_round_trick_exhaustive); the quantizers only round
// products inside [-128, 128].  Three instructions instead of eight.
__device__ __forceinline__ float round_away(float x) { return truncf(x + copysignf(0.49999997f, x)); }

// x / m for the RmsNorm (candle: x / sqrt(mean + eps) * w): y = 1 / m correctly rounded (computed once per column), then Markstein's final step
// q0 = x y, r = x - m q0 (exact in the fma), q = q0 + r y -- the correctly rounded quotient (checked against `/`: tests/test_dec_engine.py)
__device__ __forceinline__ float div_by(float x, float m, float y) {
  const float q0 = x * y;
  const float r = fmaf(-m, q0, x);
  return fmaf(r, y, q0);
}

// quantize the 4 values a lane holds at element e (all lanes of the wave together: one 256-block in Q8_K mode, 8 blocks of 32 in Q8_0 mode).
// qoff = byte offset of the lane's 4 quants inside the column's (swizzled) int8 image.
__device__ __forceinline__ void quantize4(float4 v, int e, int qoff, bool in, int mode, char *qc, float *dc, int *bsc) {
  const int lane = lane_opaque();
  if (mode == ACT_Q8K) {
    // largest and smallest of the lane's four values: |.|max for the scale, and the two ballots below need one compare each
    const float hi4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), lo4 = fminf(fminf(v.x, v.y), fminf(v.z, v.w));
    const float ax = fmaxf(hi4, -lo4);
    const float amax = wave_max_all(ax);
    // candle keeps the FIRST element with the largest magnitude and its sign decides iscale.  Only when +amax and -amax both occur in the block
    // does the order matter: two ballots settle the common case, the exact first-index search runs for such ties only (wave-uniform branch).
    const unsigned long long bp = __ballot(hi4 == amax);
    const unsigned long long bn = __ballot(lo4 == -amax);
    float mx = bn == 0 ? amax : -amax;
    if (bp != 0 && bn != 0 && amax != 0.f) {
      const int cand = fabsf(v.x) == amax ? 0 : (fabsf(v.y) == amax ? 1 : (fabsf(v.z) == amax ? 2 : (fabsf(v.w) == amax ? 3 : 1 << 20)));
      const int first = wave_min_all(lane * 4 + cand);
      const int sl = (first >> 2) & 63, comp = first & 3;
      mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp == 0 ? v.x : (comp == 1 ? v.y : (comp == 2 ? v.z : v.w))), sl));
    }
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    float dd = 0.f;
    if (amax != 0.f) {  // wave-uniform
      const float iscale = -128.f / mx;
      q0 = (int)fminf(127.f, round_away(iscale * v.x)); q1 = (int)fminf(127.f, round_away(iscale * v.y));
      q2 = (int)fminf(127.f, round_away(iscale * v.z)); q3 = (int)fminf(127.f, round_away(iscale * v.w));
      dd = 1.0f / iscale;
    }
    const int packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
    int s = dot4(packed, 0x01010101, 0);  // q0 + q1 + q2 + q3 (signed bytes)
    s += dppi<0xB1>(s);
    s += dppi<0x4E>(s);  // 4 lanes = one 16-run
    if (in) {
      *(int *)(qc + qoff) = packed;
      if ((lane & 3) == 0) bsc[e >> 4] = s;
      if (lane == 0) dc[e >> 8] = dd;
    }
  } else {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, dppf<0xB1>(amax));
    amax = fmaxf(amax, dppf<0x4E>(amax));
    amax = fmaxf(amax, dppf<0x141>(amax));  // 8 lanes = one 32-block
    const float dq = amax / 127.0f, id = dq != 0.f ? 1.0f / dq : 0.0f;
    const int q0 = (int)round_away(v.x * id), q1 = (int)round_away(v.y * id), q2 = (int)round_away(v.z * id), q3 = (int)round_away(v.w * id);
    if (in) {
      *(int *)(qc + qoff) = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
      if ((e & 31) == 0) dc[e >> 5] = half_bits_to_float(float_to_half_bits(dq));
    }
  }
}

// Whole workgroup, ends with a barrier.  `red` = NCOLS * NW floats of LDS scratch.  pre = act_issue() of column 0.
// All columns' sums of squares go through ONE barrier (per column: the same per-thread / wave / workgroup summation order as a single-column call, so a
// batched step stays bit-identical to single sequences), then every column is quantized: 2 barriers per launch instead of 2 per column.
template <int NCOLS, bool SC1, class AP = ActPre>
__device__ __forceinline__ Act act_finish(char *smem, float *red, const AP &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, float eps,
                                          int K, int mode, unsigned long long *tlp = nullptr) {
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MRS_DEC_TIMELINE
#define MRS_TLP(i) do { if (tlp && tid == 0) tlp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MRS_TLP(i) do { } while (0)
#endif
  char *q = smem;
  float *d = (float *)(smem + (size_t)NCOLS * K);
  int *bs = (int *)(d + (size_t)NCOLS * (K / 32));
  const int nv = (K + ACT_STRIDE - 1) / ACT_STRIDE;  // float4 pieces per thread
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? nw : x), (short)0, nw ? K * 4 : 0, 0x00020000);
  auto wload = [&](int j) -> float4 { return as_f4(ld_act<false>(rw, (unsigned)tid * 16u + (unsigned)j * (unsigned)(ACT_STRIDE * 4))); };
  auto col_rsrc = [&](int c) { return __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)c * ldx), (short)0, K * 4, 0x00020000); };
  float nm[NCOLS], inv[NCOLS];  // m = sqrt(mean + eps) and 1 / m
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) { nm[c] = 1.0f; inv[c] = 1.0f; }
  if (nw) {  // sum of squares: per-thread partials in element order, DPP wave sums, the wave sums in wave order
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const __amdgpu_buffer_rsrc_t rx = col_rsrc(c);
      auto xload = [&](int j) -> float4 { return as_f4(ld_act<SC1>(rx, (unsigned)tid * 16u + (unsigned)j * (unsigned)(ACT_STRIDE * 4))); };
      float ss = 0.f;
      auto sq = [&](float4 v4) { ss = fmaf(v4.x, v4.x, ss); ss = fmaf(v4.y, v4.y, ss); ss = fmaf(v4.z, v4.z, ss); ss = fmaf(v4.w, v4.w, ss); };
#pragma unroll
      for (int j = 0; j < AP::NV; ++j) if (j < nv) sq(c == 0 ? as_f4(pre.xv[j]) : xload(j));
      for (int j = AP::NV; j < nv; ++j) sq(xload(j));
      ss = wave_sum_all(ss);
      if (lane == 0) red[c * NW + wave] = ss;
    }
    MRS_TLP(19);  // wave 0: activations arrived, squares summed
    __syncthreads();
    MRS_TLP(9);   // after the norm barrier
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const float *r = red + c * NW;
      float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      if constexpr (NW == 16) tot += ((r[8] + r[9]) + (r[10] + r[11])) + ((r[12] + r[13]) + (r[14] + r[15]));
      nm[c] = sqrtf(tot / (float)K + eps);
      inv[c] = 1.0f / nm[c];
    }
  }
  // byte offset of the lane's 4 quants in the swizzled image: piece = e >> 4 = tid / 4 + 128 j, superblock = piece >> 4 = wave + 8 j, so the XOR
  // mask m(superblock) depends on the wave only and the offset is a lane constant + 2048 j
  const int qoff0 = (((tid >> 2) ^ sb_mask(wave)) << 4) | ((tid & 3) << 2);
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    const __amdgpu_buffer_rsrc_t rx = col_rsrc(c);
    auto xload = [&](int j) -> float4 { return as_f4(ld_act<SC1>(rx, (unsigned)tid * 16u + (unsigned)j * (unsigned)(ACT_STRIDE * 4))); };
    char *qc = q + (size_t)c * K;
    float *dc = d + (size_t)c * (K / 32);
    int *bsc = bs + (size_t)c * (K / 16);
    const float ic = inv[c], mc = nm[c];
    auto one = [&](int j, float4 v, float4 w4) {  // uniform trip count: every lane takes part in the cross-lane steps
      const int e = tid * 4 + j * ACT_STRIDE;
      if (nw) { v.x = div_by(v.x, mc, ic) * w4.x; v.y = div_by(v.y, mc, ic) * w4.y; v.z = div_by(v.z, mc, ic) * w4.z; v.w = div_by(v.w, mc, ic) * w4.w; }
      quantize4(v, e, qoff0 + j * ACT_STRIDE, e < K, mode, qc, dc, bsc);
    };
#pragma unroll
    for (int j = 0; j < AP::NV; ++j)
      if (j < nv) one(j, c == 0 ? as_f4(pre.xv[j]) : xload(j), j < AP::NWV ? as_f4(pre.wv[j < AP::NWV ? j : 0]) : (nw ? wload(j) : make_float4(1.f, 1.f, 1.f, 1.f)));
    for (int j = AP::NV; j < nv; ++j) one(j, xload(j), nw ? wload(j) : make_float4(1.f, 1.f, 1.f, 1.f));
  }
  MRS_TLP(20);  // wave 0 quantized its share
  __syncthreads();
  return Act{q, d, bs, K};
}

// The same prologue for ONE column, cut into stages that stream() interleaves with the issue of the weight ring (round 3).  Measured with the
// timeline build (profiles/round3_decode_timeline.md): a wave spends the first ~3 us of a launch blocked on the ISSUE of its ring loads (every CU asks
// for 64-80 KiB at once and the memory system accepts requests at HBM rate), then ~2-7 us in the prologue's VALU work while nothing new is
// requested.  The activation vector arrives long before the ring is accepted (it is first in the queue), so its arithmetic can run in the issue
// slots between two ring loads: stage 1 = squares + partial sums, stage 2 = norm barrier + inverse, stages 3.. = one quantize pass each
// (without a norm the passes start at stage 1).  Same instructions on the same values in the same order per element: bit-identical to act_finish.
template <int D> struct ActStager {
  char *smem; float *red; const float *x; const float *nw; float eps; int K, mode;
  unsigned long long *tlp;  // timeline builds
  bool staged;  // wave-uniform; false: finish() runs the whole prologue (act_finish)
  float inv, nm;
  __device__ __forceinline__ int first_q() const { return nw ? 3 : 1; }
  template <class AP> __device__ __forceinline__ void quant(int j, const AP &pre) {
    const int tid = tid_opaque(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qoff0 = (((tid >> 2) ^ sb_mask(wave)) << 4) | ((tid & 3) << 2);
    char *q = smem;
    float *d = (float *)(smem + (size_t)K);
    int *bs = (int *)(d + (size_t)(K / 32));
    const int e = tid * 4 + j * ACT_STRIDE;
    float4 v = as_f4(pre.xv[j]);
    if (nw) {
      const float4 w4 = as_f4(pre.wv[j < AP::NWV ? j : 0]);
      v.x = div_by(v.x, nm, inv) * w4.x; v.y = div_by(v.y, nm, inv) * w4.y; v.z = div_by(v.z, nm, inv) * w4.z; v.w = div_by(v.w, nm, inv) * w4.w;
    }
    quantize4(v, e, qoff0 + j * ACT_STRIDE, e < K, mode, q, d, bs);
  }
  template <int I, class AP> __device__ __forceinline__ void stage(const AP &pre) {
    if (!staged) return;
    const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nv = (K + ACT_STRIDE - 1) / ACT_STRIDE;
    if (nw) {
      if constexpr (I == 1) {
        float ss = 0.f;
        auto sq = [&](float4 v4) { ss = fmaf(v4.x, v4.x, ss); ss = fmaf(v4.y, v4.y, ss); ss = fmaf(v4.z, v4.z, ss); ss = fmaf(v4.w, v4.w, ss); };
#pragma unroll
        for (int j = 0; j < AP::NV; ++j) if (j < nv) sq(as_f4(pre.xv[j]));
        ss = wave_sum_all(ss);
        if (lane == 0) red[wave] = ss;
      } else if constexpr (I == 2) {
        __syncthreads();
        const float *r = red;
        float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if constexpr (NW == 16) tot += ((r[8] + r[9]) + (r[10] + r[11])) + ((r[12] + r[13]) + (r[14] + r[15]));
        nm = sqrtf(tot / (float)K + eps);
        inv = 1.0f / nm;
      } else if constexpr (I >= 3 && I - 3 < AP::NV) {
        if (I - 3 < nv) quant(I - 3, pre);
      }
    } else {
      if constexpr (I >= 1 && I - 1 < AP::NV) {
        if (I - 1 < nv) quant(I - 1, pre);
      }
    }
  }
  // passes the stages did not reach + the closing barrier (staged), or the whole prologue
  template <class AP> __device__ __forceinline__ Act finish(const AP &pre) {
    const int nv = (K + ACT_STRIDE - 1) / ACT_STRIDE;
    const int done = D - first_q();
#pragma unroll
    for (int j = 0; j < AP::NV; ++j) if (j >= done && j < nv) quant(j, pre);
    __syncthreads();
    return Act{smem, (const float *)(smem + (size_t)K), (const int *)(smem + (size_t)K + (size_t)(K / 32) * 4), K};
  }
  // rows the register-resident pieces cover, a norm whose weights fit the registers
  static __device__ __forceinline__ bool fits(int K, bool norm) { return K <= ACT_MAXV * ACT_STRIDE && (!norm || K <= ACT_MAXW * ACT_STRIDE); }
};

// ------------------------------------------------------------------------------------------------ per-format tiles
// A tile = what the 64 lanes of a wave take from one row in one step.  Raw = the registers a lane holds for it (filled by buffer loads);
// LaneC = lane constants (LDS offsets for tile 0 of a row); accumulate() adds the lane's share of <row, activation column> to acc[].
__device__ __forceinline__ v4u ldb128(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2); }  // aux 2 = nt: weights are read once per token
__device__ __forceinline__ unsigned ldb32(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ unsigned ldb16(__amdgpu_buffer_rsrc_t r, unsigned off) { return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0); }
__device__ __forceinline__ int dot16u(v4u q, int4 u) { return dot4((int)q.w, u.w, dot4((int)q.z, u.z, dot4((int)q.y, u.y, dot4((int)q.x, u.x, 0)))); }
__device__ __forceinline__ int mul24i(int a, int b) { return __mul24(a, b); }

template <int TYPE> struct Tile;

#ifndef MRS_DEC_DEPTH_Q4K
#define MRS_DEC_DEPTH_Q4K 8
#endif
#ifndef MRS_DEC_DEPTH_Q6K
#define MRS_DEC_DEPTH_Q6K 4
#endif
template <> struct Tile<T_Q4_K> {
  static constexpr int DEPTH = MRS_DEC_DEPTH_Q4K, UNIT = 32;
  struct Raw { v4u q; unsigned hs, hd; };
  struct LaneC { int pa, pb, ra, sbl; };
  static __device__ __forceinline__ LaneC lanec(int lane) {
    const int sbl = lane >> 3, c = (lane >> 1) & 3, hp = lane & 1, m = sb_mask(sbl), r = 4 * c + hp;
    return LaneC{(sbl * 16 + (r ^ m)) * 16, (sbl * 16 + ((r + 2) ^ m)) * 16, sbl * 16 + r, sbl};
  }
  // unit index u = t*64 + lane of row `row` (S superblocks per row); off = OOB when the unit does not exist
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, const Mat &m, unsigned row, int u, int S, bool ok) {
    const unsigned sbg = row * (unsigned)S + (unsigned)(u >> 3);
    Raw r;
    r.q = ldb128(rs, ok ? (row * (unsigned)S * 8u + (unsigned)u) * 16u : OOB);
    r.hs = ldb32(rs, ok ? m.off_hs + sbg * 16u + (unsigned)((u >> 1) & 3) * 4u : OOB);
    r.hd = ldb32(rs, ok ? m.off_hd + sbg * 4u : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void accumulate(const Raw &w, const LaneC &lc, int t, int S, const Act &act, float (&acc)[NCOLS]) {
    const v4u lo = w.q & 0x0F0F0F0Fu, hi = (w.q >> 4) & 0x0F0F0F0Fu;
    const int sca = (int)(w.hs & 0xff), scb = (int)((w.hs >> 8) & 0xff), ma = (int)((w.hs >> 16) & 0xff), mb = (int)(w.hs >> 24);
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    const int K = act.K;
    const int sb = min(t * 8 + lc.sbl, S - 1);  // lanes without a unit hold zeros; keep their scale read inside the row
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const char *qc = act.q + (size_t)c * K + t * 2048;
      const int4 ua = *(const int4 *)(qc + lc.pa), ub = *(const int4 *)(qc + lc.pb);
      const int *bsc = act.bs + (size_t)c * (K / 16) + t * 128 + lc.ra;
      const int isum = mul24i(scb, dot16u(hi, ub)) + mul24i(sca, dot16u(lo, ua));
      const int msum = mul24i(mb, bsc[2]) + mul24i(ma, bsc[0]);
      const float yd = act.d[(size_t)c * (K / 32) + sb];
      acc[c] = fmaf(d * yd, (float)isum, acc[c]);
      acc[c] = fmaf(-(dmin * yd), (float)msum, acc[c]);
    }
  }
};

template <> struct Tile<T_Q5_K> {
  static constexpr int DEPTH = 8, UNIT = 32;
  struct Raw { v4u q; unsigned xh, hs, hd; };
  using LaneC = Tile<T_Q4_K>::LaneC;
  static __device__ __forceinline__ LaneC lanec(int lane) { return Tile<T_Q4_K>::lanec(lane); }
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, const Mat &m, unsigned row, int u, int S, bool ok) {
    const unsigned sbg = row * (unsigned)S + (unsigned)(u >> 3), ug = row * (unsigned)S * 8u + (unsigned)u;
    Raw r;
    r.q = ldb128(rs, ok ? ug * 16u : OOB);
    r.xh = ldb32(rs, ok ? m.off_x + ug * 4u : OOB);
    r.hs = ldb32(rs, ok ? m.off_hs + sbg * 16u + (unsigned)((u >> 1) & 3) * 4u : OOB);
    r.hd = ldb32(rs, ok ? m.off_hd + sbg * 4u : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void accumulate(const Raw &w, const LaneC &lc, int t, int S, const Act &act, float (&acc)[NCOLS]) {
    // xh bit 8j+k = fifth bit of low-nibble weight 4k+j, bit 8j+4+k = of high-nibble weight 4k+j (k = dword, j = byte)
    v4u lo = w.q & 0x0F0F0F0Fu, hi = (w.q >> 4) & 0x0F0F0F0Fu;
    lo.x |= (w.xh & 0x01010101u) << 4; lo.y |= ((w.xh >> 1) & 0x01010101u) << 4; lo.z |= ((w.xh >> 2) & 0x01010101u) << 4; lo.w |= ((w.xh >> 3) & 0x01010101u) << 4;
    hi.x |= ((w.xh >> 4) & 0x01010101u) << 4; hi.y |= ((w.xh >> 5) & 0x01010101u) << 4; hi.z |= ((w.xh >> 6) & 0x01010101u) << 4; hi.w |= ((w.xh >> 7) & 0x01010101u) << 4;
    const int sca = (int)(w.hs & 0xff), scb = (int)((w.hs >> 8) & 0xff), ma = (int)((w.hs >> 16) & 0xff), mb = (int)(w.hs >> 24);
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    const int K = act.K;
    const int sb = min(t * 8 + lc.sbl, S - 1);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const char *qc = act.q + (size_t)c * K + t * 2048;
      const int4 ua = *(const int4 *)(qc + lc.pa), ub = *(const int4 *)(qc + lc.pb);
      const int *bsc = act.bs + (size_t)c * (K / 16) + t * 128 + lc.ra;
      const int isum = mul24i(scb, dot16u(hi, ub)) + mul24i(sca, dot16u(lo, ua));
      const int msum = mul24i(mb, bsc[2]) + mul24i(ma, bsc[0]);
      const float yd = act.d[(size_t)c * (K / 32) + sb];
      acc[c] = fmaf(d * yd, (float)isum, acc[c]);
      acc[c] = fmaf(-(dmin * yd), (float)msum, acc[c]);
    }
  }
};

template <> struct Tile<T_Q6_K> {
  static constexpr int DEPTH = MRS_DEC_DEPTH_Q6K, UNIT = 64;
  struct Raw { v4u l0, l1, x; unsigned hs, hd; };
  struct LaneC { int pa0, pa1, pb0, pb1, ra, sbl; };
  static __device__ __forceinline__ LaneC lanec(int lane) {
    const int sbl = lane >> 2, u = lane & 3, h = u >> 1, j = u & 1, m = sb_mask(sbl), r = 8 * h + 2 * j;
    return LaneC{(sbl * 16 + (r ^ m)) * 16, (sbl * 16 + ((r + 1) ^ m)) * 16, (sbl * 16 + ((r + 4) ^ m)) * 16, (sbl * 16 + ((r + 5) ^ m)) * 16, sbl * 16 + r, sbl};
  }
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, const Mat &m, unsigned row, int u, int S, bool ok) {
    const unsigned sbg = row * (unsigned)S + (unsigned)(u >> 2), ug = row * (unsigned)S * 4u + (unsigned)u;
    Raw r;
    r.l0 = ldb128(rs, ok ? ug * 32u : OOB);
    r.l1 = ldb128(rs, ok ? ug * 32u + 16u : OOB);
    r.x = ldb128(rs, ok ? m.off_x + ug * 16u : OOB);
    r.hs = ldb32(rs, ok ? m.off_hs + sbg * 16u + (unsigned)(u & 3) * 4u : OOB);
    r.hd = ldb16(rs, ok ? m.off_hd + sbg * 2u : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void accumulate(const Raw &w, const LaneC &lc, int t, int S, const Act &act, float (&acc)[NCOLS]) {
    // x byte b: bits 1:0 -> weight b (A0), 3:2 -> 16+b (A1), 5:4 -> 32+b (B0), 7:6 -> 48+b (B1) of the unit; low nibbles of l0/l1 = A, high = B
    const v4u a0 = (w.l0 & 0x0F0F0F0Fu) | ((w.x << 4) & 0x30303030u);
    const v4u a1 = (w.l1 & 0x0F0F0F0Fu) | ((w.x << 2) & 0x30303030u);
    const v4u b0 = ((w.l0 >> 4) & 0x0F0F0F0Fu) | (w.x & 0x30303030u);
    const v4u b1 = ((w.l1 >> 4) & 0x0F0F0F0Fu) | ((w.x >> 2) & 0x30303030u);
    const int s0 = (int)(int8_t)(w.hs & 0xff), s1 = (int)(int8_t)((w.hs >> 8) & 0xff), s2 = (int)(int8_t)((w.hs >> 16) & 0xff), s3 = (int)(int8_t)(w.hs >> 24);
    const float d = half_bits_to_float((uint16_t)w.hd);
    const int K = act.K;
    const int sb = min(t * 16 + lc.sbl, S - 1);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const char *qc = act.q + (size_t)c * K + t * 4096;
      const int4 u0 = *(const int4 *)(qc + lc.pa0), u1 = *(const int4 *)(qc + lc.pa1), u2 = *(const int4 *)(qc + lc.pb0), u3 = *(const int4 *)(qc + lc.pb1);
      const int *bsc = act.bs + (size_t)c * (K / 16) + t * 256 + lc.ra;
      // sum sc * <q - 32, u> = sum sc * (<q, u> - 32 * sum u), all integer
      int isum = mul24i(s0, dot16u(a0, u0)) + mul24i(s1, dot16u(a1, u1));
      isum += mul24i(s2, dot16u(b0, u2)) + mul24i(s3, dot16u(b1, u3));
      const int osum = (mul24i(s0, bsc[0]) + mul24i(s1, bsc[1])) + (mul24i(s2, bsc[4]) + mul24i(s3, bsc[5]));
      const float yd = act.d[(size_t)c * (K / 32) + sb];
      acc[c] = fmaf(d * yd, (float)(isum - 32 * osum), acc[c]);
    }
  }
};

template <> struct Tile<T_Q8_0> {
  static constexpr int DEPTH = 8, UNIT = 16;
  struct Raw { v4u q; unsigned hd; };
  struct LaneC { int pa, odd, blk; };
  static __device__ __forceinline__ LaneC lanec(int lane) { return LaneC{(lane ^ sb_mask(lane >> 4)) * 16, lane & 1, lane >> 1}; }
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, const Mat &m, unsigned row, int u, int S /* = K/16 units per row */, bool ok) {
    Raw r;
    r.q = ldb128(rs, ok ? (row * (unsigned)S + (unsigned)u) * 16u : OOB);
    r.hd = ldb16(rs, ok ? m.off_hd + (row * (unsigned)(S >> 1) + (unsigned)(u >> 1)) * 2u : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void accumulate(const Raw &w, const LaneC &lc, int t, int S, const Act &act, float (&acc)[NCOLS]) {
    const float dw = half_bits_to_float((uint16_t)w.hd);
    const int K = act.K;
    const int blk = min(t * 32 + lc.blk, (S >> 1) - 1);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const int4 u = *(const int4 *)(act.q + (size_t)c * K + t * 1024 + lc.pa);
      int isum = dot16u(w.q, u);
      isum += dppi<0xB1>(isum);  // the two halves of the 32-block sit on lanes 2b, 2b+1: block dot in integers, as vec_dot_q8_0_q8_0
      const float dx = act.d[(size_t)c * (K / 32) + blk];
      const float p = (float)isum * dw * dx;
      acc[c] += lc.odd ? 0.0f : p;
    }
  }
};

template <int TYPE> __host__ __device__ constexpr int unit_weights() { return TYPE == T_Q6_K ? 64 : TYPE == T_Q8_0 ? 16 : 32; }
// "S" argument of Tile::load / accumulate: superblocks per row for the K-quants, 16-weight units per row for Q8_0
template <int TYPE> __host__ __device__ inline int row_param(int K) { return TYPE == T_Q8_0 ? K / 16 : K / 256; }

// ------------------------------------------------------------------------------------------------ the streaming core
// One wave streams up to two row segments of the SAME format: segment i = rows [row0[i], row0[i] + nrows[i]) of tensor mat[i].
// The ring is filled before `pro()` (the activation prologue, which contains the workgroup barriers and, in the persistent step kernel, the
// wait for the producer phase) and never drains across rows or segments.  epi(seg, row, acc) is called once per finished row with
// wave-uniform sums.
struct Segs {
  Mat mat[2];
  int row0[2], nrows[2];
  int nseg;
};

// SEGCOL (NCOLS must be 1): segment s multiplies by activation column s of a 2-column image (MoE down: two experts' rows against their own activations)
struct NoStager {};
template <int TYPE, int NCOLS, bool SEGCOL = false, class Pre, class Pro, class Epi, class Stg = NoStager>
__device__ __forceinline__ void stream(const Segs &sg, int K, Pre pre, Pro pro, Epi epi, bool skip_acc = false, Stg *stg = nullptr) {
  using TL = Tile<TYPE>;
  constexpr int D = TL::DEPTH;
  const int lane = lane_opaque();
  const int upr = K / unit_weights<TYPE>();  // units per row
  const int tpr = (upr + 63) >> 6;            // tiles per row
  const int S = row_param<TYPE>(K);
  const int rows0 = sg.nrows[0], rows1 = sg.nseg > 1 ? sg.nrows[1] : 0;
  const int total = (rows0 + rows1) * tpr;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)sg.mat[0].base, (short)0, (int)sg.mat[0].bytes, 0x00020000);
  const bool two = sg.nseg > 1;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(two ? sg.mat[1].base : sg.mat[0].base), (short)0, (int)(two ? sg.mat[1].bytes : sg.mat[0].bytes), 0x00020000);
  typename TL::Raw ring[D];
  int lr = 0, lt = 0, lseg = 0;  // loader cursor: row inside the segment, tile inside the row, segment
  auto issue = [&](typename TL::Raw &slot) {
    const bool live = lseg == 0 ? lr < rows0 : (lseg == 1 && lr < rows1);
    const int u = lt * 64 + lane;
    const bool ok = live && u < upr;
    const unsigned row = (unsigned)((lseg == 0 ? sg.row0[0] : sg.row0[1]) + lr);
    slot = lseg == 0 ? TL::load(rs0, sg.mat[0], row, u, S, ok) : TL::load(rs1, sg.mat[1], row, u, S, ok);
    if (++lt == tpr) { lt = 0; if (++lr == (lseg == 0 ? rows0 : rows1) && lseg == 0 && rows1 > 0) { lr = 0; lseg = 1; } }
  };
  const auto pr = pre();  // activation loads first: they are small and must not wait behind the ring in the in-order return queue
  if constexpr (std::is_same<Stg, NoStager>::value) {
#pragma unroll
    for (int i = 0; i < D; ++i) issue(ring[i]);
  } else {  // prologue stages in the issue slots between the ring loads (ActStager)
    static_assert(D == 4 || D == 8 || D == 12 || D == 16, "ring depth");
#ifdef MRS_DEC_TIMELINE
#define MRS_TLS(i) do { if (stg->tlp && tid_opaque() == 0) stg->tlp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MRS_TLS(i) do { } while (0)
#endif
#define MRS_ISSUE_STAGE(i) __builtin_amdgcn_sched_barrier(0); issue(ring[i]); MRS_TLS(21 + i); __builtin_amdgcn_sched_barrier(0); stg->template stage<i>(pr)
    issue(ring[0]); MRS_TLS(21);
    MRS_ISSUE_STAGE(1); MRS_ISSUE_STAGE(2); MRS_ISSUE_STAGE(3);
    if constexpr (D >= 8) { MRS_ISSUE_STAGE(4); MRS_ISSUE_STAGE(5); MRS_ISSUE_STAGE(6); MRS_ISSUE_STAGE(7); }
    if constexpr (D >= 12) { MRS_ISSUE_STAGE(8); MRS_ISSUE_STAGE(9); MRS_ISSUE_STAGE(10); MRS_ISSUE_STAGE(11); }
    if constexpr (D >= 16) { MRS_ISSUE_STAGE(12); MRS_ISSUE_STAGE(13); MRS_ISSUE_STAGE(14); MRS_ISSUE_STAGE(15); }
    __builtin_amdgcn_sched_barrier(0);
  }
  const Act act = pro(pr);
  const Act act1 = Act{act.q + K, act.d + K / 32, act.bs + K / 16, K};  // column 1 (SEGCOL)
  const typename TL::LaneC lc = TL::lanec(lane);
  float acc[NCOLS];
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) acc[c] = 0.0f;
  int cr = 0, ct = 0, cseg = 0;
  for (int g = 0; g < total; g += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (g + i < total) {  // wave-uniform
        if constexpr (SEGCOL) { if (cseg == 0) TL::template accumulate<NCOLS>(ring[i], lc, ct, S, act, acc); else TL::template accumulate<NCOLS>(ring[i], lc, ct, S, act1, acc); }
        else if (!skip_acc) TL::template accumulate<NCOLS>(ring[i], lc, ct, S, act, acc);
        else acc[0] += __uint_as_float(ring[i].hd & 1u);  // experiment: keep the loads alive (in-order return: the last load of the slot), drop the arithmetic
        if (++ct == tpr) {
          float sum[NCOLS];
#pragma unroll
          for (int c = 0; c < NCOLS; ++c) { sum[c] = wave_sum_all(acc[c]); acc[c] = 0.0f; }
          epi(cseg, (cseg == 0 ? sg.row0[0] : sg.row0[1]) + cr, sum);
          ct = 0;
          if (++cr == (cseg == 0 ? rows0 : rows1) && cseg == 0) { cr = 0; cseg = 1; }
        }
      }
      issue(ring[i]);
    }
  }
}

}  // namespace dec
}  // namespace mrs
