// dec_core2.cuh -- MI355X decode engine, GEMV core (round 5, "v3"): four lanes = one superblock, tiles of 16 superblocks, deep register ring.
//
// What the reference does on this path: GgufMatMul::forward_raw -> candle QMatMul::forward with f32 activations (mistralrs-quant/src/gguf/mod.rs:465-478):
// every activation row is quantized to the vec_dot partner of the weight format -- Q8_K (one f32 scale per 256 values, int8 quants, per-16 sums) for the
// K-quants, Q8_0 (f16 scale per 32) for Q8_0 -- and every output is a sum over superblocks of (d_w d_x) <integer dot> (- (dmin d_x) <integer min term>).
// The engine computes exactly these integers; the f32 combination follows ONE order that the batch-1 GEMV (this file), the batched GEMV and the prompt GEMM
// on the matrix cores (ext_gemm_qi.hip) all share, so a token's result does not depend on which kernel produced it ("ORD-U", restated in plain C in
// oracle/cpu_path_oracle.c orc_gemv_engine, which the kernels equal bit for bit):
//     T_sb   = one f32 term per 256-value superblock from the superblock's EXACT integer sums
//                Q4_K / Q5_K: fma(d yd, (float)isum, -((dmin yd) (float)msum));  Q6_K: (d yd) (float)isum;  Q8_0: ((isum_b dw_b) dx_b) summed over its 8 blocks in order
//     chunks = the row's S superblocks cut into 4 runs of Cs = ceil(S / 4); c_p = T summed left to right inside run p
//     row    = ((c_0 + c_1) + c_2) + c_3
//
// MI355X design, round 5.  Round 4 gave a lane a WHOLE superblock (37-68 VGPRs per record in flight): the register ring was 1-3 records deep, the kernels ran two
// waves per SIMD with ~200 VGPRs, a launch's memory pipe went idle for the 3-4 us of the activation prologue, and the batched instantiations spilled
// (VERDICT round 4, items 1-3, 5, 7).  Here:
//   * a TILE is 16 superblocks = 4 consecutive rows x the 4 ORD-U chunks, FOUR lanes per superblock: lane (r, p, c) = row r of the group, chunk p, quarter c of
//     the superblock (64 weights = one sub-block pair of the K-quants, two blocks of Q8_0).  A row group is Cs consecutive tiles (tile t = superblock t of every chunk),
//     so chunk p's left-to-right f32 sum is an in-lane accumulation over the group's tiles and the row sum is three DPP steps at the last tile: no LDS / readlane
//     traffic per tile, and the same lane <-> row mapping for every K;
//   * inside a tile every plane is lane-major (piece i of all 64 lanes contiguous): a `buffer_load_dwordx4 ... nt` of the wave reads 1 KiB of consecutive bytes;
//     a lane holds 10 (Q4_K) .. 17 (Q8_0) registers per tile in flight, so the ring is 4-8 tiles deep (up to 150 KiB in flight per CU with 8 waves): everything the
//     prologue window needs is already requested when the prologue starts, and the memory pipe stays busy through it;
//   * integer sub-block dots per quarter (v_dot4_i32_i8), 6/8-bit scale products, then TWO quad_perm DPP adds per integer give every lane of the quad the
//     superblock's exact isum / msum; T is computed redundantly on the four lanes;
//   * per activation column the state is one f32 accumulator: the batched (2..8 columns) kernels keep everything in registers;
//   * activations in LDS: int8 per superblock, chunk-major with a 16-byte skew per chunk (the 16 (chunk, quarter) pairs of a ds_read_b128 lane group hit 16
//     different bank groups), f32 block scales, int32 per-16 sums.
#pragma once
#include "gguf_blocks.cuh"
#include <type_traits>
#include <utility>

#ifndef MRS_WAVE_SYNC
#define MRS_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

namespace mrs {
namespace dec2 {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#ifndef MRS_DEC2_NT
#define MRS_DEC2_NT 512
#endif
constexpr int NT = MRS_DEC2_NT, NW = NT / 64;  // threads / waves per workgroup (one workgroup per CU)

__host__ __device__ inline bool dec_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }
enum : int { ACT_Q8K = 0, ACT_Q80 = 1 };
__host__ __device__ inline int act_mode_for(int type) { return type == T_Q8_0 ? ACT_Q80 : ACT_Q8K; }

// ------------------------------------------------------------------------------------------------ geometry
// S superblocks per row, Cs per chunk.  A record group = R = 4 consecutive rows = TPC = Cs tiles; inside a tile: lane = (r * 4 + p) * 4 + c.
// (LPC / W / A keep the names the launchers and epilogues have used since round 4: lanes per chunk of a row, superblocks per chunk and tile, slots per tile.)
struct Geo { int S, Cs, LPC, TPC, W, R, A; };
__host__ __device__ inline Geo geo_for(int K) {
  Geo g;
  g.S = K / 256;
  g.Cs = (g.S + 3) / 4;
  g.LPC = 4;
  g.TPC = g.Cs;
  g.W = 1;
  g.R = 4;
  g.A = 16;
  return g;
}
// bytes of one superblock in the decode layout
__host__ __device__ constexpr int slot_bytes(int type) { return type == T_Q4_K ? 148 : type == T_Q5_K ? 180 : type == T_Q6_K ? 210 : 272; }
__host__ __device__ constexpr unsigned tile_bytes(int type) { return 16u * (unsigned)slot_bytes(type); }
__host__ __device__ inline size_t rec_bytes(int type, const Geo &) { return tile_bytes(type); }
__host__ __device__ inline size_t tensor_bytes(int type, long long n, long long k) {
  const Geo g = geo_for((int)k);
  const size_t rgs = (size_t)((n + g.R - 1) / g.R);
  return rgs * g.TPC * (size_t)tile_bytes(type);
}

// one tensor in decode layout, as the kernels see it
struct Mat {
  const uint8_t *base;
  unsigned bytes;
  int type, n, k;
};

// ------------------------------------------------------------------------------------------------ small wave helpers
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float rlf(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ float wave_max_all(float v) {
  v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v)); v = fmaxf(v, dppf<0x140>(v));
  return fmaxf(fmaxf(rlf(v, 0), rlf(v, 16)), fmaxf(rlf(v, 32), rlf(v, 48)));
}
__device__ __forceinline__ int wave_min_all(int v) {
  v = min(v, dppi<0xB1>(v)); v = min(v, dppi<0x4E>(v)); v = min(v, dppi<0x141>(v)); v = min(v, dppi<0x140>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// sum over the wave, every lane gets it: DPP inside rows of 16 (xor 1, xor 2, mirror 8, mirror 16), then (r0 + r16) + (r32 + r48)
__device__ __forceinline__ float wave_sum_all(float v) {
  v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); v += dppf<0x140>(v);
  return (rlf(v, 0) + rlf(v, 16)) + (rlf(v, 32) + rlf(v, 48));
}
// round half away from zero (Rust f32::round): trunc(x + copysign(0.49999997, x)) == roundf(x) for every |x| <= 129 (tests/test_oracle.py, exhaustive)
__device__ __forceinline__ float round_away(float x) { return truncf(x + copysignf(0.49999997f, x)); }
// x / m, correctly rounded, from y = 1 / m: q0 = x y, r = x - m q0 (exact in the fma), q = q0 + r y  (checked against `/`: tests/test_dec_engine.py)
__device__ __forceinline__ float div_by(float x, float m, float y) {
  const float q0 = x * y;
  const float r = fmaf(-m, q0, x);
  return fmaf(r, y, q0);
}
__device__ __forceinline__ float4 as_f4(v4u v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
__device__ __forceinline__ int dot16(v4u q, int4 u, int acc) { return dot4((int)q.w, u.w, dot4((int)q.z, u.z, dot4((int)q.y, u.y, dot4((int)q.x, u.x, acc)))); }
// sum over the four lanes of a quad, every lane gets it (integers: exact in any order)
__device__ __forceinline__ int quad_sum(int v) { v += dppi<0xB1>(v); v += dppi<0x4E>(v); return v; }

// ------------------------------------------------------------------------------------------------ activations in LDS (and, byte for byte, the pre-quantized image
// a producer kernel leaves in global memory).  ncols columns of K values:
//   q  [ncols][4][CQ]   int8: chunk p of a column at p * CQ, CQ = Cs * 256 + 16; superblock p * Cs + t at t * 256 inside its chunk, element order
//                       (a tile's lanes read (chunk p, quarter c) at p * CQ + c * 64 + const: 16 bank groups for the 16 pairs)
//   d  [ncols][Sp][DW]  f32: Q8_K mode DW = 1: d of the superblock;  Q8_0 mode DW = 12: f32(f16(d)) of the 8 blocks (4 pad floats);  Sp = 4 Cs >= S slots (a tile's
//                       dead slots -- superblock index >= S -- read inside the image and are discarded)
//   bs [ncols][Sp][20]  int32 sums of the 16 runs of 16 (Q8_K mode; 4 pad ints)
constexpr int ACT_BS = 20;
__host__ __device__ inline int act_dw(int mode) { return mode == ACT_Q80 ? 12 : 1; }
__host__ __device__ inline int act_cq(int K) { return ((K / 256 + 3) / 4) * 256 + 16; }                                                   // bytes of one chunk of one column
__host__ __device__ inline int act_sp(int K) { return 4 * ((K / 256 + 3) / 4); }  // superblock slots of a row in the d / bs regions: the 4 x Cs slots of a row group's tiles (>= S)
__host__ __device__ inline size_t act_bytes(int K, int ncols) { return (size_t)ncols * ((size_t)4 * act_cq(K) + (size_t)act_sp(K) * (48 + ACT_BS * 4)); }  // both modes fit
// chunk p = sb / Cs of superblock sb < 4 Cs, without the integer division (a run-time divisor: ~40 instructions and a reciprocal's latency per superblock quantized)
__host__ __device__ inline int chunk_of(int sb, int Cs) { return (sb >= Cs ? 1 : 0) + (sb >= 2 * Cs ? 1 : 0) + (sb >= 3 * Cs ? 1 : 0); }
struct Act {
  const char *q;
  const float *d;
  const int *bs;
  int K, S, dw, Cs, CQ, Sp;
  // byte offset of superblock sb's quants inside its column
  __device__ __forceinline__ int qoff(int sb) const { const int p = chunk_of(sb, Cs); return p * CQ + (sb - p * Cs) * 256; }
  __device__ __forceinline__ size_t qcol(int c) const { return (size_t)c * 4 * CQ; }
};
__device__ __forceinline__ Act act_view(char *smem, int K, int ncols, int mode) {
  const int S = K / 256, dw = act_dw(mode), CQ = act_cq(K), Sp = act_sp(K);
  char *d = smem + (size_t)ncols * 4 * CQ;
  return Act{smem, (const float *)d, (const int *)(d + (size_t)ncols * Sp * dw * 4), K, S, dw, (S + 3) / 4, CQ, Sp};
}

// quantize N superblocks at once: lane l holds elements 4 l .. 4 l + 3 of each (v[n], superblock sb[n], live[n] wave-uniform) -> column c of the image.
// Every step is a loop over n so that the N reduction chains interleave (one chain alone is ~100 dependent instructions).
// Quantizers: candle BlockQ8K::from_float (amax with its sign, iscale = -128 / max, q = min(127, round(iscale x)), d = 1 / iscale, the FIRST
// element of largest magnitude wins) and quantize_row_q8_0 (d = amax / 127, q = round(x / d), d kept as f16) -- oracle/ggml_oracle.c.
template <int N>
__device__ __forceinline__ void quantize_multi(const float4 (&v)[N], const int (&sb)[N], const bool (&live)[N], int c, int mode, char *img, int K, int ncols) {
  const int lane = lane_opaque(), S = K / 256, dw = act_dw(mode), Cs = (S + 3) / 4, CQ = act_cq(K), Sp = act_sp(K);
  char *q0p = img + (size_t)c * 4 * CQ;
  float *d0p = (float *)(img + (size_t)ncols * 4 * CQ) + (size_t)c * Sp * dw;
  int *b0p = (int *)(img + (size_t)ncols * 4 * CQ + (size_t)ncols * Sp * dw * 4) + (size_t)c * Sp * ACT_BS;
  int qo[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { const int sbn = live[n] ? sb[n] : 0, p = chunk_of(sbn, Cs); qo[n] = p * CQ + (sbn - p * Cs) * 256; }
  if (mode == ACT_Q8K) {
    float amax[N], mx[N];
    bool tie = false;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float hi4 = fmaxf(fmaxf(v[n].x, v[n].y), fmaxf(v[n].z, v[n].w)), lo4 = fminf(fminf(v[n].x, v[n].y), fminf(v[n].z, v[n].w));
      amax[n] = wave_max_all(fmaxf(hi4, -lo4));
      // the sign of the first element with the largest magnitude decides iscale; only when +amax and -amax both occur does the order matter
      const unsigned long long bp = __ballot(hi4 == amax[n]), bn = __ballot(lo4 == -amax[n]);
      mx[n] = bn == 0 ? amax[n] : -amax[n];
      tie = tie || (bp != 0 && bn != 0 && amax[n] != 0.f);
    }
    if (tie) {  // wave-uniform, rare: the exact first-index search
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const int cand = fabsf(v[n].x) == amax[n] ? 0 : (fabsf(v[n].y) == amax[n] ? 1 : (fabsf(v[n].z) == amax[n] ? 2 : (fabsf(v[n].w) == amax[n] ? 3 : 1 << 20)));
        const int first = wave_min_all(lane * 4 + cand);
        const int sl = (first >> 2) & 63, comp = first & 3;
        const float m1 = rlf(comp == 0 ? v[n].x : (comp == 1 ? v[n].y : (comp == 2 ? v[n].z : v[n].w)), sl);
        mx[n] = amax[n] != 0.f ? m1 : mx[n];
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const bool nz = amax[n] != 0.f;
      const float iscale = nz ? -128.f / mx[n] : 0.f;  // all-zero block: every quant 0, d = 0
      const int q0 = (int)fminf(127.f, round_away(iscale * v[n].x)), q1 = (int)fminf(127.f, round_away(iscale * v[n].y));
      const int q2 = (int)fminf(127.f, round_away(iscale * v[n].z)), q3 = (int)fminf(127.f, round_away(iscale * v[n].w));
      const float dd = nz ? 1.0f / iscale : 0.f;
      const int packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
      int sum = dot4(packed, 0x01010101, 0);
      sum += dppi<0xB1>(sum);
      sum += dppi<0x4E>(sum);  // 4 lanes = one run of 16
      if (live[n]) {
        *(int *)(q0p + qo[n] + lane * 4) = packed;
        if ((lane & 3) == 0) b0p[(size_t)sb[n] * ACT_BS + (lane >> 2)] = sum;
        if (lane == 0) d0p[(size_t)sb[n] * dw] = dd;
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float amax = fmaxf(fmaxf(fabsf(v[n].x), fabsf(v[n].y)), fmaxf(fabsf(v[n].z), fabsf(v[n].w)));
      amax = fmaxf(amax, dppf<0xB1>(amax));
      amax = fmaxf(amax, dppf<0x4E>(amax));
      amax = fmaxf(amax, dppf<0x141>(amax));  // 8 lanes = one block of 32
      const float dq = amax / 127.0f, id = dq != 0.f ? 1.0f / dq : 0.0f;
      const int q0 = (int)round_away(v[n].x * id), q1 = (int)round_away(v[n].y * id), q2 = (int)round_away(v[n].z * id), q3 = (int)round_away(v[n].w * id);
      if (live[n]) {
        *(int *)(q0p + qo[n] + lane * 4) = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
        if ((lane & 7) == 0) d0p[(size_t)sb[n] * dw + (lane >> 3)] = half_bits_to_float(float_to_half_bits(dq));
      }
    }
  }
}
__device__ __forceinline__ void quantize_sb(float4 v, int sb, int c, int mode, char *img, int K, int ncols) {
  const float4 va[1] = {v}; const int sa[1] = {sb}; const bool la[1] = {true};
  quantize_multi<1>(va, sa, la, c, mode, img, K, ncols);
}
__device__ __forceinline__ float4 norm4(float4 v, float4 w4, float nm, float inv) {
  v.x = div_by(v.x, nm, inv) * w4.x; v.y = div_by(v.y, nm, inv) * w4.y; v.z = div_by(v.z, nm, inv) * w4.z; v.w = div_by(v.w, nm, inv) * w4.w;
  return v;
}

// ---- the activation prologue: f32 activations x [NCOLS][ldx] (optionally RmsNorm(x) w first: RmsNorm::forward, mistralrs-core/src/layers.rs:403-414) -> the image.
// Sum of squares: 512 "virtual threads" = 8 virtual waves, virtual thread t sums the squares of its float4 pieces 4 t + 2048 j (j ascending, x y z w, fma);
// wave_sum_all per virtual wave; the 8 wave sums as ((0+1)+(2+3))+((4+5)+(6+7)) -- the order the engine has had since round 2 (oracle: orc_rms_norm_engine).
// Virtual wave v quantizes superblocks v, v + 8, ... (its own pieces: piece j of virtual wave v IS superblock v + 8 j), two at a time.
// The work is done by P = 8 / VW PARTICIPANT waves, participant q playing the virtual waves q, q + P, ...: VW = 1: all 8 waves of a workgroup (the prompt path's
// quantizer, ext_gemm_qi.hip); VW = 2: the 4 prologue waves of the decode GEMV (stream()), while the other 4 waves request weights.  Every value is loaded ONCE per
// workgroup (256 workgroups reading the same 16 KB row are a hot spot in the L2 channels: a participant that re-read the whole row -- round 4's SPEC schedule --
// needed 3 us for its requests alone).  Register-resident per virtual thread: NPX pieces of x (rows of <= 2048 NPX values), NPW of the norm weight; longer rows
// and the other columns of a batch take plain loads.
constexpr int NPX = 7, NPW = 4;
template <int VW> struct ActRegs { v4u xv[VW][NPX]; v4u wv[VW][NPW]; };
// xbytes: bytes of the row (K * 4), or of a pre-quantized image when the same registers carry one (img_finish_all) -- ONE producer of the register set for both
// cases: a struct assigned from two different calls under a run-time branch is demoted to scratch memory by hipcc (round 4's 48-byte frame, VERDICT weak 3)
template <int VW> __device__ __forceinline__ ActRegs<VW> act_issue_all(const void *x, unsigned xbytes, const float *nw, int K, int q) {
  ActRegs<VW> p;
  constexpr int P = 8 / VW;
  const int lane = lane_opaque();
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? (const void *)nw : x), (short)0, nw ? K * 4 : 0, 0x00020000);
  const int nv = (int)((xbytes + 8191u) / 8192u);
#pragma unroll
  for (int vw = 0; vw < VW; ++vw) {
    const unsigned off = (unsigned)((q + vw * P) * 64 + lane) * 16u;
#pragma unroll
    for (int j = 0; j < NPX; ++j) { p.xv[vw][j] = v4u{0u, 0u, 0u, 0u}; if (j < nv) p.xv[vw][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned)j * 8192u, 0, 0); }
  }
#pragma unroll
  for (int vw = 0; vw < VW; ++vw) {
    const unsigned off = (unsigned)((q + vw * P) * 64 + lane) * 16u;
#pragma unroll
    for (int j = 0; j < NPW; ++j) { p.wv[vw][j] = v4u{0u, 0u, 0u, 0u}; if (nw && j < nv) p.wv[vw][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, off + (unsigned)j * 8192u, 0, 0); }
  }
  return p;
}
// `red`: NCOLS * 8 floats of LDS.  Two halves so that weights can be requested between them (stream()):
//   act_sumsq_all     squares, virtual-wave sums -> red.  The CALLER's workgroup barrier follows (none needed when there is no norm weight)
//   act_quantize_all  norm factors from red (per column, read back from LDS: an array of factors indexed by a column loop that hipcc does not unroll -- 7 - 8
//                     columns -- would live in scratch memory), normalise + quantize the participant's superblocks into the image.  No trailing barrier.
template <int NCOLS, int VW>
__device__ __forceinline__ void act_sumsq_all(float *red, const ActRegs<VW> &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, int K, int q) {
  constexpr int P = 8 / VW;
  const int lane = lane_opaque();
  const int nv = (K + 2047) / 2048;
  if (!nw) return;
#pragma unroll
  for (int vw = 0; vw < VW; ++vw) {
    const int v = q + vw * P, vt = v * 64 + lane;
    auto ldx4 = [&](int c, int j) -> float4 { const int e = vt * 4 + j * 2048; return e < K ? *(const float4 *)(x + (size_t)c * ldx + e) : make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      float ss = 0.f;
      auto sq = [&](float4 f) { ss = fmaf(f.x, f.x, ss); ss = fmaf(f.y, f.y, ss); ss = fmaf(f.z, f.z, ss); ss = fmaf(f.w, f.w, ss); };
#pragma unroll
      for (int j = 0; j < NPX; ++j) if (j < nv) sq(c == 0 ? as_f4(pre.xv[vw][j]) : ldx4(c, j));
      for (int j = NPX; j < nv; ++j) sq(ldx4(c, j));
      ss = wave_sum_all(ss);
      if (lane == 0) red[c * 8 + v] = ss;
    }
  }
}
template <int NCOLS, int VW>
__device__ __forceinline__ void act_quantize_all(char *img, const float *red, const ActRegs<VW> &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, float eps, int K, int mode, int q,
                                                 int c0 = 0, int ncols_img = NCOLS) {  // column c of x -> column c0 + c of an image of ncols_img columns (dec_act_image_kernel: one workgroup per column)
  constexpr int P = 8 / VW;
  const int lane = lane_opaque();
  const int nv = (K + 2047) / 2048;
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    float nm = 1.0f, inv = 1.0f;
    if (nw) {
      const float *r = red + c * 8;
      const float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      nm = sqrtf(tot / (float)K + eps);
      inv = 1.0f / nm;
    }
#pragma unroll
    for (int vw = 0; vw < VW; ++vw) {
      const int v = q + vw * P, vt = v * 64 + lane;
      auto ldx4 = [&](int j) -> float4 { const int e = vt * 4 + j * 2048; return e < K ? *(const float4 *)(x + (size_t)c * ldx + e) : make_float4(0.f, 0.f, 0.f, 0.f); };
      auto ldw4 = [&](int j) -> float4 { const int e = vt * 4 + j * 2048; return e < K ? *(const float4 *)(nw + e) : make_float4(0.f, 0.f, 0.f, 0.f); };
      // the virtual wave's 256 values at 2048 j + 256 v = superblock v + 8 j; FOUR superblocks per quantizer call (four interleaved reduction chains: a call is a
      // ~150-instruction dependent chain per superblock, ~0.4 us for two -- down_proj's 7 superblocks per wave were 4 calls of two)
#pragma unroll
      for (int j0 = 0; j0 < NPX + 1; j0 += 4) {
        if (j0 < nv) {  // wave-uniform
          float4 vv[4]; int sb[4]; bool live[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int j = j0 + i;
            sb[i] = v + 8 * j; live[i] = j < nv && sb[i] * 256 < K;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NPX) { xv = c == 0 ? as_f4(pre.xv[vw][j]) : (live[i] ? ldx4(j) : xv); } else if (live[i]) xv = ldx4(j);
            if (nw) { if (j < NPW) w4 = as_f4(pre.wv[vw][j]); else if (live[i]) w4 = ldw4(j); }
            vv[i] = nw ? norm4(xv, w4, nm, inv) : xv;
          }
          if (j0 + 2 < nv) quantize_multi<4>(vv, sb, live, c0 + c, mode, img, K, ncols_img);
          else {  // one or two live: the two-chain body
            const float4 v2[2] = {vv[0], vv[1]}; const int s2[2] = {sb[0], sb[1]}; const bool l2[2] = {live[0], live[1]};
            quantize_multi<2>(v2, s2, l2, c0 + c, mode, img, K, ncols_img);
          }
        }
      }
      for (int j = NPX + 1; j < nv; ++j) {
        const int sb = v + 8 * j;
        if (sb * 256 < K) quantize_sb(nw ? norm4(ldx4(j), ldw4(j), nm, inv) : ldx4(j), sb, c0 + c, mode, img, K, ncols_img);
      }
    }
  }
}
// all 8 waves of a workgroup as participants, both halves with the barrier in between (the prompt path's quantizer kernel)
template <int NCOLS>
__device__ __forceinline__ void act_finish_all(char *img, float *red, const ActRegs<1> &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, float eps, int K, int mode, int q) {
  act_sumsq_all<NCOLS, 1>(red, pre, x, ldx, nw, K, q);
  if (nw) __syncthreads();
  act_quantize_all<NCOLS, 1>(img, red, pre, x, ldx, nw, eps, K, mode, q);
}
// a pre-quantized image (act_bytes(K, ncols) bytes, 16-byte aligned, written by a producer kernel; requested by act_issue_all) -> LDS: the 16-byte pieces at
// vt * 16 + j * 8192 of the participant's virtual threads
template <int VW> __device__ __forceinline__ void img_finish_all(char *smem, const ActRegs<VW> &pre, const void *img, size_t bytes, int q) {
  constexpr int P = 8 / VW;
  const int lane = lane_opaque();
#pragma unroll
  for (int vw = 0; vw < VW; ++vw) {
    const int vt = (q + vw * P) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NPX; ++j) if ((size_t)(vt * 16 + j * 8192) < bytes) *(v4u *)(smem + vt * 16 + j * 8192) = pre.xv[vw][j];
    for (size_t o = (size_t)vt * 16 + (size_t)NPX * 8192; o < bytes; o += 8192) *(v4u *)(smem + o) = *(const v4u *)((const char *)img + o);
  }
}

// ------------------------------------------------------------------------------------------------ per-format tiles
// Raw = the registers a lane holds for its QUARTER of a superblock (filled by buffer loads); term() turns the quad's four quarters into T_sb for NCOLS columns
// (every lane of the quad ends up with the same T).  Tile layouts (byte offsets inside a tile, L = lane, s = L >> 2 = superblock slot (r * 4 + p)):
#ifndef MRS_DEC2_LD_AUX
#define MRS_DEC2_LD_AUX 2  // nt: read once per token (6.8 vs 6.0 TB/s on a 430 MB stream, profiles/round5_decode.md section 2); 0 = default policy (experiments)
#endif
__device__ __forceinline__ v4u ldb128(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MRS_DEC2_LD_AUX); }
__device__ __forceinline__ v2u ldb64(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, MRS_DEC2_LD_AUX); }
__device__ __forceinline__ unsigned ldb32(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, MRS_DEC2_LD_AUX); }
__device__ __forceinline__ unsigned ldb16(__amdgpu_buffer_rsrc_t r, unsigned off) { return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, MRS_DEC2_LD_AUX); }
__device__ __forceinline__ int byte_of(unsigned w, int i) { return (int)((w >> (8 * i)) & 0xffu); }
__device__ __forceinline__ int sbyte_of(unsigned w, int i) { return (int)(int8_t)((w >> (8 * i)) & 0xffu); }

#ifndef MRS_DEC2_NS_Q4K
#define MRS_DEC2_NS_Q4K 4
#endif
#ifndef MRS_DEC2_NS_Q5K
#define MRS_DEC2_NS_Q5K 4
#endif
#ifndef MRS_DEC2_NS_Q6K
#define MRS_DEC2_NS_Q6K 4
#endif
#ifndef MRS_DEC2_NS_Q80
#define MRS_DEC2_NS_Q80 4
#endif

// operands of one (superblock, column) in LDS, for the lane's quarter c: the 64 int8 of the quarter, its four per-16 sums, the scale(s)
struct ActQ { int4 a0, a1, a2, a3; };
__device__ __forceinline__ ActQ act_quarter(const Act &act, int col, int qo, int) {  // qo: byte offset of the lane's quarter inside the column
  const char *qc = act.q + act.qcol(col) + qo;
  return ActQ{*(const int4 *)(qc), *(const int4 *)(qc + 16), *(const int4 *)(qc + 32), *(const int4 *)(qc + 48)};
}

template <int TYPE> struct Tile;

// Q4_K tile (2368 B): q0 [L][16] at 0, q1 [L][16] at 1024: GGUF qs bytes 32 c .. 32 c + 15 / + 16 .. + 31 of the superblock with bit 7 of every byte flipped (the
// high nibble then reads as the SIGNED nibble q - 8, times 16, through a plain mask); hs [L][4] at 2048 = {sc(2c), sc(2c+1), m(2c), m(2c+1)} (6-bit values as
// bytes); hd [s][4] at 2304 = d | dmin << 16 (f16 bits).  Quarter c = sub-block 2c (low nibbles, activations 64 c .. + 31) and 2c + 1 (high, + 32 .. + 63).
template <> struct Tile<T_Q4_K> {
  static constexpr int NS = MRS_DEC2_NS_Q4K;
  struct Raw { v4u q0, q1; unsigned hs, hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned tile, int lane) {
    Raw r;
    r.q0 = ldb128(rs, tile + (unsigned)lane * 16u);
    r.q1 = ldb128(rs, tile + 1024u + (unsigned)lane * 16u);
    r.hs = ldb32(rs, tile + 2048u + (unsigned)lane * 4u);
    r.hd = ldb32(rs, tile + 2304u + (unsigned)(lane >> 2) * 4u);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, int qo, int c, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    const v4u lo0 = w.q0 & 0x0F0F0F0Fu, hi0 = w.q0 & 0xF0F0F0F0u, lo1 = w.q1 & 0x0F0F0F0Fu, hi1 = w.q1 & 0xF0F0F0F0u;
    const int sca16 = byte_of(w.hs, 0) << 4, scb = byte_of(w.hs, 1), ma = byte_of(w.hs, 2), mb = byte_of(w.hs, 3);
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) {
      const ActQ a = act_quarter(act, col0 + k, qo, c);
      const int4 b = *(const int4 *)(act.bs + ((size_t)(col0 + k) * act.Sp + sb) * ACT_BS + 4 * c);  // runs 4c .. 4c+3
      const int bsa = b.x + b.y, bsb = b.z + b.w;                                                      // sub-blocks 2c, 2c+1
      const int dlo = dot16(lo1, a.a1, dot16(lo0, a.a0, 0)), dhi = dot16(hi1, a.a3, dot16(hi0, a.a2, 0));
      const int isum16 = quad_sum(__mul24(sca16, dlo) + __mul24(scb, dhi + (bsb << 7)));  // 16 sum q a = sum 16 (q - 8) a + 128 sum a
      const int msum = quad_sum(__mul24(ma, bsa) + __mul24(mb, bsb));
      const float yd = act.d[((size_t)(col0 + k) * act.Sp + sb) * act.dw];
      T[k] = fmaf(d * yd, (float)isum16 * 0.0625f, -((dmin * yd) * (float)msum));
    }
  }
};

// Q5_K tile (2880 B): q0 / q1 as Q4_K without the bit flip; xh [L][8] at 2048: dword h = the fifth bits of piece 2c + h: bit 8 j + k = bit of LOW-nibble weight 4 k + j
// of the piece (k = dword, j = byte), bit 8 j + 4 + k = of the HIGH-nibble weight; hs [L][4] at 2560, hd [s][4] at 2816 as Q4_K
template <> struct Tile<T_Q5_K> {
  static constexpr int NS = MRS_DEC2_NS_Q5K;
  struct Raw { v4u q0, q1; v2u xh; unsigned hs, hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned tile, int lane) {
    Raw r;
    r.q0 = ldb128(rs, tile + (unsigned)lane * 16u);
    r.q1 = ldb128(rs, tile + 1024u + (unsigned)lane * 16u);
    r.xh = ldb64(rs, tile + 2048u + (unsigned)lane * 8u);
    r.hs = ldb32(rs, tile + 2560u + (unsigned)lane * 4u);
    r.hd = ldb32(rs, tile + 2816u + (unsigned)(lane >> 2) * 4u);
    return r;
  }
  static __device__ __forceinline__ void fifth(v4u q, unsigned xh, v4u &lo, v4u &hi) {
    lo = q & 0x0F0F0F0Fu; hi = (q >> 4) & 0x0F0F0F0Fu;
    lo.x |= (xh & 0x01010101u) << 4; lo.y |= ((xh >> 1) & 0x01010101u) << 4; lo.z |= ((xh >> 2) & 0x01010101u) << 4; lo.w |= ((xh >> 3) & 0x01010101u) << 4;
    hi.x |= ((xh >> 4) & 0x01010101u) << 4; hi.y |= ((xh >> 5) & 0x01010101u) << 4; hi.z |= ((xh >> 6) & 0x01010101u) << 4; hi.w |= ((xh >> 7) & 0x01010101u) << 4;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, int qo, int c, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    v4u lo0, hi0, lo1, hi1;
    fifth(w.q0, w.xh.x, lo0, hi0);
    fifth(w.q1, w.xh.y, lo1, hi1);
    const int sca = byte_of(w.hs, 0), scb = byte_of(w.hs, 1), ma = byte_of(w.hs, 2), mb = byte_of(w.hs, 3);
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) {
      const ActQ a = act_quarter(act, col0 + k, qo, c);
      const int4 b = *(const int4 *)(act.bs + ((size_t)(col0 + k) * act.Sp + sb) * ACT_BS + 4 * c);
      const int bsa = b.x + b.y, bsb = b.z + b.w;
      const int dlo = dot16(lo1, a.a1, dot16(lo0, a.a0, 0)), dhi = dot16(hi1, a.a3, dot16(hi0, a.a2, 0));
      const int isum = quad_sum(__mul24(sca, dlo) + __mul24(scb, dhi));
      const int msum = quad_sum(__mul24(ma, bsa) + __mul24(mb, bsb));
      const float yd = act.d[((size_t)(col0 + k) * act.Sp + sb) * act.dw];
      T[k] = fmaf(d * yd, (float)isum, -((dmin * yd) * (float)msum));
    }
  }
};

// Q6_K tile (3360 B), weights in ELEMENT order (run r = elements 16 r .. 16 r + 15): ql0 [L][16] at 0: byte b = low 4 bits of weight b of run 4c (low nibble) and of
// run 4c + 1 (high nibble); ql1 at 1024: runs 4c + 2, 4c + 3; qh [L][16] at 2048: byte b = the top 2 bits of weight b of runs 4c .. 4c + 3 (bits 1:0, 3:2, 5:4, 7:6);
// sc [L][4] at 3072: the int8 scales of runs 4c .. 4c + 3; d [s][2] at 3328 (f16).  q - 32 is applied through the per-16 activation sums.
template <> struct Tile<T_Q6_K> {
  static constexpr int NS = MRS_DEC2_NS_Q6K;
  struct Raw { v4u ql0, ql1, qh; unsigned sc, hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned tile, int lane) {
    Raw r;
    r.ql0 = ldb128(rs, tile + (unsigned)lane * 16u);
    r.ql1 = ldb128(rs, tile + 1024u + (unsigned)lane * 16u);
    r.qh = ldb128(rs, tile + 2048u + (unsigned)lane * 16u);
    r.sc = ldb32(rs, tile + 3072u + (unsigned)lane * 4u);
    r.hd = ldb16(rs, tile + 3328u + (unsigned)(lane >> 2) * 2u);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, int qo, int c, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)w.hd);
    const v4u h = w.qh;
    const v4u q0 = (w.ql0 & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
    const v4u q1 = ((w.ql0 >> 4) & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
    const v4u q2 = (w.ql1 & 0x0F0F0F0Fu) | (h & 0x30303030u);
    const v4u q3 = ((w.ql1 >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
    const int s0 = sbyte_of(w.sc, 0), s1 = sbyte_of(w.sc, 1), s2 = sbyte_of(w.sc, 2), s3 = sbyte_of(w.sc, 3);
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) {
      const ActQ a = act_quarter(act, col0 + k, qo, c);
      const int4 b = *(const int4 *)(act.bs + ((size_t)(col0 + k) * act.Sp + sb) * ACT_BS + 4 * c);
      // sum sc <q - 32, u> = sum sc (<q, u> - 32 sum u), all integer
      int iq = __mul24(s0, dot16(q0, a.a0, 0) - 32 * b.x) + __mul24(s1, dot16(q1, a.a1, 0) - 32 * b.y);
      iq += __mul24(s2, dot16(q2, a.a2, 0) - 32 * b.z) + __mul24(s3, dot16(q3, a.a3, 0) - 32 * b.w);
      const int isum = quad_sum(iq);
      const float yd = act.d[((size_t)(col0 + k) * act.Sp + sb) * act.dw];
      T[k] = (d * yd) * (float)isum;
    }
  }
};

// Q8_0 "superblock" = 8 consecutive blocks of 32.  Tile (4352 B): q_i [L][16] at 1024 i (i = 0 .. 3) = the int8 quants of elements 64 c + 16 i .. + 15;
// dh [L][4] at 4096 = the f16 scales of blocks 2c (low half) and 2c + 1.  Activations: Q8_0 blocks.  T = the 8 block terms added in block order: a chain over the quad.
template <> struct Tile<T_Q8_0> {
  static constexpr int NS = MRS_DEC2_NS_Q80;
  struct Raw { v4u q0, q1, q2, q3; unsigned dh; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned tile, int lane) {
    Raw r;
    r.q0 = ldb128(rs, tile + (unsigned)lane * 16u);
    r.q1 = ldb128(rs, tile + 1024u + (unsigned)lane * 16u);
    r.q2 = ldb128(rs, tile + 2048u + (unsigned)lane * 16u);
    r.q3 = ldb128(rs, tile + 3072u + (unsigned)lane * 16u);
    r.dh = ldb32(rs, tile + 4096u + (unsigned)lane * 4u);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, int qo, int c, const Act &act, int col0, float (&T)[NCOLS]) {
    const float dwa = half_bits_to_float((uint16_t)(w.dh & 0xffff)), dwb = half_bits_to_float((uint16_t)(w.dh >> 16));
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) {
      const ActQ a = act_quarter(act, col0 + k, qo, c);
      const float2 dx = *(const float2 *)(act.d + ((size_t)(col0 + k) * act.Sp + sb) * act.dw + 2 * c);
      const int isa = dot16(w.q1, a.a1, dot16(w.q0, a.a0, 0)), isb = dot16(w.q3, a.a3, dot16(w.q2, a.a2, 0));
      const float pa = (float)isa * dwa * dx.x, pb = (float)isb * dwb * dx.y;
      float r = pa + pb;  // quarter 0: t = p0; t = t + p1
#pragma unroll
      for (int step = 1; step < 4; ++step) {
        const float prev = dppf<0x90>(r);  // quad_perm [0, 0, 1, 2]: the running sum of the quarter before
        const float cand = (prev + pa) + pb;
        r = c == step ? cand : r;
      }
      T[k] = dppf<0xFF>(r);  // quad_perm [3, 3, 3, 3]
    }
  }
};

// T for NCOLS activation columns, two at a time: the weight registers are unpacked once per pair and the LDS operands of a pair are in flight together
template <int TYPE, int NCOLS, int C0> struct TermCols {
  static __device__ __forceinline__ void run(const typename Tile<TYPE>::Raw &w, int sb, int qo, int c, const Act &act, int col0, float (&T)[NCOLS]) {
    if constexpr (C0 < NCOLS) {
      if constexpr (NCOLS - C0 >= 2) {
        float t2[2];
        Tile<TYPE>::template term<2>(w, sb, qo, c, act, col0 + C0, t2);
        T[C0] = t2[0]; T[C0 + 1] = t2[1];
      } else {
        float t1[1];
        Tile<TYPE>::template term<1>(w, sb, qo, c, act, col0 + C0, t1);
        T[C0] = t1[0];
      }
      if constexpr (C0 + 2 < NCOLS && (C0 & 2) != 0) __builtin_amdgcn_sched_barrier(0);  // four columns at a time: their LDS operands are in flight together (one pair at a time exposed an LDS round trip per pair: batch 8 was latency-bound), more would not fit the registers
      TermCols<TYPE, NCOLS, C0 + 2>::run(w, sb, qo, c, act, col0, T);
    }
  }
};

// ------------------------------------------------------------------------------------------------ the streaming core
// A workgroup owns the UNITS [u0, u1) of a launch; unit u = rgpu consecutive record groups (a record group = 4 consecutive rows = Cs tiles) of each of the launch's
// nseg tensors (gate and up rows of the same index travel together).  Wave w takes units u0 + w, u0 + w + NW, ...: T tiles, known up front; it keeps NS tiles
// requested ahead of the one it is computing.  epi(seg, row0, nvalid, rgl, sums, aux) is called once per finished record group; inside a unit: segment 0 before
// segment 1, record groups ascending; the call is lane-parallel: every lane of the chunk-3 quad of row rr = lane / 16 of the group (lanes 16 rr + 12 .. + 15) holds
// that row's sum, lane 16 rr + 12 is the row's owner, nvalid rows exist.  aux(row0) runs with every REQUEST of the group's tiles (operands of the epilogue --
// residual values, RoPE factors -- travel with the weights instead of costing a dependent load after the row sum); the copy that came with the last tile comes back to epi.
//
// What the MI355X measurements of round 5 say (profiles/round5_decode.md, profiles/experiments/stream_probe.hip):
//   * a launch that only streams (no prologue, trivial arithmetic) needs 3.0 / 3.8 / 7.5 / 12.2 / 66 us for 9 / 14 / 40 / 66 / 430 MB, boundary included, and FOUR
//     1-KiB loads in flight per wave with 8 waves per CU reach that; deeper rings are slower on the short launches (the memory system accepts a CU's requests at its HBM
//     share: a wave that requests more is blocked in the issue and cannot do anything else);
//   * an out-of-range ("dead") buffer load still costs its 16 cycles in the CU's one texture addresser: a pass of dead requests of 8 waves is ~0.5 us;
//   * nothing can be multiplied before the activation prologue is done, and the prologue's workgroup barrier waits for the SLOWEST wave's blocked issue.
// Hence: (1) the activation row is requested first, ONE tile per wave next, the sum of squares and its barrier run while those fly, then the rest of the ring, then
// the quantization and the publishing barrier; (2) the ring is NS = 4 tiles; (3) no dead requests in the common case: passes whose NS re-requests are all live run in
// a loop, the last tiles are computed by straight-line tails that request nothing (or only the r < NS live ones that are left + NS - r dead).
struct Job {
  Mat mat[2];
  int nseg, rgpu;
  int nrows;            // rows per tensor (per expert slot) that take part
  int u0, u1;           // this workgroup's units
  const int32_t *sel;   // stacked experts [E * rows][K]: device array of expert ids per slot, or nullptr (dense)
  int sel_mode;         // 1: slot = unit / upe (all top-k experts of a token in one launch; upe = every unit when there is one slot);  2: slot = segment
  int upe;              // units per expert slot
  int ergs;             // record groups per expert in the stacked tensor
  unsigned long long *tl;  // experiments: 16 s_memrealtime stamps (100 MHz) per wave, or nullptr
};
#define MRS_TL2(jb, i) do { if ((jb).tl && (tid_opaque() & 63) == 0) (jb).tl[((size_t)blockIdx.x * NW + (tid_opaque() >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the lane of a record group's row rr that owns the row sum in the epilogue (and loads the row's epilogue operands): 16 rr + owner_off
__host__ __device__ inline int owner_off(const Geo &) { return 12; }
// what a ring slot remembers of its tile (scalars): ts | seg << 24, or -1 for "nothing live was requested"; the first row of its record group in the launch's row
// numbering (slot * rows-per-slot + local row); the number of rows of the group that exist
struct RecMeta { int ts_seg, row0, nvalid; };
struct NoAux {};

// SEGCOL (NCOLS must be 1): segment s multiplies by activation column s of a 2-column image (MoE down: two experts' rows against their own activations)
// stage(0, q): request the activation row / image;  stage(1, q): squares -> red (nothing for an image);  stage(2, q): normalise + quantize / copy into LDS --
// every wave is participant q = wave of the prologue (one virtual wave each);  sbar: the prologue has a sum-of-squares barrier (a norm weight)
// RING2: a ring of 2 tiles instead of the format's NS (4): launches in which all 8 waves of a workgroup stream (>= 8 units per workgroup) want ~37 KB per CU in
// flight -- with 4 tiles per wave the waves are blocked in the issue of the 3 tiles behind the sum-of-squares barrier for ~1.7 us before they can quantize
// (gate + up 16.6 -> 16.0 us); launches with 4 streaming waves (down_proj, o_proj) want the 4 (down_proj 12.8 vs 15.1 us with 2): profiles/round5_decode.md
template <int TYPE, int NCOLS, bool SEGCOL = false, bool RING2 = false, class Stage, class AuxF, class Epi>
__device__ __forceinline__ void stream(const Job &jb, int K, int ncols_img, int mode, char *smem, int *ctr, bool sbar, Stage stage, AuxF auxf, Epi epi) {
  using TL = Tile<TYPE>;
  using AuxT = decltype(auxf(0));
  constexpr int NS = (NCOLS <= 4 && !RING2) ? TL::NS : (TL::NS > 2 ? 2 : TL::NS);  // wide batches: a tile's arithmetic is NCOLS times longer, a short ring covers the same time (and fits the registers)
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = geo_for(K);
  const int Cs = g.Cs;
  constexpr unsigned tileb = tile_bytes(TYPE);
  const int p = (lane >> 2) & 3, c = lane & 3;
  // the wave's share: units u0 + wave, + NW, ...
  const int nun = jb.u1 - jb.u0 - wave;
  const int myunits = nun > 0 ? (nun + NW - 1) / NW : 0;
  const int tps = jb.rgpu * Cs;  // tiles per segment of a unit
  const int T = myunits * jb.nseg * tps;
  // Units are handed out by a counter in LDS when a unit is a whole number of ring passes (every K = 4096 / 8192 shape): the waves of a workgroup do not run at the
  // same pace (the second wave of a SIMD loses the issue arbitration to the first: lm_head's waves 4 .. 7 finished 9 us after waves 0 .. 3 with equal static shares),
  // and a wave that is ahead simply takes the next unit.  Otherwise (down_proj at K = 14336: 14 tiles per unit, one unit per wave) the share is static.
  const int tpu = jb.nseg * tps;
  const bool dyn = tpu % NS == 0;
  if (tid == 0) *ctr = jb.u0 + NW;  // published by barrier A
  if (jb.u0 + wave >= jb.u1) {  // a wave without tiles (small launches): its share of the prologue and the barriers, no requests at all
    stage(0, wave);
    __syncthreads();  // A
    stage(1, wave);
    if (sbar) __syncthreads();  // S
    stage(2, wave);
    __syncthreads();  // B
    return;
  }
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)jb.mat[0].base, (short)0, (int)jb.mat[0].bytes, 0x00020000);
  const bool two = jb.nseg > 1;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(two ? jb.mat[1].base : jb.mat[0].base), (short)0, (int)(two ? jb.mat[1].bytes : jb.mat[0].bytes), 0x00020000);
  typename TL::Raw ring[NS];
  RecMeta meta[NS];
  AuxT auxv[NS];
  // expert ids of the launch's slots (MoE), read ONCE, before the first store of the kernel, as scalars: a load of sel[] inside the request loop is a VMEM load
  // (the compiler cannot prove that the epilogue's stores leave it alone), and waiting for it is `s_waitcnt vmcnt(0)` -- it would drain the whole ring at every request
  // (it did, in rounds 3-4: one tile in flight per wave whatever the ring depth).  Eight 8-bit ids in one 64-bit scalar: an array would be indexed in scratch memory,
  // i.e. through vmcnt again; the launchers refuse > 256 experts.
  unsigned long long selp = 0ull;
  const bool moe = jb.sel != nullptr;
  if (moe) {
    const int nsl = jb.sel_mode == 2 ? jb.nseg : min(8, (jb.u1 + jb.upe - 1) / jb.upe);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < nsl) selp |= (unsigned long long)(unsigned)(__builtin_amdgcn_readfirstlane(jb.sel[i]) & 0xff) << (8 * i);
  }
  auto sel_at = [&](int i) { return (int)((selp >> (8 * (i & 7))) & 0xffull); };
  // the wave's request cursor (all wave-uniform): the unit, the segment, tiles left in the segment, the byte offset of the next tile (tiles of a segment of a unit
  // are consecutive in memory), the group's first row and the tile index inside the group
  int lunit = 0, lseg = 0, lleft = 0, lts = 0, lrow0 = 0, lslot_row0 = 0;
  unsigned ltoff = 0, lubase1 = 0;
  auto open_unit = [&](int u) {  // wave-uniform; u may be past the end (the cursor then stays dead)
    lunit = u; lseg = 0; lts = 0; lleft = tps;
    if (u < jb.u1) {
      int local = u * jb.rgpu, e0 = 0, e1 = 0, sl = 0;
      if (moe) {
        if (jb.sel_mode == 2) { e0 = sel_at(0); e1 = sel_at(1); }
        else { sl = u / jb.upe; e0 = e1 = sel_at(sl); local -= sl * jb.upe * jb.rgpu; }
      }
      ltoff = (unsigned)(e0 * jb.ergs + local) * (unsigned)Cs * tileb;
      lubase1 = (unsigned)(e1 * jb.ergs + local) * (unsigned)Cs * tileb;
      lrow0 = u * jb.rgpu * g.R;
      lslot_row0 = sl * (jb.upe * jb.rgpu * g.R);
    }
  };
  // Every request is UNCONDITIONAL (a cursor past its last tile asks for an out-of-range offset: zeros, no traffic) and every tile asks for the same number of
  // loads: hipcc's s_waitcnt insertion merges the counter state of control-flow paths conservatively, so ONE conditional load between a tile's request and its use
  // makes the wait for that tile stricter by one, and a conditional request of a whole tile (rounds 3-4) collapses every wait to vmcnt(0) -- the ring then holds one
  // tile in flight whatever its depth.  With straight-line requests the waits come out exact: vmcnt((NS - 1) x loads per tile).
  constexpr unsigned DEAD = 0xF0000000u;  // beyond every tensor (make_mat refuses tensors of 0xF0000000 bytes and more); + the tile's plane offsets: no wrap
  auto issue = [&](typename TL::Raw &slot, RecMeta &m, AuxT &ax) {
    const bool livel = lunit < jb.u1;
    const int lrow = lrow0 - lslot_row0;  // local row inside the expert slot
    m = RecMeta{livel ? (lts | (lseg << 24)) : -1, lrow0, min(g.R, jb.nrows - lrow)};
    slot = TL::load(lseg == 0 ? rs0 : rs1, livel ? ltoff : DEAD, lane);
    ax = auxf(livel ? lrow0 : 0);
    if (livel) {  // scalar bookkeeping only: no memory instruction under this branch
      ltoff += tileb;
      if (++lts == Cs) { lts = 0; lrow0 += g.R; }
      if (--lleft == 0) {
        if (lseg + 1 < jb.nseg) { lseg = 1; lleft = tps; ltoff = lubase1; lrow0 -= jb.rgpu * g.R; }
        else if (dyn) {  // next unit from the workgroup's counter (LDS atomic: lgkmcnt, not vmcnt)
          int nu = 0;
          if (lane == 0) nu = atomicAdd(ctr, 1);
          open_unit(__builtin_amdgcn_readfirstlane(nu));
        } else open_unit(lunit + NW);
      }
    }
  };
  open_unit(jb.u0 + wave);
  stage(0, wave);  // the activation row (or its image) -> registers: requested before any weights
  MRS_TL2(jb, 0);
  __syncthreads();  // A (publishes the unit counter)
  issue(ring[0], meta[0], auxv[0]);
  stage(1, wave);  // squares, wave sums -- while the first tile of every wave is in flight
  if (sbar) __syncthreads();  // S
  MRS_TL2(jb, 1);
  // (publishing the image BEFORE the rest of the ring is requested -- tried for the short launches, round 5 -- made qkv 0.9 us slower in the graph: the requests that
  // would have overlapped the quantization arithmetic then queue behind it)
#pragma unroll
  for (int i = 1; i < NS; ++i) issue(ring[i], meta[i], auxv[i]);
  stage(2, wave);  // normalise + quantize (or copy the image) into LDS
  MRS_TL2(jb, 2);
  __syncthreads();  // B
  MRS_TL2(jb, 3);
  const Act act = act_view(smem, K, ncols_img, mode);
  // per-lane constants of the LDS operands: chunk p, quarter c
  const int pCs = p * Cs;
  const int qo_lane = p * act.CQ + c * 64;
  float acc[NCOLS];
#pragma unroll
  for (int k = 0; k < NCOLS; ++k) acc[k] = 0.0f;
  auto compute = [&](int i) {  // the tile in slot i: T_sb of the lane's superblock, the chunk sum, the group's epilogue after its last tile
    const int ts = meta[i].ts_seg & 0xffffff, seg = meta[i].ts_seg >> 24;
    const int sb = pCs + ts;
    const bool live = sb < g.S;  // a chunk of the last quarter may be short (or empty): its slots are zeros in memory and take no part in the sum
    float Tm[NCOLS];
    TermCols<TYPE, NCOLS, 0>::run(ring[i], sb, qo_lane + ts * 256, c, act, SEGCOL ? seg : 0, Tm);
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) acc[k] = ts == 0 ? (live ? Tm[k] : 0.0f) : (live ? acc[k] + Tm[k] : acc[k]);  // c_p: left to right inside the chunk
    if (ts == Cs - 1) {
      // the four chunk sums sit in the four quads of the row's 16 lanes (a DPP row); three row_ror:4 steps leave ((c0 + c1) + c2) + c3 in the chunk-3 quad
      float tot[NCOLS];
#pragma unroll
      for (int k = 0; k < NCOLS; ++k) {
        float u = dppf<0x124>(acc[k]) + acc[k];
        u = dppf<0x124>(u) + acc[k];
        tot[k] = dppf<0x124>(u) + acc[k];
      }
      epi(seg, meta[i].row0, meta[i].nvalid, (meta[i].row0 / g.R) & (jb.rgpu - 1), tot, auxv[i]);  // rgpu is 1 or 2
    }
  };
  if (dyn) {
    // a pass = NS tiles of ONE unit: its re-requests are all live or there is nothing left -- no dead requests, whatever the number of units the wave ends up with.
    // (One loop exit, at the latch: every exit of a loop is funnelled through the latch block by hipcc, and counter states merged there would make the header's
    // first wait a full drain.)
    while (lunit < jb.u1) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { compute(i); issue(ring[i], meta[i], auxv[i]); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) compute(i);  // the NS tiles in flight
  } else {
    // static share: passes whose NS re-requests are all live, then straight-line tails
    const int rest = T > NS ? T - NS : 0;  // tiles not yet requested
    const int full = rest / NS;
    for (int ps = 0; ps < full; ++ps) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { compute(i); issue(ring[i], meta[i], auxv[i]); }
    }
    if (rest - full * NS > 0) {  // r = 1 .. NS - 1 live requests left: one more pass (its last NS - r requests are dead), then the tail
#pragma unroll
      for (int i = 0; i < NS; ++i) { compute(i); issue(ring[i], meta[i], auxv[i]); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) if (meta[i].ts_seg >= 0) compute(i);
  }
  MRS_TL2(jb, 14);
}

}  // namespace dec2
}  // namespace mrs
