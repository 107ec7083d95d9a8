// dec_core2.cuh -- MI355X decode engine, GEMV core (round 4): one lane = one whole superblock.
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
// MI355X design.  Round 3's core gave a lane 16 bytes of a superblock: 8 lanes shared one superblock, every lane
// decoded scales, converted, multiplied and reduced in f32 -- ~60 wave instructions per KiB of weights, and the kernels were issue / latency bound
// (weights resident in the Infinity Cache ran no faster).  Here:
//   * weights live in a DECODE LAYOUT made once at load time (mrs_dec_repack): the tensor is cut into RECORDS of up to 64 superblocks = one wave
//     instruction's worth: R consecutive rows x 4 chunks x W superblocks; inside a record piece i of all its superblocks is contiguous, so a
//     `buffer_load_dwordx4` of 64 lanes still reads up to 1 KiB of consecutive bytes, but after the record's 10 (Q4_K) loads lane l owns superblock l
//     completely: 8 sub-block dots, integer scale / min combination, ONE f32 term -- ~22 wave instructions per KiB;
//   * a wave has 1 .. 16 records per launch: the first NS are requested before anything else happens, the rest through a ring of NS register sets;
//   * the activation prologue (RMSNorm + Q8_K quantization of the row) runs on waves 0 .. 3 only, while waves 4 .. 7 are already blocked on the issue of
//     their weight loads (the memory system accepts requests at HBM rate): the prologue no longer sits between "ring issued" and "first tile computed";
//   * activations in LDS: int8 per superblock at a stride of 272 bytes (the 16 lanes of a ds_read_b128 group then hit 16 different bank groups with a
//     compile-time piece offset), f32 block scales, int32 per-16 sums at a stride of 80 bytes;
//   * row sums: W - 1 `v_add_f32 row_ror:1` steps inside the chunk (lane W - 1 of the group ends up with the chunk's left-to-right sum), then the four
//     chunk sums through ds_bpermute.
#pragma once
#include "gguf_blocks.cuh"
#include <type_traits>

#ifndef MRS_WAVE_SYNC
#define MRS_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

namespace mrs {
namespace dec2 {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#ifndef MRS_DEC2_NT
#define MRS_DEC2_NT 512
#endif
constexpr int NT = MRS_DEC2_NT, NW = NT / 64;  // threads / waves per workgroup (one workgroup per CU): 512 or 1024; the prologue always runs on the first 8 (or PW) waves
constexpr int PW = 4;                  // prologue waves (0 .. PW-1)
constexpr unsigned OOB = 0xFFFFFF00u;  // buffer offset out of range for every tensor: the load returns zeros and costs no traffic

__host__ __device__ inline bool dec_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }
enum : int { ACT_Q8K = 0, ACT_Q80 = 1 };
__host__ __device__ inline int act_mode_for(int type) { return type == T_Q8_0 ? ACT_Q80 : ACT_Q8K; }

// ------------------------------------------------------------------------------------------------ geometry
// S superblocks per row, Cs per chunk, LPC lanes per chunk (power of two <= 16), TPC records ("tile steps") per chunk, W superblocks per chunk and
// record, R rows per record, A = R * 4 * W stored slots per record.
struct Geo { int S, Cs, LPC, TPC, W, R, A; };
__host__ __device__ inline Geo geo_for(int K) {
  Geo g;
  g.S = K / 256;
  g.Cs = (g.S + 3) / 4;
  int lpc = 1;
  while (lpc < g.Cs && lpc < 16) lpc <<= 1;
  g.LPC = lpc;
  g.TPC = (g.Cs + lpc - 1) / lpc;
  g.W = (g.Cs + g.TPC - 1) / g.TPC;
  g.R = 16 / lpc;
  g.A = g.R * 4 * g.W;
  return g;
}
// bytes of one superblock in the decode layout
__host__ __device__ constexpr int slot_bytes(int type) { return type == T_Q4_K ? 148 : type == T_Q5_K ? 180 : type == T_Q6_K ? 210 : 272; }
__host__ __device__ inline size_t rec_bytes(int type, const Geo &g) { return ((size_t)g.A * slot_bytes(type) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t tensor_bytes(int type, long long n, long long k) {
  const Geo g = geo_for((int)k);
  const size_t rgs = (size_t)((n + g.R - 1) / g.R);
  return rgs * g.TPC * rec_bytes(type, g);
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

// ------------------------------------------------------------------------------------------------ activations in LDS (and, byte for byte, the pre-quantized image
// a producer kernel leaves in global memory).  ncols columns of K values:
//   q  [ncols][S][272]  int8 of superblock sb in element order (16 pieces of 16 bytes; 16 pad bytes)
//   d  [ncols][S][DW]   f32: Q8_K mode DW = 1: d of the superblock;  Q8_0 mode DW = 12: f32(f16(d)) of the 8 blocks (4 pad floats)
//   bs [ncols][S][20]   int32 sums of the 16 runs of 16 (Q8_K mode; 4 pad ints)
constexpr int ACT_QS = 272, ACT_BS = 20;
__host__ __device__ inline int act_dw(int mode) { return mode == ACT_Q80 ? 12 : 1; }
__host__ __device__ inline size_t act_bytes(int K, int ncols) { return (size_t)ncols * (size_t)(K / 256) * (ACT_QS + 48 + ACT_BS * 4); }  // both modes fit
struct Act {
  const char *q;
  const float *d;
  const int *bs;
  int K, S, dw;
};
__device__ __forceinline__ Act act_view(char *smem, int K, int ncols, int mode) {
  const int S = K / 256, dw = act_dw(mode);
  char *d = smem + (size_t)ncols * S * ACT_QS;
  return Act{smem, (const float *)d, (const int *)(d + (size_t)ncols * S * dw * 4), K, S, dw};
}

// quantize N superblocks at once: lane l holds elements 4 l .. 4 l + 3 of each (v[n], superblock sb[n], live[n] wave-uniform) -> column c of the image.
// Every step is a loop over n so that the N reduction chains interleave (one chain alone is ~100 dependent instructions).
// Quantizers: candle BlockQ8K::from_float (amax with its sign, iscale = -128 / max, q = min(127, round(iscale x)), d = 1 / iscale, the FIRST
// element of largest magnitude wins) and quantize_row_q8_0 (d = amax / 127, q = round(x / d), d kept as f16) -- oracle/ggml_oracle.c.
template <int N>
__device__ __forceinline__ void quantize_multi(const float4 (&v)[N], const int (&sb)[N], const bool (&live)[N], int c, int mode, char *img, int K, int ncols) {
  const int lane = lane_opaque(), S = K / 256, dw = act_dw(mode);
  char *q0p = img + (size_t)c * S * ACT_QS;
  float *d0p = (float *)(img + (size_t)ncols * S * ACT_QS) + (size_t)c * S * dw;
  int *b0p = (int *)(img + (size_t)ncols * S * ACT_QS + (size_t)ncols * S * dw * 4) + (size_t)c * S * ACT_BS;
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
        *(int *)(q0p + (size_t)sb[n] * ACT_QS + lane * 4) = packed;
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
        *(int *)(q0p + (size_t)sb[n] * ACT_QS + lane * 4) = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
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
// Sum of squares, every path: 512 "virtual threads", thread t sums the squares of its float4 pieces 4 t + 2048 j (j ascending, x y z w, fma); wave_sum_all per
// 64 threads; the 8 wave sums as ((0+1)+(2+3))+((4+5)+(6+7)) -- the order the engine has had since round 2 (oracle: orc_rms_norm_engine).
// A CU has ONE in-order memory pipe: whatever the prologue needs is requested before any wave of the workgroup requests weights (the caller's first
// barrier sits between the *_issue and the weight requests).  Two schedules with the same bits:
//   ALL   (small launches) all 8 waves load their own pieces (act_issue_all), request weight records, then act_finish_all squares / reduces through `red` +
//         one workgroup barrier, and wave w quantizes superblocks w, w + 8, ... (two at a time)
//   SPEC  (launches whose weight requests keep the memory pipe busy for microseconds: a wave is blocked on their issue) waves 0 .. PW-1 request the row
//         (act_issue_spec), and run act_finish_spec while waves PW .. 7 request weights: every prologue wave computes the whole sum of squares itself (no
//         workgroup barrier), then quantizes superblocks w, w + PW, ... (four at a time); its own weight requests come after the prologue.
constexpr int MAXP = 8;  // register-resident float4 pieces per thread (ALL): rows of <= 16384 values; longer rows take the rest with plain loads
template <int NP> struct ActRegs { v4u xv[NP]; v4u wv[NP]; };
template <int NP> __device__ __forceinline__ ActRegs<NP> act_issue_all(const float *x, const float *nw, int K) {
  ActRegs<NP> p;
  const unsigned off = (unsigned)tid_opaque() * 16u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? nw : x), (short)0, nw ? K * 4 : 0, 0x00020000);
  const int nv = __builtin_amdgcn_readfirstlane(tid_opaque() >> 6) < 8 ? (K + 2047) / 2048 : 0;  // waves 8 .. 15 of a 1024-thread workgroup take no part
#pragma unroll
  for (int j = 0; j < NP; ++j) { p.xv[j] = v4u{0u, 0u, 0u, 0u}; if (j < nv) p.xv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned)j * 8192u, 0, 0); }
#pragma unroll
  for (int j = 0; j < NP; ++j) { p.wv[j] = v4u{0u, 0u, 0u, 0u}; if (nw && j < nv) p.wv[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, off + (unsigned)j * 8192u, 0, 0); }
  return p;
}
// `red`: NCOLS * 8 floats of LDS.  No trailing barrier (the caller's barrier publishes the image).
template <int NCOLS, int NP>
__device__ __forceinline__ void act_finish_all(char *img, float *red, const ActRegs<NP> &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, float eps, int K, int mode) {
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nv = wave < 8 ? (K + 2047) / 2048 : 0;  // waves 8 .. 15 of a 1024-thread workgroup only pass the barrier
  float nm[NCOLS], inv[NCOLS];
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) { nm[c] = 1.0f; inv[c] = 1.0f; }
  auto ldx4 = [&](int c, int j) -> float4 { const int e = tid * 4 + j * 2048; return e < K ? *(const float4 *)(x + (size_t)c * ldx + e) : make_float4(0.f, 0.f, 0.f, 0.f); };
  auto ldw4 = [&](int j) -> float4 { const int e = tid * 4 + j * 2048; return e < K ? *(const float4 *)(nw + e) : make_float4(0.f, 0.f, 0.f, 0.f); };
  if (nw) {
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      float ss = 0.f;
      auto sq = [&](float4 f) { ss = fmaf(f.x, f.x, ss); ss = fmaf(f.y, f.y, ss); ss = fmaf(f.z, f.z, ss); ss = fmaf(f.w, f.w, ss); };
#pragma unroll
      for (int j = 0; j < NP; ++j) if (j < nv) sq(c == 0 ? as_f4(pre.xv[j]) : ldx4(c, j));
      for (int j = NP; j < nv; ++j) sq(ldx4(c, j));
      ss = wave_sum_all(ss);
      if (lane == 0 && wave < 8) red[c * 8 + wave] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const float *r = red + c * 8;
      const float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      nm[c] = sqrtf(tot / (float)K + eps);
      inv[c] = 1.0f / nm[c];
    }
  }
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    // the wave's 256 values at 2048 j + 256 wave = superblock wave + 8 j; two superblocks per quantizer call
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += 2) {
      if (j0 < nv) {  // wave-uniform
        float4 v[2]; int sb[2]; bool live[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int j = j0 + i;
          sb[i] = wave + 8 * j; live[i] = j < nv && sb[i] * 256 < K;
          const float4 xv = c == 0 ? as_f4(pre.xv[j]) : (live[i] ? ldx4(c, j) : make_float4(0.f, 0.f, 0.f, 0.f));
          v[i] = nw ? norm4(xv, as_f4(pre.wv[j]), nm[c], inv[c]) : xv;
        }
        quantize_multi<2>(v, sb, live, c, mode, img, K, NCOLS);
      }
    }
    for (int j = NP; j < nv; ++j) {
      const int sb = wave + 8 * j;
      if (sb * 256 < K) quantize_sb(nw ? norm4(ldx4(c, j), ldw4(j), nm[c], inv[c]) : ldx4(c, j), sb, c, mode, img, K, NCOLS);
    }
  }
}
// SPEC: waves 0 .. PW-1 only.  Registers: the whole row for the sum of squares (16 pieces per lane and batch) + the wave's own superblocks (<= 16).
constexpr int SPEC_OWN = 16;
struct SpecRegs { v4u xa[16]; v4u xo[SPEC_OWN]; v4u wo[SPEC_OWN]; };
__device__ __forceinline__ SpecRegs act_issue_spec(const float *x, const float *nw, int K, int wave) {
  SpecRegs p;
  const int lane = lane_opaque(), S = K / 256;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? nw : x), (short)0, nw ? K * 4 : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // virtual thread v * 64 + lane, piece jj: element (v * 64 + lane) * 4 + 2048 jj;  i = 8 jj + v  (first 4096 values)
    p.xa[i] = v4u{0u, 0u, 0u, 0u};
    if (nw && (i >> 3) * 2048 < K) p.xa[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(((i & 7) * 64 + lane) * 16 + (i >> 3) * 8192), 0, 0);
  }
#pragma unroll
  for (int i = 0; i < SPEC_OWN; ++i) {
    p.xo[i] = v4u{0u, 0u, 0u, 0u}; p.wo[i] = v4u{0u, 0u, 0u, 0u};
    const int sb = wave + i * PW;
    if (sb < S) {  // wave-uniform
      p.xo[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(sb * 1024 + lane * 16), 0, 0);
      if (nw) p.wo[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(sb * 1024 + lane * 16), 0, 0);
    }
  }
  return p;
}
template <int NCOLS>
__device__ __forceinline__ void act_finish_spec(char *img, const SpecRegs &pre, const float *__restrict__ x, int ldx, const float *__restrict__ nw, float eps, int K, int mode, int wave) {
  const int lane = lane_opaque(), S = K / 256;
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    const float *xr = x + (size_t)c * ldx;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)xr, (short)0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(nw ? nw : xr), (short)0, nw ? K * 4 : 0, 0x00020000);
    float nm = 1.0f, inv = 1.0f;
    if (nw) {
      float ss[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) ss[v] = 0.f;
      for (int j0 = 0; j0 * 2048 < K; j0 += 2) {  // two pieces (16 loads) per batch
        v4u f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (c == 0 && j0 == 0) f[i] = pre.xa[i];
          else { f[i] = v4u{0u, 0u, 0u, 0u}; if ((j0 + (i >> 3)) * 2048 < K) f[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(((i & 7) * 64 + lane) * 16 + (j0 + (i >> 3)) * 8192), 0, 0); }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float4 g = as_f4(f[i]); float &s1 = ss[i & 7]; s1 = fmaf(g.x, g.x, s1); s1 = fmaf(g.y, g.y, s1); s1 = fmaf(g.z, g.z, s1); s1 = fmaf(g.w, g.w, s1); }
      }
      float ws[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) ws[v] = wave_sum_all(ss[v]);
      const float tot = ((ws[0] + ws[1]) + (ws[2] + ws[3])) + ((ws[4] + ws[5]) + (ws[6] + ws[7]));
      nm = sqrtf(tot / (float)K + eps);
      inv = 1.0f / nm;
    }
    // own superblocks wave + i PW, four per quantizer call
#pragma unroll
    for (int i0 = 0; i0 < SPEC_OWN; i0 += 4) {
      if (wave + i0 * PW < S) {  // wave-uniform
        float4 v[4]; int sb[4]; bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k;
          sb[k] = wave + i * PW; live[k] = sb[k] < S;
          v4u xr4 = pre.xo[i], wr4 = pre.wo[i];
          if (c != 0) {
            xr4 = v4u{0u, 0u, 0u, 0u};
            if (live[k]) xr4 = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(sb[k] * 1024 + lane * 16), 0, 0);
          }
          v[k] = nw ? norm4(as_f4(xr4), as_f4(wr4), nm, inv) : as_f4(xr4);
        }
        quantize_multi<4>(v, sb, live, c, mode, img, K, NCOLS);
      }
    }
    for (int sbx = wave + SPEC_OWN * PW; sbx < S; sbx += PW) {  // rows beyond 16384 values
      const float4 xv = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(sbx * 1024 + lane * 16), 0, 0));
      const float4 w4 = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(sbx * 1024 + lane * 16), 0, 0));
      quantize_sb(nw ? norm4(xv, w4, nm, inv) : xv, sbx, c, mode, img, K, NCOLS);
    }
  }
}
// a pre-quantized image (act_bytes(K, ncols) bytes, 16-byte aligned, written by a producer kernel) -> LDS: the 16-byte pieces at tid * 16 + j * 8192
template <int NP> __device__ __forceinline__ ActRegs<NP> img_issue_all(const void *img, size_t bytes) {
  ActRegs<NP> p;
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)img, (short)0, (int)bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < NP; ++j) { p.xv[j] = v4u{0u, 0u, 0u, 0u}; if ((size_t)j * 8192 < bytes) p.xv[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, (unsigned)tid_opaque() * 16u + (unsigned)j * 8192u, 0, 0); }
  return p;
}
template <int NP> __device__ __forceinline__ void img_finish_all(char *smem, const ActRegs<NP> &pre, const void *img, size_t bytes) {
  const int tid = tid_opaque();
#pragma unroll
  for (int j = 0; j < NP; ++j) if ((size_t)(tid * 16 + j * 8192) < bytes) *(v4u *)(smem + tid * 16 + j * 8192) = pre.xv[j];
  for (size_t o = (size_t)tid * 16 + (size_t)NP * 8192; o < bytes; o += 8192) *(v4u *)(smem + o) = *(const v4u *)((const char *)img + o);
}

// ------------------------------------------------------------------------------------------------ per-format records
// Raw = the registers a lane holds for its superblock (filled by buffer loads); term() turns them into T_sb for NCOLS activation columns.
__device__ __forceinline__ v4u ldb128(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2); }  // aux 2 = nt: read once per token
__device__ __forceinline__ unsigned ldb32(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 2); }
__device__ __forceinline__ unsigned ldb16(__amdgpu_buffer_rsrc_t r, unsigned off) { return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 2); }
__device__ __forceinline__ int byte_of(unsigned w, int i) { return (int)((w >> (8 * i)) & 0xffu); }
__device__ __forceinline__ int sbyte_of(unsigned w, int i) { return (int)(int8_t)((w >> (8 * i)) & 0xffu); }

template <int TYPE> struct Tile;

// Q4_K slot: q[8][16] = the GGUF qs bytes with bit 7 of every byte flipped (the high nibble then reads as the SIGNED nibble q - 8, times 16, through a plain
// mask), hs[16] = the 8 sub-block scales, then the 8 mins, as bytes; hd = d | dmin << 16 (f16 bits).  Sub-block 2c <- low nibbles of pieces 2c, 2c+1
// (activation runs 4c, 4c+1), sub-block 2c+1 <- high nibbles (runs 4c+2, 4c+3).
#ifndef MRS_DEC2_NS_Q4K
#define MRS_DEC2_NS_Q4K 3
#endif
#ifndef MRS_DEC2_NS_Q6K
#define MRS_DEC2_NS_Q6K 2
#endif
template <> struct Tile<T_Q4_K> {
  static constexpr int NS = MRS_DEC2_NS_Q4K;
  struct Raw { v4u q[8]; v4u hs; unsigned hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned rec, int a, int A, bool ok) {
    Raw r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.q[i] = ldb128(rs, ok ? rec + (unsigned)(i * A + a) * 16u : OOB);
    r.hs = ldb128(rs, ok ? rec + (unsigned)(8 * A + a) * 16u : OOB);
    r.hd = ldb32(rs, ok ? rec + (unsigned)(144 * A + 4 * a) : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, bool live, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    int dlo[NCOLS][4], dhi[NCOLS][4];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) { dlo[c][g] = 0; dhi[c][g] = 0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const v4u lo = w.q[i] & 0x0F0F0F0Fu, hi = w.q[i] & 0xF0F0F0F0u;
      const int g = i >> 1, ra = 4 * g + (i & 1), rb = ra + 2;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const char *qc = act.q + ((size_t)(col0 + c) * act.S + sb) * ACT_QS;
        dlo[c][g] = dot16(lo, *(const int4 *)(qc + ra * 16), dlo[c][g]);
        dhi[c][g] = dot16(hi, *(const int4 *)(qc + rb * 16), dhi[c][g]);
      }
    }
    const unsigned scw[2] = {w.hs.x, w.hs.y}, mw[2] = {w.hs.z, w.hs.w};
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const int *bsp = act.bs + ((size_t)(col0 + c) * act.S + sb) * ACT_BS;
      int isum16 = 0, msum = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 b = *(const int4 *)(bsp + 4 * g);  // runs 4g .. 4g+3
        const int bsa = b.x + b.y, bsb = b.z + b.w;   // sub-blocks 2g, 2g+1
        const int sca = byte_of(scw[g >> 1], 2 * (g & 1)), scb = byte_of(scw[g >> 1], 2 * (g & 1) + 1);
        const int ma = byte_of(mw[g >> 1], 2 * (g & 1)), mb = byte_of(mw[g >> 1], 2 * (g & 1) + 1);
        isum16 += __mul24(sca << 4, dlo[c][g]) + __mul24(scb, dhi[c][g] + (bsb << 7));  // 16 sum q a = sum 16 (q - 8) a + 128 sum a
        msum += __mul24(ma, bsa) + __mul24(mb, bsb);
      }
      const float yd = act.d[((size_t)(col0 + c) * act.S + sb) * act.dw];
      const float t = fmaf(d * yd, (float)isum16 * 0.0625f, -((dmin * yd) * (float)msum));
      T[c] = live ? t : 0.0f;
    }
  }
};

// Q5_K slot: q[8][16] = the GGUF qs bytes; xh[2][16]: dword i (0..7) = the fifth bits of piece i: bit 8 j' + k = bit of LOW-nibble weight 4 k + j' (k = dword of
// the piece, j' = byte), bit 8 j' + 4 + k = of the HIGH-nibble weight; hs, hd as Q4_K
template <> struct Tile<T_Q5_K> {
  static constexpr int NS = 2;
  struct Raw { v4u q[8]; v4u xh[2]; v4u hs; unsigned hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned rec, int a, int A, bool ok) {
    Raw r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.q[i] = ldb128(rs, ok ? rec + (unsigned)(i * A + a) * 16u : OOB);
#pragma unroll
    for (int i = 0; i < 2; ++i) r.xh[i] = ldb128(rs, ok ? rec + (unsigned)((8 + i) * A + a) * 16u : OOB);
    r.hs = ldb128(rs, ok ? rec + (unsigned)(10 * A + a) * 16u : OOB);
    r.hd = ldb32(rs, ok ? rec + (unsigned)(176 * A + 4 * a) : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, bool live, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff)), dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
    int dlo[NCOLS][4], dhi[NCOLS][4];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) { dlo[c][g] = 0; dhi[c][g] = 0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned xh = i < 4 ? (i == 0 ? w.xh[0].x : i == 1 ? w.xh[0].y : i == 2 ? w.xh[0].z : w.xh[0].w) : (i == 4 ? w.xh[1].x : i == 5 ? w.xh[1].y : i == 6 ? w.xh[1].z : w.xh[1].w);
      v4u lo = w.q[i] & 0x0F0F0F0Fu, hi = (w.q[i] >> 4) & 0x0F0F0F0Fu;
      lo.x |= (xh & 0x01010101u) << 4; lo.y |= ((xh >> 1) & 0x01010101u) << 4; lo.z |= ((xh >> 2) & 0x01010101u) << 4; lo.w |= ((xh >> 3) & 0x01010101u) << 4;
      hi.x |= ((xh >> 4) & 0x01010101u) << 4; hi.y |= ((xh >> 5) & 0x01010101u) << 4; hi.z |= ((xh >> 6) & 0x01010101u) << 4; hi.w |= ((xh >> 7) & 0x01010101u) << 4;
      const int g = i >> 1, ra = 4 * g + (i & 1), rb = ra + 2;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const char *qc = act.q + ((size_t)(col0 + c) * act.S + sb) * ACT_QS;
        dlo[c][g] = dot16(lo, *(const int4 *)(qc + ra * 16), dlo[c][g]);
        dhi[c][g] = dot16(hi, *(const int4 *)(qc + rb * 16), dhi[c][g]);
      }
    }
    const unsigned scw[2] = {w.hs.x, w.hs.y}, mw[2] = {w.hs.z, w.hs.w};
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const int *bsp = act.bs + ((size_t)(col0 + c) * act.S + sb) * ACT_BS;
      int isum = 0, msum = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 b = *(const int4 *)(bsp + 4 * g);
        const int bsa = b.x + b.y, bsb = b.z + b.w;
        const int sca = byte_of(scw[g >> 1], 2 * (g & 1)), scb = byte_of(scw[g >> 1], 2 * (g & 1) + 1);
        const int ma = byte_of(mw[g >> 1], 2 * (g & 1)), mb = byte_of(mw[g >> 1], 2 * (g & 1) + 1);
        isum += __mul24(sca, dlo[c][g]) + __mul24(scb, dhi[c][g]);
        msum += __mul24(ma, bsa) + __mul24(mb, bsb);
      }
      const float yd = act.d[((size_t)(col0 + c) * act.S + sb) * act.dw];
      const float t = fmaf(d * yd, (float)isum, -((dmin * yd) * (float)msum));
      T[c] = live ? t : 0.0f;
    }
  }
};

// Q6_K slot: ql[8][16]: byte b of piece i = low 4 bits of weight b of run 2i (low nibble) and of run 2i+1 (high nibble); qh[4][16]: byte b of piece g = the
// top 2 bits of weight b of runs 4g .. 4g+3 (bits 1:0, 3:2, 5:4, 7:6); sc[16] int8; d f16.  q - 32 is applied through the per-16 activation sums.
template <> struct Tile<T_Q6_K> {
  static constexpr int NS = MRS_DEC2_NS_Q6K;
  struct Raw { v4u ql[8]; v4u qh[4]; v4u sc; unsigned hd; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned rec, int a, int A, bool ok) {
    Raw r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.ql[i] = ldb128(rs, ok ? rec + (unsigned)(i * A + a) * 16u : OOB);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.qh[i] = ldb128(rs, ok ? rec + (unsigned)((8 + i) * A + a) * 16u : OOB);
    r.sc = ldb128(rs, ok ? rec + (unsigned)(12 * A + a) * 16u : OOB);
    r.hd = ldb16(rs, ok ? rec + (unsigned)(208 * A + 2 * a) : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, bool live, const Act &act, int col0, float (&T)[NCOLS]) {
    const float d = half_bits_to_float((uint16_t)w.hd);
    int isum[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) isum[c] = 0;
    const unsigned scw[4] = {w.sc.x, w.sc.y, w.sc.z, w.sc.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // runs 4g .. 4g+3
      const v4u h = w.qh[g];
      const v4u q0 = (w.ql[2 * g] & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
      const v4u q1 = ((w.ql[2 * g] >> 4) & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
      const v4u q2 = (w.ql[2 * g + 1] & 0x0F0F0F0Fu) | (h & 0x30303030u);
      const v4u q3 = ((w.ql[2 * g + 1] >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
      const int s0 = sbyte_of(scw[g], 0), s1 = sbyte_of(scw[g], 1), s2 = sbyte_of(scw[g], 2), s3 = sbyte_of(scw[g], 3);
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const char *qc = act.q + ((size_t)(col0 + c) * act.S + sb) * ACT_QS + g * 64;
        const int4 b = *(const int4 *)(act.bs + ((size_t)(col0 + c) * act.S + sb) * ACT_BS + 4 * g);
        // sum sc <q - 32, u> = sum sc (<q, u> - 32 sum u), all integer
        isum[c] += __mul24(s0, dot16(q0, *(const int4 *)(qc), 0) - 32 * b.x) + __mul24(s1, dot16(q1, *(const int4 *)(qc + 16), 0) - 32 * b.y);
        isum[c] += __mul24(s2, dot16(q2, *(const int4 *)(qc + 32), 0) - 32 * b.z) + __mul24(s3, dot16(q3, *(const int4 *)(qc + 48), 0) - 32 * b.w);
      }
    }
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const float yd = act.d[((size_t)(col0 + c) * act.S + sb) * act.dw];
      const float t = (d * yd) * (float)isum[c];
      T[c] = live ? t : 0.0f;
    }
  }
};

// Q8_0 "superblock" = 8 consecutive blocks of 32: q[16][16] = the int8 quants in element order, dh[16] = the 8 f16 scales.  Activations: Q8_0 blocks.
template <> struct Tile<T_Q8_0> {
  static constexpr int NS = 1;
  struct Raw { v4u q[16]; v4u dh; };
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, unsigned rec, int a, int A, bool ok) {
    Raw r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.q[i] = ldb128(rs, ok ? rec + (unsigned)(i * A + a) * 16u : OOB);
    r.dh = ldb128(rs, ok ? rec + (unsigned)(16 * A + a) * 16u : OOB);
    return r;
  }
  template <int NCOLS> static __device__ __forceinline__ void term(const Raw &w, int sb, bool live, const Act &act, int col0, float (&T)[NCOLS]) {
    const unsigned dw4[4] = {w.dh.x, w.dh.y, w.dh.z, w.dh.w};
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      const char *qc = act.q + ((size_t)(col0 + c) * act.S + sb) * ACT_QS;
      const float *dx = act.d + ((size_t)(col0 + c) * act.S + sb) * act.dw;
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int is = dot16(w.q[2 * b + 1], *(const int4 *)(qc + (2 * b + 1) * 16), dot16(w.q[2 * b], *(const int4 *)(qc + 2 * b * 16), 0));
        const float dwb = half_bits_to_float((uint16_t)(b & 1 ? dw4[b >> 1] >> 16 : dw4[b >> 1] & 0xffff));
        const float p = (float)is * dwb * dx[b];
        t = b == 0 ? p : t + p;
      }
      T[c] = live ? t : 0.0f;
    }
  }
};

// T for NCOLS activation columns, two at a time: the weight registers are unpacked once per pair, and the integer accumulators of a pair fit the register file
template <int TYPE, int NCOLS, int C0> struct TermCols {
  static __device__ __forceinline__ void run(const typename Tile<TYPE>::Raw &w, int sb, bool live, const Act &act, int col0, float (&T)[NCOLS]) {
    if constexpr (C0 < NCOLS) {
      if constexpr (NCOLS - C0 >= 2) {
        float t2[2];
        Tile<TYPE>::template term<2>(w, sb, live, act, col0 + C0, t2);
        T[C0] = t2[0]; T[C0 + 1] = t2[1];
      } else {
        float t1[1];
        Tile<TYPE>::template term<1>(w, sb, live, act, col0 + C0, t1);
        T[C0] = t1[0];
      }
      if constexpr (C0 + 2 < NCOLS) __builtin_amdgcn_sched_barrier(0);  // one pair at a time: interleaved pairs spill
      TermCols<TYPE, NCOLS, C0 + 2>::run(w, sb, live, act, col0, T);
    }
  }
};

// ------------------------------------------------------------------------------------------------ the streaming core
// A workgroup owns the UNITS [u0, u1) of a launch; unit u = rgpu consecutive record groups (a record group = R consecutive rows) of each of the launch's
// nseg tensors (gate and up rows of the same index travel together; rgpu = 2 keeps a RoPE pair in one wave when R = 1).  Waves take units one at a time from a
// counter in LDS (the first one statically), so a wave that starts late -- the prologue waves of the SPEC schedule -- simply ends up with fewer; a wave
// keeps up to NS records requested ahead of the one it is computing.  epi(seg, row0, nvalid, rgl, sums, aux) is called once per finished record group; inside a
// unit: segment 0 before segment 1, record groups ascending; the call is lane-parallel: every lane of row rr = lane / (64 / R) of the group holds that row's sum,
// lane rr * (64 / R) is the row's owner, nvalid rows exist.  aux(unit, seg, rgl) runs when a record is REQUESTED (operands of the epilogue -- residual values, RoPE
// factors -- travel with the weights instead of costing a dependent load after the row sum); its result comes back to epi for that record's rows.
struct Job {
  Mat mat[2];
  int nseg, rgpu;
  int nrows;            // rows per tensor (per expert slot) that take part
  int u0, u1;           // this workgroup's units
  int ring;             // records a wave keeps requested ahead (<= the format's NS register sets): measured on the MI355X, short launches want 1 (the time to the
                        // first computed record decides), long streams the full ring (profiles/round4_decode.md)
  const int32_t *sel;   // stacked experts [E * rows][K]: device array of expert ids per slot, or nullptr (dense)
  int sel_mode;         // 1: slot = unit / upe (all top-k experts of a token in one launch; upe = every unit when there is one slot);  2: slot = segment
  int upe;              // units per expert slot
  int ergs;             // record groups per expert in the stacked tensor
  unsigned long long *tl;  // experiments: 16 s_memrealtime stamps (100 MHz) per wave, or nullptr
};
#define MRS_TL2(jb, i) do { if ((jb).tl && (tid_opaque() & 63) == 0) (jb).tl[((size_t)blockIdx.x * NW + (tid_opaque() >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the lane of a record group's row rr that holds the row sum in the epilogue (and loads the row's epilogue operands)
__host__ __device__ inline int owner_off(const Geo &g) { return 3 * g.LPC + g.W - 1; }
struct RecMeta { int unit, seg, rgl, ts; };  // unit < 0: nothing was requested into the slot
struct NoAux {};

// SEGCOL (NCOLS must be 1): segment s multiplies by activation column s of a 2-column image (MoE down: two experts' rows against their own activations)
template <int TYPE, int NCOLS, bool SPEC, bool SEGCOL = false, class Stage, class AuxF, class Epi>
__device__ __forceinline__ void stream(const Job &jb, int K, int ncols_img, int mode, char *smem, int *ctr, Stage stage, AuxF auxf, Epi epi) {
  using TL = Tile<TYPE>;
  using AuxT = decltype(auxf(0, 0, 0));
  constexpr int NS = NCOLS == 1 ? TL::NS : (TL::NS > 2 ? 2 : TL::NS);
  constexpr bool spec = SPEC;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = geo_for(K);
  const unsigned recb = (unsigned)rec_bytes(TYPE, g);
  const int lpr = 4 * g.LPC;  // lanes per row of the record
  const int r = lane / lpr, p = (lane / g.LPC) & 3, j = lane & (g.LPC - 1);
  const bool lane_ok = j < g.W;
  const int a = (r * 4 + p) * g.W + j;
  const int rps = jb.rgpu * g.TPC, rpu = jb.nseg * rps;  // records per segment of a unit, per unit
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)jb.mat[0].base, (short)0, (int)jb.mat[0].bytes, 0x00020000);
  const bool two = jb.nseg > 1;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(two ? jb.mat[1].base : jb.mat[0].base), (short)0, (int)(two ? jb.mat[1].bytes : jb.mat[0].bytes), 0x00020000);
  typename TL::Raw ring[NS];
  RecMeta meta[NS];
  AuxT auxv[NS];
  // first unit: static (ALL: wave w takes u0 + w; SPEC: the streaming waves take u0 + w - PW, the prologue waves come to the counter after the barrier)
  const bool late = spec && wave < PW;
  const int nstatic = spec ? NW - PW : NW;
  int lunit = late ? -1 : jb.u0 + (spec ? wave - PW : wave), lpos = 0;
  bool started = !late;
  auto issue = [&](typename TL::Raw &slot, RecMeta &m, AuxT &ax) {
    if (started && lunit >= 0 && lpos == rpu) {  // next unit (wave-uniform): ALL -- every NW-th unit, no traffic; SPEC -- from the counter in LDS (late waves take fewer)
      if constexpr (SPEC) {
        int nu = 0;
        if (lane == 0) nu = atomicAdd(ctr, 1);
        lunit = __builtin_amdgcn_readfirstlane(nu);
      } else {
        lunit += NW;
      }
      lpos = 0;
    }
    const bool livel = started && lunit >= 0 && lunit < jb.u1;
    const int seg = lpos / rps, rem = lpos - seg * rps, rgl = rem / g.TPC, ts = rem - rgl * g.TPC;
    m = RecMeta{livel ? lunit : -1, seg, rgl, ts};
    if (livel) {  // a wave without a record requests nothing: an out-of-range load still costs its 16 cycles in the CU's one texture addresser
      int local = lunit * jb.rgpu + rgl, expert = 0;
      if (jb.sel) {
        if (jb.sel_mode == 2) expert = jb.sel[seg];
        else { const int slot = lunit / jb.upe; expert = jb.sel[slot]; local -= slot * jb.upe * jb.rgpu; }
      }
      const unsigned rec = ((unsigned)(expert * jb.ergs + local) * (unsigned)g.TPC + (unsigned)ts) * recb;
      slot = seg == 0 ? TL::load(rs0, rec, a, g.A, lane_ok) : TL::load(rs1, rec, a, g.A, lane_ok);
      if (ts == g.TPC - 1) ax = auxf(lunit, seg, rgl);  // the record whose slot the epilogue runs from
      ++lpos;
    } else {
      lunit = -1;
    }
  };
  const int nsr = jb.ring < 1 ? 1 : (jb.ring > NS ? NS : jb.ring);
#pragma unroll
  for (int i = 0; i < NS; ++i) meta[i] = RecMeta{-1, 0, 0, 0};
  auto fill = [&]() {
#pragma unroll
    for (int i = 0; i < NS; ++i) if (i < nsr) issue(ring[i], meta[i], auxv[i]);
  };
  // A CU has one in-order memory pipe: what the prologue needs from memory is requested (stage 0), by every wave that takes part, BEFORE any wave of the
  // workgroup requests weights -- the first barrier sits in between.  The branches below execute the same two barriers.
  if constexpr (SPEC) {
    if (late) {  // prologue waves: the row -> registers, prologue while the other waves request weights, then their own first unit from the counter
      stage(0);
      MRS_TL2(jb, 0);
      if (tid == 0) *ctr = jb.u0 + nstatic;
      __syncthreads();
      stage(1);
      MRS_TL2(jb, 2);
      __syncthreads();
      MRS_TL2(jb, 3);
      int nu = 0;
      if (lane == 0) nu = atomicAdd(ctr, 1);
      lunit = __builtin_amdgcn_readfirstlane(nu);
      started = true;
      fill();
    } else {
      MRS_TL2(jb, 0);
      __syncthreads();
      fill();
      MRS_TL2(jb, 1);
      __syncthreads();
      MRS_TL2(jb, 3);
    }
  } else {
    stage(0);
    MRS_TL2(jb, 0);
    if (tid == 0) *ctr = jb.u0 + nstatic;
    __syncthreads();
    fill();
    MRS_TL2(jb, 1);
    stage(1);  // squares, (barrier), quantize
    MRS_TL2(jb, 2);
    __syncthreads();
    MRS_TL2(jb, 3);
  }
  const Act act = act_view(smem, K, ncols_img, mode);
  float carry[NCOLS];
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) carry[c] = 0.0f;
  const int gbase = lane & ~(g.LPC - 1), rbase = lane & ~(lpr - 1);
  bool more = true;
  int nrec = 0;
  // T of the record in slot i for this lane's superblock
  auto terms = [&](int i, float (&T)[NCOLS]) {
    const int sbi = meta[i].ts * g.W + j, sb = p * g.Cs + sbi;  // index inside the chunk, superblock
    const bool live = lane_ok && sbi < g.Cs && sb < g.S;
    TermCols<TYPE, NCOLS, 0>::run(ring[i], live ? sb : 0, live, act, SEGCOL ? meta[i].seg : 0, T);
  };
  // chunk scan, row combination, epilogue of slot i; then the slot's next request
  auto finish = [&](int i, const float (&T)[NCOLS]) {
    const int cts = meta[i].ts, cseg = meta[i].seg;
    float s[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) s[c] = (cts > 0 && j == 0) ? carry[c] + T[c] : T[c];
    for (int it = 1; it < g.W; ++it) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) s[c] = dppf<0x121>(s[c]) + T[c];  // row_ror:1: lane i reads lane i - 1; after step `it` lane `it` of a group holds the sum of its first it + 1 terms
    }
    if (cts + 1 < g.TPC) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) carry[c] = __shfl(s[c], gbase + g.W - 1, 64);
    } else {
      // the four chunk sums sit in lanes k * LPC + W - 1 of the row; row = ((c0 + c1) + c2) + c3 must end up (at least) in the row's OWNER lane 3 * LPC + W - 1.
      // No LDS round trips (ds_bpermute) on this path: it is a serial latency chain behind every record.
      float tot[NCOLS];
      if (g.LPC == 4) {  // rows of 16 lanes = DPP rows: three row_ror:4 steps walk the sum from lane W - 1 to lane 12 + W - 1
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          float u = dppf<0x124>(s[c]) + s[c];
          u = dppf<0x124>(u) + s[c];
          tot[c] = dppf<0x124>(u) + s[c];
        }
      } else if (g.LPC >= 8) {  // one or two rows per record: the chunk sums through readlane (scalar operands)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          const float a0 = rlf(s[c], g.W - 1), a1 = rlf(s[c], g.LPC + g.W - 1), a2 = rlf(s[c], 2 * g.LPC + g.W - 1), a3 = rlf(s[c], 3 * g.LPC + g.W - 1);
          float t0 = ((a0 + a1) + a2) + a3;
          if (g.LPC == 8) {
            const float b0 = rlf(s[c], 32 + g.W - 1), b1 = rlf(s[c], 40 + g.W - 1), b2 = rlf(s[c], 48 + g.W - 1), b3 = rlf(s[c], 56 + g.W - 1);
            const float t1 = ((b0 + b1) + b2) + b3;
            t0 = lane < 32 ? t0 : t1;
          }
          tot[c] = t0;
        }
      } else {  // 8 or 16 rows per record (K <= 2048)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          const float c0 = __shfl(s[c], rbase + g.W - 1, 64), c1 = __shfl(s[c], rbase + g.LPC + g.W - 1, 64);
          const float c2 = __shfl(s[c], rbase + 2 * g.LPC + g.W - 1, 64), c3 = __shfl(s[c], rbase + 3 * g.LPC + g.W - 1, 64);
          tot[c] = ((c0 + c1) + c2) + c3;
        }
      }
      // one epilogue call per record group, lane-parallel over its R rows: the owner lane of row rr is rr * lpr + 3 * LPC + W - 1 (owner_off())
      const int urow = meta[i].unit * jb.rgpu * g.R + meta[i].rgl * g.R;          // first row of the group in the launch's numbering (slot * rows-per-slot + local row)
      const int lrow = urow - (meta[i].unit / jb.upe) * (jb.upe * jb.rgpu * g.R);  // local row inside the expert slot
      epi(cseg, urow, min(g.R, jb.nrows - lrow), meta[i].rgl, tot, auxv[i]);
    }
    issue(ring[i], meta[i], auxv[i]);
    if (nrec < 10) MRS_TL2(jb, 4 + nrec);
    ++nrec;
  };
  while (more) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i >= nsr) continue;
      if (!(more && meta[i].unit >= 0)) { more = false; continue; }  // wave-uniform
      // (two records at a time -- interleaving their integer dots -- was built and measured: the second set of accumulators spills, every launch 15-25 % slower)
      float T0[NCOLS];
      terms(i, T0);
      finish(i, T0);
    }
  }
  MRS_TL2(jb, 14);
}

}  // namespace dec2
}  // namespace mrs
