// dec_gemv.cuh -- the GEMV phase kernel of the decode engine (argument block, epilogues, kernel template); instantiated per activation-column count in
// ext_dec_gemv.hip (one translation unit per count: the instantiations are large), launched from ext_dec.hip.
#pragma once
#include "dec_core2.cuh"
#include <algorithm>

namespace mrs {
namespace dec {
using namespace mrs::dec2;

enum : int { EPI_STORE = 0, EPI_RESID = 1, EPI_GLU = 2, EPI_QKV = 3, EPI_RESID2 = 4 };

struct GemvArgs {
  Mat m[3];
  int nrows[3];  // logical rows of the phase per tensor (per expert for stacked experts)
  int K;
  const float *x; int ldx; const float *norm_w; float eps;
  float *out; int out_stride; float resid_scale;
  int activation;
  float *q_out; void *k_cache, *v_cache; const int64_t *slot_mapping; const int32_t *positions; const float *cos_t, *sin_t;
  int head_dim, rot_pairs, num_kv_heads, block_size, cache_x, kv_f16;
  int neox;       // EPI_QKV: rows of q / k are stored in PAIR order (original rows i, i + head_dim / 2 of a head adjacent): rotate-half RoPE; results go back to i, i + head_dim / 2
  int wg0[4];     // EPI_QKV: first workgroup of q, k, v and the total (a workgroup streams ONE tensor: the three may have different formats)
  const int32_t *expert_sel;  // stacked experts [E * nrows][K]: rows of expert e start at e * nrows (nullptr = dense)
  const float *acc_scale;     // RESID: out = out * resid_scale + (*acc_scale) * W.x  (routing weight)
  int slots, slot_out_stride; // GLU with several experts of ONE token in a launch (MoE top-k): row r of the launch -> slot r / nrows[0], output out + slot * slot_out_stride
  const void *x_img;          // activations already quantized by the producer (dec_attn2_kernel): the LDS image of NCOLS columns, byte for byte
  int units[3], rgpu;         // units of the launch (per tensor for QKV) and record groups per unit
  unsigned long long *tl;
};

#define MRS_DEC_TYPE_SWITCH(t, ...)                              \
  switch (t) {                                                   \
  case T_Q4_K: { constexpr int TT = T_Q4_K; __VA_ARGS__ } break; \
  case T_Q5_K: { constexpr int TT = T_Q5_K; __VA_ARGS__ } break; \
  case T_Q6_K: { constexpr int TT = T_Q6_K; __VA_ARGS__ } break; \
  case T_Q8_0: { constexpr int TT = T_Q8_0; __VA_ARGS__ } break; \
  default: break;                                                \
  }

template <int N> struct AuxV { float a[N], b[N]; };

template <int NCOLS, int EPI, bool SPEC>
__device__ __forceinline__ void gemv_phase(const GemvArgs &a, char *smem, float *red, int *ctr) {
  const int tid0 = tid_opaque();
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane = tid0 & 63;
  const int K = a.K;
  const Geo g = geo_for(K);
  // which tensor this workgroup streams (QKV: one of three), its units and its share of them
  int mi = 0;
  if constexpr (EPI == EPI_QKV) mi = (int)blockIdx.x >= a.wg0[2] ? 2 : ((int)blockIdx.x >= a.wg0[1] ? 1 : 0);
  const int wgb = EPI == EPI_QKV ? (mi == 0 ? a.wg0[0] : (mi == 1 ? a.wg0[1] : a.wg0[2])) : 0;
  const int wge = EPI == EPI_QKV ? (mi == 0 ? a.wg0[1] : (mi == 1 ? a.wg0[2] : a.wg0[3])) : (int)gridDim.x;
  const int units = EPI == EPI_QKV ? (mi == 0 ? a.units[0] : (mi == 1 ? a.units[1] : a.units[2])) : a.units[0];
  const int nwg = wge - wgb, wi = (int)blockIdx.x - wgb;
  Job jb{};
  jb.nseg = (EPI == EPI_GLU || EPI == EPI_RESID2) ? 2 : 1;
  jb.rgpu = a.rgpu;
  jb.mat[0] = mi == 0 ? a.m[0] : (mi == 1 ? a.m[1] : a.m[2]);
  jb.mat[1] = EPI == EPI_GLU ? a.m[1] : a.m[0];
  jb.nrows = mi == 0 ? a.nrows[0] : (mi == 1 ? a.nrows[1] : a.nrows[2]);
  jb.u0 = (int)((long long)wi * units / nwg); jb.u1 = (int)((long long)(wi + 1) * units / nwg);
  jb.sel = a.expert_sel; jb.sel_mode = EPI == EPI_RESID2 ? 2 : 1;
  jb.upe = EPI == EPI_RESID2 ? units : units / (a.slots > 1 ? a.slots : 1);
  if (jb.upe < 1) jb.upe = 1;
  jb.ergs = (a.nrows[0] + g.R - 1) / g.R;
  jb.tl = a.tl;
  const int mode = act_mode_for(jb.mat[0].type);
  constexpr int NCI = EPI == EPI_RESID2 ? 2 : NCOLS;  // columns of the activation image

  // the activation prologue (dec_core2.cuh): a pre-quantized image is copied, an f32 vector is normalised / quantized on the ALL or the SPEC schedule
  ActRegs<2> pre;
  SpecRegs spre;
  const size_t img_bytes = act_bytes(K, NCI);
  auto stage = [&](int st) {
    if (a.x_img) {
      if (st == 0) pre = img_issue_all<2>(a.x_img, img_bytes); else img_finish_all<2>(smem, pre, a.x_img, img_bytes);
    } else if constexpr (SPEC) {
      if (st == 0) spre = act_issue_spec(a.x, a.norm_w, K, wave); else act_finish_spec<NCI>(smem, spre, a.x, a.ldx, a.norm_w, a.eps, K, mode, wave);
    } else {
      if (st == 0) pre = act_issue_all<2>(a.x, a.norm_w, K); else act_finish_all<NCI, 2>(smem, red, pre, a.x, a.ldx, a.norm_w, a.eps, K, mode);
    }
  };

  if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
    const float ascale = a.acc_scale ? *a.acc_scale : 1.0f;
    // residual values travel with the record: lane rr <-> row rr of the record group
    auto auxf = [&](int unit, int, int rgl) {
      AuxV<NCOLS> v;
      const int row = (unit * jb.rgpu + rgl) * g.R + lane;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) { v.a[c] = 0.f; v.b[c] = 0.f; if (EPI == EPI_RESID && lane < g.R && row < jb.nrows) v.a[c] = a.out[(size_t)c * a.out_stride + row]; }
      return v;
    };
    auto epi = [&](int, int row, int rr, const float(&sum)[NCOLS], const AuxV<NCOLS> &ax) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        float *o = a.out + (size_t)c * a.out_stride + row;
        if constexpr (EPI == EPI_RESID) {
          const float old = rlf(ax.a[c], rr);
          if (lane == 0) *o = old * a.resid_scale + sum[c] * ascale;
        } else {
          if (lane == 0) *o = sum[c];
        }
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, SPEC>(jb, K, NCI, mode, smem, ctr, stage, auxf, epi); })
  } else if constexpr (EPI == EPI_RESID2) {
    // MoE down of the two experts of one token in one launch (the image has two columns = the two experts' activation vectors): a unit streams its rows of
    // expert sel[0] against column 0, then the same rows of expert sel[1] against column 1, and writes (out * resid_scale + w0 s0) * 1 + w1 s1 -- the two
    // roundings of two consecutive EPI_RESID launches, bit for bit
    static_assert(EPI != EPI_RESID2 || NCOLS == 1, "one accumulator column; the image has two");
    const float w0 = a.acc_scale[0], w1 = a.acc_scale[1];
    float s0save = 0.0f;  // lane rr keeps the first expert's sum of row rr of the record group
    auto auxf = [&](int unit, int seg, int rgl) {
      AuxV<1> v; v.a[0] = 0.f; v.b[0] = 0.f;
      const int row = (unit * jb.rgpu + rgl) * g.R + lane;
      if (seg == 1 && lane < g.R && row < jb.nrows) v.a[0] = a.out[row];
      return v;
    };
    auto epi = [&](int seg, int row, int rr, const float(&sum)[1], const AuxV<1> &ax) {
      if (seg == 0) {
        s0save = lane == rr ? sum[0] : s0save;
      } else {
        const float s0 = rlf(s0save, rr), old = rlf(ax.a[0], rr);
        const float h1 = old * a.resid_scale + s0 * w0;
        if (lane == 0) a.out[row] = h1 * 1.0f + sum[0] * w1;
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, 1, SPEC, true>(jb, K, NCI, mode, smem, ctr, stage, auxf, epi); })
  } else if constexpr (EPI == EPI_GLU) {
    float gsave[NCOLS];  // lane rr keeps gate row rr of the record group until the matching up row arrives
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) gsave[c] = 0.0f;
    auto auxf = [](int, int, int) { return NoAux{}; };
    auto epi = [&](int seg, int row, int rr, const float(&sum)[NCOLS], const NoAux &) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        if (seg == 0) {
          gsave[c] = lane == rr ? sum[c] : gsave[c];
        } else {
          const float gt = rlf(gsave[c], rr);
          const int slot = a.slots > 1 ? row / a.nrows[0] : 0;
          if (lane == 0) a.out[(size_t)slot * a.slot_out_stride + (size_t)c * a.out_stride + (row - slot * a.nrows[0])] = (a.activation == 0 ? silu_engine(gt) : glu_act(gt, a.activation)) * sum[c];
        }
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, SPEC>(jb, K, NCI, mode, smem, ctr, stage, auxf, epi); })
  } else {  // EPI_QKV: rows 2i, 2i + 1 of a tensor are a RoPE pair; a record group holds whole pairs (R >= 2) or a unit holds two record groups (R = 1)
    // RoPE factors and the KV slot travel with the record: lane pp <-> pair pp of the record group
    auto auxf = [&](int unit, int, int rgl) {
      AuxV<NCOLS> v;
      const int row = (unit * jb.rgpu + rgl) * g.R + 2 * lane;
      const int pair_i = (row % a.head_dim) >> 1;
      const bool rot = mi < 2 && pair_i < a.rot_pairs;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        v.a[c] = 1.0f; v.b[c] = 0.0f;  // identity rotation for v and unrotated dims (x*1 - y*0 = x exactly)
        if (rot && 2 * lane < (g.R > 1 ? g.R : 2) && row < jb.nrows) {
          const size_t ti = (size_t)a.positions[c] * a.rot_pairs + pair_i;
          v.a[c] = a.cos_t[ti]; v.b[c] = a.sin_t[ti];
        }
      }
      return v;
    };
    float prev[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) prev[c] = 0.0f;
    auto epi = [&](int, int row, int rr, const float(&sum)[NCOLS], const AuxV<NCOLS> &ax) {  // row = local row of tensor mi
      if ((row & 1) == 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) prev[c] = sum[c];
        return;
      }
      const int lr = row - 1;          // even row of the pair
      const int pi = g.R > 1 ? (rr >> 1) : 0;  // pair index inside the record group (R = 1: the pair spans two record groups, factors loaded by lane 0 of each)
      const int head = lr / a.head_dim, dd = lr % a.head_dim;
      // where the two results live: adjacent dims (interleaved RoPE), or dims i and i + head_dim / 2 when the rows were stored in pair order (v: never)
      const bool nx = a.neox && mi < 2;
      const int d0 = nx ? dd >> 1 : dd, d1 = nx ? d0 + (a.head_dim >> 1) : dd + 1;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const float cs = rlf(ax.a[c], pi), sn = rlf(ax.b[c], pi);
        float x, y;
        rope_pair<float>(prev[c], sum[c], cs, sn, x, y);
        if (lane == 0) {
          if (mi == 0) {
            a.q_out[(size_t)c * a.nrows[0] + head * a.head_dim + d0] = x;
            a.q_out[(size_t)c * a.nrows[0] + head * a.head_dim + d1] = y;
          } else {
            const int64_t slot = a.slot_mapping[c];
            if (slot >= 0) {
              const int64_t blk = slot / a.block_size, off = slot % a.block_size;
              uint16_t *kc = (uint16_t *)a.k_cache, *vc = (uint16_t *)a.v_cache;
              const uint16_t xb = a.kv_f16 ? float_to_half_bits(x) : float_to_bf16_bits(x), yb = a.kv_f16 ? float_to_half_bits(y) : float_to_bf16_bits(y);
              if (mi == 1) {
                const int X = a.cache_x;
                const int64_t hb = (blk * a.num_kv_heads + head) * (a.head_dim / X);
                kc[(hb + d0 / X) * a.block_size * X + off * X + d0 % X] = xb;
                kc[(hb + d1 / X) * a.block_size * X + off * X + d1 % X] = yb;  // interleaved: d1 = d0 + 1, same 16-byte group
              } else {
                const int64_t o = ((blk * a.num_kv_heads + head) * a.head_dim + dd) * a.block_size + off;
                vc[o] = xb;
                vc[o + a.block_size] = yb;
              }
            }
          }
        }
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, SPEC>(jb, K, NCI, mode, smem, ctr, stage, auxf, epi); })
  }
}

template <int NCOLS, int EPI, bool SPEC>
__global__ void __launch_bounds__(NT) dec_gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8 * 8];  // RMSNorm partials: [column][wave]
  __shared__ int ctr;           // the workgroup's unit counter
  gemv_phase<NCOLS, EPI, SPEC>(a, smem, red, &ctr);
}


// launch one GEMV phase with NCOLS activation columns (ext_dec_gemv.hip, one definition per NCOLS)
template <int NCOLS> int gemv_launch(int epi, bool spec, int grid, size_t lds, const GemvArgs &a, hipStream_t s);

}  // namespace dec
}  // namespace mrs
