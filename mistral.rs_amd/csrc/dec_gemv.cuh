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
  int hd_shift, bs_shift, x_shift;  // log2 of head_dim / block_size / cache_x (the launcher refuses other values): the epilogue's index arithmetic is shifts and masks --
                                    // with run-time divisors it was ~10 integer divisions per column and row pair (~40 VALU each), 10 us of the batch-8 qkv launch
  int neox;       // EPI_QKV: rows of q / k are stored in PAIR order (original rows i, i + head_dim / 2 of a head adjacent): rotate-half RoPE; results go back to i, i + head_dim / 2
  int wg0[4];     // EPI_QKV: first workgroup of q, k, v and the total (a workgroup streams ONE tensor: the three may have different formats)
  const int32_t *expert_sel;  // stacked experts [E * nrows][K]: rows of expert e start at e * nrows (nullptr = dense)
  const float *acc_scale;     // RESID: out = out * resid_scale + (*acc_scale) * W.x  (routing weight)
  int slots, slot_out_stride; // GLU with several experts of ONE token in a launch (MoE top-k): row r of the launch -> slot r / nrows[0], output out + slot * slot_out_stride
  const void *x_img;          // activations already quantized by the producer (dec_attn2_kernel): the LDS image of NCOLS columns, byte for byte
  int units[3], rgpu;         // units of the launch (per tensor for QKV) and record groups per unit
  int ubase[3], urem[3], upe; // a tensor's units over its workgroups: workgroup wi takes ubase + (wi < urem) units from wi * ubase + min(wi, urem) (computed by the launcher:
                              // the kernel used to open with two 64-bit and one 32-bit integer division -- ~300 scalar instructions ahead of the first request of every
                              // launch); upe = units per expert slot
  unsigned long long *amax;   // EPI_STORE, one column (round 6): the launch's arg-max folded into the epilogue -- atomicMax of the packed (value, index) key of the
                              // workgroup's largest output (ext_decode.hip pack_max: largest value, lowest index; what argmax_partial_kernel computes from the stored
                              // logits in a launch of its own, sample_cuda_top1_row's role, mistralrs-core/src/ops.rs:2206); nullptr = off
  unsigned long long *tl;
};
// order-preserving key of (value, index): larger value wins, then the smaller index (== ext_decode.hip pack_max)
__device__ __forceinline__ unsigned long long pack_max_key(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - idx);
}

// TMASK: which weight formats a kernel instantiation contains (bit 0 Q4_K, 1 Q5_K, 2 Q6_K, 3 Q8_0).  One kernel with all four carries the register
// footprint of the fattest format; the QKV phase (three tensors, usually two formats, the heaviest epilogue) is instantiated per format set.
enum : int { TM_Q4K = 1, TM_Q5K = 2, TM_Q6K = 4, TM_Q80 = 8, TM_ALL = 15 };
__host__ __device__ constexpr int tmask_of(int type) { return type == T_Q4_K ? TM_Q4K : type == T_Q5_K ? TM_Q5K : type == T_Q6_K ? TM_Q6K : type == T_Q8_0 ? TM_Q80 : 0; }
#define MRS_DEC_TYPE_SWITCH(t, ...)                                                              \
  switch (t) {                                                                                   \
  case T_Q4_K: if constexpr ((TMASK & TM_Q4K) != 0) { constexpr int TT = T_Q4_K; __VA_ARGS__ } break; \
  case T_Q5_K: if constexpr ((TMASK & TM_Q5K) != 0) { constexpr int TT = T_Q5_K; __VA_ARGS__ } break; \
  case T_Q6_K: if constexpr ((TMASK & TM_Q6K) != 0) { constexpr int TT = T_Q6_K; __VA_ARGS__ } break; \
  case T_Q8_0: if constexpr ((TMASK & TM_Q80) != 0) { constexpr int TT = T_Q8_0; __VA_ARGS__ } break; \
  default: break;                                                                                \
  }

template <int N> struct AuxV { float a[N], b[N]; };

template <int NCOLS, int EPI, int TMASK, bool RING2>
__device__ __forceinline__ void gemv_phase(const GemvArgs &a, char *smem, float *red, int *ctr, unsigned long long *best) {
  const int tid0 = tid_opaque();
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane = tid0 & 63;
  const int K = a.K;
  const Geo g = geo_for(K);
  // which tensor this workgroup streams (QKV: one of three), its units and its share of them.  Every field is read into a local FIRST and the locals are
  // selected: a select between kernel-argument ADDRESSES makes hipcc copy the whole argument block to scratch memory and index it there.
  const int w1 = a.wg0[1], w2 = a.wg0[2], w3 = a.wg0[3];
  const int un0 = a.units[0], un1 = a.units[1], un2 = a.units[2];
  const int ub0 = a.ubase[0], ub1 = a.ubase[1], ub2 = a.ubase[2], ur0 = a.urem[0], ur1 = a.urem[1], ur2 = a.urem[2];
  const int nr0 = a.nrows[0], nr1 = a.nrows[1], nr2 = a.nrows[2];
  const uint8_t *b0 = a.m[0].base, *b1 = a.m[1].base, *b2 = a.m[2].base;
  const unsigned by0 = a.m[0].bytes, by1 = a.m[1].bytes, by2 = a.m[2].bytes;
  const int ty0 = a.m[0].type, ty1 = a.m[1].type, ty2 = a.m[2].type;
  const int bx = (int)blockIdx.x;
  int mi = 0;
  if constexpr (EPI == EPI_QKV) mi = bx >= w2 ? 2 : (bx >= w1 ? 1 : 0);
  const int wgb = EPI == EPI_QKV ? (mi == 0 ? 0 : (mi == 1 ? w1 : w2)) : 0;
  const int wge = EPI == EPI_QKV ? (mi == 0 ? w1 : (mi == 1 ? w2 : w3)) : (int)gridDim.x;
  const int units = EPI == EPI_QKV ? (mi == 0 ? un0 : (mi == 1 ? un1 : un2)) : un0;
  const int wi = bx - wgb;
  const int ub = EPI == EPI_QKV ? (mi == 0 ? ub0 : (mi == 1 ? ub1 : ub2)) : ub0, ur = EPI == EPI_QKV ? (mi == 0 ? ur0 : (mi == 1 ? ur1 : ur2)) : ur0;
  (void)wge; (void)units;
  Job jb{};
  jb.nseg = (EPI == EPI_GLU || EPI == EPI_RESID2) ? 2 : 1;
  jb.rgpu = a.rgpu;
  jb.mat[0].base = mi == 0 ? b0 : (mi == 1 ? b1 : b2);
  jb.mat[0].bytes = mi == 0 ? by0 : (mi == 1 ? by1 : by2);
  jb.mat[0].type = mi == 0 ? ty0 : (mi == 1 ? ty1 : ty2);
  jb.mat[0].n = 0; jb.mat[0].k = K;
  if constexpr (EPI == EPI_GLU) { jb.mat[1].base = b1; jb.mat[1].bytes = by1; jb.mat[1].type = ty1; jb.mat[1].n = 0; jb.mat[1].k = K; } else jb.mat[1] = jb.mat[0];
  jb.nrows = mi == 0 ? nr0 : (mi == 1 ? nr1 : nr2);
  jb.u0 = wi * ub + min(wi, ur); jb.u1 = jb.u0 + ub + (wi < ur ? 1 : 0);
  jb.sel = a.expert_sel; jb.sel_mode = EPI == EPI_RESID2 ? 2 : 1;
  jb.upe = a.upe;
  jb.ergs = (a.nrows[0] + g.R - 1) / g.R;
  jb.tl = a.tl;
  const int mode = act_mode_for(jb.mat[0].type);
  constexpr int NCI = EPI == EPI_RESID2 ? 2 : NCOLS;  // columns of the activation image

  // the activation prologue (dec_core2.cuh), every wave is a participant (one virtual wave each): a pre-quantized image is copied, an f32 vector is
  // normalised / quantized.  ONE producer of the register set for both cases (act_issue_all): a struct assigned from two different calls under a run-time branch is
  // demoted to scratch memory by hipcc.
  ActRegs<1> pre;
  const size_t img_bytes = act_bytes(K, NCI);
  const bool from_img = a.x_img != nullptr;
  const void *xsrc = from_img ? a.x_img : (const void *)a.x;
  const unsigned xbytes = from_img ? (unsigned)img_bytes : (unsigned)K * 4u;
  const float *nw_eff = from_img ? nullptr : a.norm_w;
  auto stage = [&](int st, int q) {
    if (st == 0) pre = act_issue_all<1>(xsrc, xbytes, nw_eff, K, q);
    else if (st == 1) act_sumsq_all<NCI, 1>(red, pre, a.x, a.ldx, nw_eff, K, q);  // (an image has no norm weight: nothing to do)
    else if (from_img) img_finish_all<1>(smem, pre, a.x_img, img_bytes, q);
    else act_quantize_all<NCI, 1>(smem, red, pre, a.x, a.ldx, a.norm_w, a.eps, K, mode, q);
  };
  const bool sbar = nw_eff != nullptr;
  constexpr bool AUX_RING = NCOLS == 1;  // epilogue operands travel with the tiles (latency-bound small batches) or are loaded by the epilogue
  const int lpr = 4 * g.LPC;             // lanes per row of a record group: row rr of the group = lanes [rr * lpr, (rr + 1) * lpr), owner lane = rr * lpr + owner_off(g)
  const int rr = lane / lpr;
  const bool own = (lane & (lpr - 1)) == owner_off(g);
  if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
    const float ascale = a.acc_scale ? *a.acc_scale : 1.0f;
    // residual values travel with the tiles of their record group (requested behind the weights; batch 1 and 2: AUX_RING), or are loaded by the epilogue (wider
    // batches: the ring of operands would cost NS x 2 x NCOLS registers).  Unconditional loads from clamped addresses: a lane-conditional load is a branch around a
    // VMEM instruction, and every such branch makes the tile waits of dec_core2.cuh stream() one load stricter.
    auto load_aux = [&](int row) {
      AuxV<NCOLS> v;
      const int rc = min(row, jb.nrows - 1);
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) { v.a[c] = 0.f; v.b[c] = 0.f; if constexpr (EPI == EPI_RESID) v.a[c] = a.out[(size_t)c * a.out_stride + rc]; }
      return v;
    };
    auto auxf = [&](int row0) {
      if constexpr (AUX_RING) return load_aux(row0 + rr); else return NoAux{};
    };
    auto epi = [&](int, int row0, int nvalid, int, const float(&sum)[NCOLS], const auto &axr) {
      AuxV<NCOLS> ax;
      if constexpr (AUX_RING) ax = axr; else ax = load_aux(row0 + rr);
      if (own && rr < nvalid) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          float *o = a.out + (size_t)c * a.out_stride + row0 + rr;
          if constexpr (EPI == EPI_RESID) *o = ax.a[c] * a.resid_scale + sum[c] * ascale;
          else {
            *o = sum[c];
            if constexpr (NCOLS == 1) { if (a.amax) { const unsigned long long key = pack_max_key(sum[c], row0 + rr); *best = key > *best ? key : *best; } }
          }
        }
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, false, RING2>(jb, K, NCI, mode, smem, ctr, sbar, stage, auxf, epi); })
  } else if constexpr (EPI == EPI_RESID2) {
    // MoE down of the two experts of one token in one launch (the image has two columns = the two experts' activation vectors): a unit streams its rows of
    // expert sel[0] against column 0, then the same rows of expert sel[1] against column 1, and writes (out * resid_scale + w0 s0) * 1 + w1 s1 -- the two
    // roundings of two consecutive EPI_RESID launches, bit for bit
    static_assert(EPI != EPI_RESID2 || NCOLS == 1, "one accumulator column; the image has two");
    const float w0 = a.acc_scale[0], w1 = a.acc_scale[1];
    float s0save = 0.0f;  // the first expert's sums of the record group
    auto auxf = [&](int row0) {
      AuxV<1> v; v.a[0] = 0.f; v.b[0] = 0.f;
      const int row = row0 + rr;
      v.a[0] = a.out[min(row, jb.nrows - 1)];  // unconditional (dec_core2.cuh stream(): every request of a tile is straight-line code); used by segment 1's epilogue
      return v;
    };
    auto epi = [&](int seg, int row0, int nvalid, int, const float(&sum)[1], const AuxV<1> &ax) {
      if (seg == 0) {
        s0save = sum[0];
      } else if (own && rr < nvalid) {
        const float h1 = ax.a[0] * a.resid_scale + s0save * w0;
        a.out[row0 + rr] = h1 * 1.0f + sum[0] * w1;
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, 1, true, RING2>(jb, K, NCI, mode, smem, ctr, sbar, stage, auxf, epi); })
  } else if constexpr (EPI == EPI_GLU) {
    float gsave[NCOLS];  // the gate sums of the record group until the matching up rows arrive (same unit, same lanes)
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) gsave[c] = 0.0f;
    auto auxf = [](int) { return NoAux{}; };
    auto epi = [&](int seg, int row0, int nvalid, int, const float(&sum)[NCOLS], const NoAux &) {
      if (seg == 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) gsave[c] = sum[c];
      } else if (own && rr < nvalid) {
        const int row = row0 + rr;
        const int slot = a.slots > 1 ? row / a.nrows[0] : 0;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c)
          a.out[(size_t)slot * a.slot_out_stride + (size_t)c * a.out_stride + (row - slot * a.nrows[0])] = (a.activation == 0 ? silu_engine(gsave[c]) : glu_act(gsave[c], a.activation)) * sum[c];
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, false, RING2>(jb, K, NCI, mode, smem, ctr, sbar, stage, auxf, epi); })
  } else {  // EPI_QKV: rows 2i, 2i + 1 of a tensor are a RoPE pair; a record group holds whole pairs (R >= 2) or a unit holds two record groups (R = 1)
    // positions and KV slots: a handful of scalars, loaded before anything else; the RoPE factors travel with the record (owner lanes of the pair's two rows)
    int posv[NCOLS], slotv[NCOLS];  // slots are block * block_size + offset of a cache that fits 32-bit indexing per layer (checked by the launcher)
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) { posv[c] = a.positions[c]; slotv[c] = mi == 0 ? 0 : (int)a.slot_mapping[c]; }
    auto load_aux = [&](int row) {  // RoPE factors of the row's pair for every column; identity for v and unrotated dims (x*1 - y*0 = x exactly).  Unconditional loads.
      AuxV<NCOLS> v;
      const int pair_i = (row & (a.head_dim - 1)) >> 1;
      const bool rot = mi < 2 && pair_i < a.rot_pairs;
      const int pi = min(pair_i, a.rot_pairs - 1);
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const size_t ti = (size_t)posv[c] * a.rot_pairs + pi;
        const float cs = a.cos_t[ti], sn = a.sin_t[ti];
        v.a[c] = rot ? cs : 1.0f; v.b[c] = rot ? sn : 0.0f;
      }
      return v;
    };
    auto auxf = [&](int row0) {
      if constexpr (AUX_RING) return load_aux(row0 + rr); else return NoAux{};
    };
    float prev[NCOLS];  // R = 1: the even row's sum waits for the odd row's record group
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) prev[c] = 0.0f;
    auto epi = [&](int, int row0, int nvalid, int rgl, const float(&sum)[NCOLS], const auto &axr) {  // rows are local rows of tensor mi
      AuxV<NCOLS> ax;
      if constexpr (AUX_RING) ax = axr; else ax = load_aux(row0 + rr);
      const bool single = g.R == 1;
      if (single && (rgl & 1) == 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) prev[c] = sum[c];
        return;
      }
      const int row = row0 + rr;                   // this lane's row; its pair partner sits lpr lanes away (R = 1: in prev)
      const bool odd = single ? true : (row & 1) != 0;
      const int lr = single ? row - 1 : (row & ~1);  // even row of the pair
      const int head = lr >> a.hd_shift, dd = lr & (a.head_dim - 1);
      // where the two results live: adjacent dims (interleaved RoPE), or dims i and i + head_dim / 2 when the rows were stored in pair order (v: never)
      const bool nx = a.neox && mi < 2;
      const int d0 = nx ? dd >> 1 : dd, d1 = nx ? d0 + (a.head_dim >> 1) : dd + 1;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const float other = single ? prev[c] : __shfl(sum[c], lane ^ lpr, 64);
        const float xs = odd ? other : sum[c], ys = odd ? sum[c] : other;
        float x, y;
        rope_pair<float>(xs, ys, ax.a[c], ax.b[c], x, y);
        const bool wr0 = single ? own : (own && rr < nvalid && !odd), wr1 = single ? own : (own && rr < nvalid && odd);  // R = 1: one lane writes both
        if (wr0 || wr1) {
          if (mi == 0) {
            if (wr0) a.q_out[(size_t)c * a.nrows[0] + head * a.head_dim + d0] = x;
            if (wr1) a.q_out[(size_t)c * a.nrows[0] + head * a.head_dim + d1] = y;
          } else {
            const int slot = slotv[c];
            if (slot >= 0) {
              const unsigned blk = (unsigned)slot >> a.bs_shift, off = (unsigned)slot & (unsigned)(a.block_size - 1);
              uint16_t *kc = (uint16_t *)a.k_cache, *vc = (uint16_t *)a.v_cache;
              const uint16_t xb = a.kv_f16 ? float_to_half_bits(x) : float_to_bf16_bits(x), yb = a.kv_f16 ? float_to_half_bits(y) : float_to_bf16_bits(y);
              if (mi == 1) {
                const int X = a.cache_x;
                const size_t hb = ((size_t)blk * a.num_kv_heads + head) * (size_t)(a.head_dim >> a.x_shift);
                if (wr0) kc[(hb + ((unsigned)d0 >> a.x_shift)) * a.block_size * X + off * X + ((unsigned)d0 & (unsigned)(X - 1))] = xb;
                if (wr1) kc[(hb + ((unsigned)d1 >> a.x_shift)) * a.block_size * X + off * X + ((unsigned)d1 & (unsigned)(X - 1))] = yb;
              } else {
                const size_t o = (((size_t)blk * a.num_kv_heads + head) * a.head_dim + dd) * a.block_size + off;
                if (wr0) vc[o] = xb;
                if (wr1) vc[o + a.block_size] = yb;
              }
            }
          }
        }
      }
    };
    MRS_DEC_TYPE_SWITCH(jb.mat[0].type, { stream<TT, NCOLS, false, RING2>(jb, K, NCI, mode, smem, ctr, sbar, stage, auxf, epi); })
  }
}

template <int NCOLS, int EPI, int TMASK = TM_ALL, bool RING2 = false>
__global__ void __launch_bounds__(NT) dec_gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8 * 8];  // RMSNorm partials: [column][wave]
  __shared__ int ctr;           // the workgroup's unit counter
  if constexpr (EPI == EPI_STORE && NCOLS == 1) {
    __shared__ unsigned long long wbest[NW];
    unsigned long long best = 0ull;
    gemv_phase<NCOLS, EPI, TMASK, RING2>(a, smem, red, &ctr, &best);
    if (a.amax) {  // kernel-argument uniform
      const int tid = tid_opaque();
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) { const unsigned long long o = __shfl_xor(best, m, 64); best = o > best ? o : best; }
      if ((tid & 63) == 0) wbest[tid >> 6] = best;
      __syncthreads();
      if (tid == 0) {
        unsigned long long b2 = wbest[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) b2 = wbest[w] > b2 ? wbest[w] : b2;
        if (b2) atomicMax(a.amax, b2);  // one per workgroup (~12 ns each on one address); a workgroup without rows keeps 0
      }
    }
  } else gemv_phase<NCOLS, EPI, TMASK, RING2>(a, smem, red, &ctr, nullptr);
}


// launch one GEMV phase with NCOLS activation columns (ext_dec_gemv.hip, one definition per NCOLS)
template <int NCOLS> int gemv_launch(int epi, int tmask, bool ring2, int grid, size_t lds, const GemvArgs &a, hipStream_t s);

}  // namespace dec
}  // namespace mrs
