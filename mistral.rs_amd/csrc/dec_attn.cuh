// dec_attn.cuh -- decode attention of the decode engine as wave-level device functions: the split kernel with the last-arriver merge
// (ext_dec.hip dec_attn2_kernel, the default), the persistent step kernel and the one-launch short-context kernel all call these, so every
// engine path produces the same bits.  One wave per range of 32-token KV blocks of one (sequence, kv head): online softmax in f32 for the G query
// heads of the GQA group, probabilities kept in f32, partial (m, l, o) per split; then the splits are merged per head.
// Arithmetic = the reference CPU attention (mistralrs-core/src/attention/backends/cpu/single_q.rs): tile max, correction and probabilities
// through the reference's fast_exp (elem.rs:417-433, common.cuh fast_exp_ref), merge as run_barrier (:108-157: weights exp(m_c - m_all), sums and
// outputs accumulated chunk by chunk with separate multiply and add, out = acc * (1 / s_all), no epsilon); the f32 summation ORDER inside a
// block is the kernel's own and is written down in oracle/cpu_path_oracle.c orc_attention_engine, which the kernels equal bit for bit.
#pragma once
#include "paged_attention.cuh"

namespace mrs {
namespace dec {

struct AttnArgs {
  const float *q;            // [seqs][q_stride]
  const uint16_t *k_cache, *v_cache;
  const uint32_t *block_tables, *context_lens;
  float *part_o, *part_m, *part_l;  // [seqs][heads][max_splits]([128])
  float *out;                // [seqs][heads * 128]
  int num_heads, num_kv_heads, max_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, bpw, max_splits, num_seqs;
  float scale;
  int window;  // > 0: a query attends the last `window` positions only, itself included (Mistral sliding window; mask rule of
               // mistralrs-core/src/paged_attention/layers/paged_attention.rs:551-553: key <= pos - window is too old); 0 = all
};

// One 32-token KV block of one kv head in registers: lane (t = lane & 31, half = lane >> 5) holds K dims [(half * 8 + c) * 8, +8) of token t in kr[c];
// lane d holds V dims d (vr[0..3]) and d + 64 (vr[4..7]) of the 32 tokens.
template <bool WITH_V>
__device__ __forceinline__ void attn_load_block(const AttnArgs &a, size_t base, int4 (&kr)[8], int4 (&vr)[8]) {
  constexpr int BS = 32;
  const int lane = lane_opaque(), t = lane & 31, half = lane >> 5;
  const uint16_t *kb = a.k_cache + base + (size_t)(half * 8) * BS * 8 + t * 8;
  const uint16_t *vb = a.v_cache + base;
#pragma unroll
  for (int c = 0; c < 8; ++c) kr[c] = *(const int4 *)(kb + (size_t)c * BS * 8);
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) vr[r * 4 + c] = *(const int4 *)(vb + (size_t)(lane + 64 * r) * BS + c * 8);
  }
}
// The online-softmax update of one block for the G query heads in q_s (G * 128 floats of wave-private LDS; p_s: G * 32): scores by fma chains over the lane's 64
// dims + the other half, block max / sum trees, fast_exp, (WITH_V) o = o * alpha + P.V by fma chains.  `first` = first block of the split (l = o = 0: no correction).
// EVERY engine attention path (decode split kernel, prompt kernel) goes through this function: same bits.
template <int G, class CT, bool WITH_V>
__device__ __forceinline__ void attn_block_update(const AttnArgs &a, const int4 (&kr)[8], const int4 (&vr)[8], const float *q_s, float *p_s, int b, bool first, int ctx,
                                                  int lo, float (&m)[G], float (&l)[G], float (&o0)[G], float (&o1)[G]) {
  constexpr int HD = 128, BS = 32;
  const int lane = lane_opaque(), t = lane & 31, half = lane >> 5;
  float s[G];
#pragma unroll
  for (int g = 0; g < G; ++g) s[g] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float kf[8];
    unpack16<CT>(kr[c], kf);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 qa = *(const float4 *)(q_s + g * HD + (half * 8 + c) * 8);
      const float4 qb = *(const float4 *)(q_s + g * HD + (half * 8 + c) * 8 + 4);
      s[g] = fmaf(qa.x, kf[0], s[g]); s[g] = fmaf(qa.y, kf[1], s[g]); s[g] = fmaf(qa.z, kf[2], s[g]); s[g] = fmaf(qa.w, kf[3], s[g]);
      s[g] = fmaf(qb.x, kf[4], s[g]); s[g] = fmaf(qb.y, kf[5], s[g]); s[g] = fmaf(qb.z, kf[6], s[g]); s[g] = fmaf(qb.w, kf[7], s[g]);
    }
  }
  const bool valid = b * BS + t < ctx && b * BS + t >= lo;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float v = s[g] + __shfl_xor(s[g], 32, 64);
    v = valid ? v * a.scale : -FLT_MAX;
    float mx = v;
    mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx)); mx = fmaxf(mx, dpp_f<0x140>(mx));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    const float mn = fmaxf(m[g], mx);
    const float p = valid ? fast_exp_ref(v - mn) : 0.f;
    float ps = p;
    ps += dpp_f<0xB1>(ps); ps += dpp_f<0x4E>(ps); ps += dpp_f<0x141>(ps); ps += dpp_f<0x140>(ps);
    ps += __shfl_xor(ps, 16, 64);
    if (first) {  // first block of the split: l = o = 0, so l * alpha + ps == ps and o * alpha == 0 bit for bit -- no correction to compute
      l[g] = ps;
    } else {
      const float alpha = fast_exp_ref(m[g] - mn);
      l[g] = l[g] * alpha + ps;
      if (WITH_V) { o0[g] *= alpha; o1[g] *= alpha; }
    }
    m[g] = mn;
    if (WITH_V && half == 0) p_s[g * BS + t] = p;
  }
  if (!WITH_V) return;
  MRS_WAVE_SYNC();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float vf[8];
      unpack16<CT>(vr[r * 4 + c], vf);
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = (b * BS + c * 8 + j < ctx) ? vf[j] : 0.f;  // stale slots may hold NaNs
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 pa = *(const float4 *)(p_s + g * BS + c * 8);
        const float4 pb = *(const float4 *)(p_s + g * BS + c * 8 + 4);
        float acc = r == 0 ? o0[g] : o1[g];
        acc = fmaf(pa.x, vf[0], acc); acc = fmaf(pa.y, vf[1], acc); acc = fmaf(pa.z, vf[2], acc); acc = fmaf(pa.w, vf[3], acc);
        acc = fmaf(pb.x, vf[4], acc); acc = fmaf(pb.y, vf[5], acc); acc = fmaf(pb.z, vf[6], acc); acc = fmaf(pb.w, vf[7], acc);
        if (r == 0) o0[g] = acc; else o1[g] = acc;
      }
    }
  }
  MRS_WAVE_SYNC();
}

// q_s: G * 128 floats, p_s: G * 32 floats of wave-private LDS; the item covers query heads [head0, head0 + G) of kv head kvh
// blocks [b0, b1) of the sequence; sink(g, o0, o1, m, l) receives the un-normalised partial of query head head0 + g (o0 / o1: dims lane / lane + 64)
template <int G, class CT, class Sink>
__device__ __forceinline__ void attn_split_core(const AttnArgs &a, int kvh, int head0, int seq, int b0, int b1, float *q_s, float *p_s, Sink sink, int first_page = -1) {
  constexpr int HD = 128;
  const int lane = lane_opaque();
  const int ctx = (int)a.context_lens[seq];
  const int lo = a.window > 0 && ctx > a.window ? ctx - a.window : 0;  // first position inside the window (the query sits at ctx - 1)
  if (b0 >= b1) return;  // wave-uniform
  const uint32_t *bt = a.block_tables + (size_t)seq * a.max_blocks_per_seq;
  // the first block's K / V (HBM: evicted by 4.7 GB of weights since the last token) are requested BEFORE the query is staged through LDS: the two latencies overlap
  // instead of adding up (round 5; the query rows are L2-resident, written by the qkv launch just before)
  int4 kr[8], vr[8];
  {
    const size_t base0 = (size_t)(first_page >= 0 ? (unsigned)first_page : bt[b0]) * a.kv_block_stride + (size_t)kvh * a.kv_head_stride;  // first_page: loaded by the caller ahead of the context length
    attn_load_block<true>(a, base0, kr, vr);
  }
  const float *qg = a.q + (size_t)seq * a.q_stride + (size_t)head0 * HD;
  for (int i = lane * 4; i < G * HD; i += 256) *(float4 *)(q_s + i) = *(const float4 *)(qg + i);
  MRS_WAVE_SYNC();
  float m[G], l[G], o0[G], o1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { m[g] = -FLT_MAX; l[g] = 0.f; o0[g] = 0.f; o1[g] = 0.f; }
  for (int b = b0; b < b1; ++b) {
    if (b > b0) {
      const size_t base = (size_t)bt[b] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride;
      attn_load_block<true>(a, base, kr, vr);
    }
    attn_block_update<G, CT, true>(a, kr, vr, q_s, p_s, b, b == b0, ctx, lo, m, l, o0, o1);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) sink(g, o0[g], o1[g], m[g], l[g]);
}
template <int G, class CT>
__device__ __forceinline__ void attn_split_item(const AttnArgs &a, int kvh, int head0, int seq, int split, float *q_s, float *p_s) {
  const int lane = lane_opaque();
  const int nblk = ((int)a.context_lens[seq] + 31) / 32;
  const int b0 = split * a.bpw, b1 = min(b0 + a.bpw, nblk);
  attn_split_core<G, CT>(a, kvh, head0, seq, b0, b1, q_s, p_s, [&](int g, float o0, float o1, float m, float l) {
    const size_t pi = ((size_t)seq * a.num_heads + head0 + g) * a.max_splits + split;
    a.part_o[pi * 128 + lane] = o0;
    a.part_o[pi * 128 + lane + 64] = o1;
    if (lane == 0) { a.part_m[pi] = m; a.part_l[pi] = l; }
  });
}

// one wave merges the splits of (seq, head): lane j <-> split j for the weights, lane d / d + 64 for the output dims
// ns (<= 64) partials of one head: pm / pl [ns], po [ns][128]; returns the head's output at dims lane (v0) and lane + 64 (v1).
// Order (single_q.rs:108-157): w_j = fast_exp(m_j - max m); s += l_j * w_j and acc += o_j * w_j for j ascending, multiply and add separate.
__device__ __forceinline__ void attn_merge_core(int ns, const float *pm, const float *pl, const float *po, float &v0, float &v1) {
  constexpr int HD = 128;
  const int lane = lane_opaque();
  const float mj = lane < ns ? pm[lane] : -FLT_MAX;
  const float lj = lane < ns ? pl[lane] : 0.f;
  const float mx = wave_max(mj);
  const float w = fast_exp_ref(mj - mx);
  float s_all = 0.f, a0 = 0.f, a1 = 0.f;
  for (int j = 0; j < ns; ++j) {
    const float wj = __shfl(w, j, 64), lw = __shfl(lj, j, 64) * wj;
    s_all = s_all + lw;
    const float t0 = po[(size_t)j * HD + lane] * wj, t1 = po[(size_t)j * HD + lane + 64] * wj;
    a0 = a0 + t0;
    a1 = a1 + t1;
  }
  const float inv = 1.0f / s_all;
  v0 = a0 * inv;
  v1 = a1 * inv;
}
__device__ __forceinline__ void attn_merge_item(const AttnArgs &a, int head, int seq) {
  const int lane = lane_opaque();
  const int nblk = ((int)a.context_lens[seq] + 31) / 32;
  const int ns = (nblk + a.bpw - 1) / a.bpw;  // <= 64
  const size_t p0 = ((size_t)seq * a.num_heads + head) * a.max_splits;
  float v0, v1;
  attn_merge_core(ns, a.part_m + p0, a.part_l + p0, a.part_o + p0 * 128, v0, v1);
  a.out[((size_t)seq * a.num_heads + head) * 128 + lane] = v0;
  a.out[((size_t)seq * a.num_heads + head) * 128 + lane + 64] = v1;
}

}  // namespace dec
}  // namespace mrs
