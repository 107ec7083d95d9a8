// dec_attn.cuh -- decode attention of the decode engine as wave-level device functions (used by the persistent step kernel of ext_dec.hip).
// Same arithmetic, in the same order, as decode_attn_wave_kernel<G, CT, false> + decode_attn_merge_q8_1_kernel<128, true> of
// paged_attention.cuh (so the persistent step and the launch-per-phase step produce identical bits): one wave per 32-token KV chunk range of
// one (sequence, kv head): online softmax in f32 for the G query heads of the GQA group, probabilities kept in f32 (the reference CPU path,
// attention/backends/cpu/single_q.rs), partial (m, l, o) per split; then one wave per (sequence, head) merges the splits.
// Semantics of the split / merge: pagedattention.cuh:110-486 (v2 partitions + reduce), 1 / (sum + 1e-6).
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
};

// q_s: G * 128 floats, p_s: G * 32 floats of wave-private LDS; the item covers query heads [head0, head0 + G) of kv head kvh
// blocks [b0, b1) of the sequence; sink(g, o0, o1, m, l) receives the un-normalised partial of query head head0 + g (o0 / o1: dims lane / lane + 64)
template <int G, class CT, class Sink>
__device__ __forceinline__ void attn_split_core(const AttnArgs &a, int kvh, int head0, int seq, int b0, int b1, float *q_s, float *p_s, Sink sink) {
  constexpr int HD = 128, BS = 32;
  const int lane = lane_opaque();
  const int ctx = (int)a.context_lens[seq];
  if (b0 >= b1) return;  // wave-uniform
  const float *qg = a.q + (size_t)seq * a.q_stride + (size_t)head0 * HD;
  for (int i = lane * 4; i < G * HD; i += 256) *(float4 *)(q_s + i) = *(const float4 *)(qg + i);
  MRS_WAVE_SYNC();
  const uint32_t *bt = a.block_tables + (size_t)seq * a.max_blocks_per_seq;
  const int t = lane & 31, half = lane >> 5;
  float m[G], l[G], o0[G], o1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { m[g] = -FLT_MAX; l[g] = 0.f; o0[g] = 0.f; o1[g] = 0.f; }
  for (int b = b0; b < b1; ++b) {
    const size_t base = (size_t)bt[b] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride;
    const uint16_t *kb = a.k_cache + base + (size_t)(half * 8) * BS * 8 + t * 8;
    const uint16_t *vb = a.v_cache + base;
    int4 kr[8], vr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kr[c] = *(const int4 *)(kb + (size_t)c * BS * 8);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) vr[r * 4 + c] = *(const int4 *)(vb + (size_t)(lane + 64 * r) * BS + c * 8);
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
    const bool valid = b * BS + t < ctx;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = s[g] + __shfl_xor(s[g], 32, 64);
      v = valid ? v * a.scale : -FLT_MAX;
      float mx = v;
      mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx)); mx = fmaxf(mx, dpp_f<0x140>(mx));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      const float mn = fmaxf(m[g], mx);
      const float p = valid ? __expf(v - mn) : 0.f;
      float ps = p;
      ps += dpp_f<0xB1>(ps); ps += dpp_f<0x4E>(ps); ps += dpp_f<0x141>(ps); ps += dpp_f<0x140>(ps);
      ps += __shfl_xor(ps, 16, 64);
      const float alpha = __expf(m[g] - mn);
      l[g] = l[g] * alpha + ps;
      o0[g] *= alpha; o1[g] *= alpha;
      m[g] = mn;
      if (half == 0) p_s[g * BS + t] = p;
    }
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
// ns (<= 64) partials of one head: pm / pl [ns], po [ns][128]; returns the head's output at dims lane (v0) and lane + 64 (v1)
__device__ __forceinline__ void attn_merge_core(int ns, const float *pm, const float *pl, const float *po, float &v0, float &v1) {
  constexpr int HD = 128;
  const int lane = lane_opaque();
  const float mj = lane < ns ? pm[lane] : -FLT_MAX;
  const float lj = lane < ns ? pl[lane] : 0.f;
  const float mx = wave_max(mj);
  const float r = lane < ns ? __expf(mj - mx) : 0.f;
  const float gs = wave_sum(lj * r);
  // the standalone merge kernel sums four interleaved chains over the splits: same order here
  float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f}, a3[2] = {0.f, 0.f};
  int j = 0;
  for (; j + 4 <= ns; j += 4) {
    const float r0 = __shfl(r, j, 64), r1 = __shfl(r, j + 1, 64), r2 = __shfl(r, j + 2, 64), r3 = __shfl(r, j + 3, 64);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float *pp = po + lane + 64 * h;
      a0[h] = fmaf(pp[(size_t)(j + 0) * HD], r0, a0[h]); a1[h] = fmaf(pp[(size_t)(j + 1) * HD], r1, a1[h]);
      a2[h] = fmaf(pp[(size_t)(j + 2) * HD], r2, a2[h]); a3[h] = fmaf(pp[(size_t)(j + 3) * HD], r3, a3[h]);
    }
  }
  for (; j < ns; ++j) {
    const float rj = __shfl(r, j, 64);
#pragma unroll
    for (int h = 0; h < 2; ++h) a0[h] = fmaf(po[(size_t)j * HD + lane + 64 * h], rj, a0[h]);
  }
  const float inv = 1.0f / (gs + 1e-6f);
  v0 = ((a0[0] + a1[0]) + (a2[0] + a3[0])) * inv;
  v1 = ((a0[1] + a1[1]) + (a2[1] + a3[1])) * inv;
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
