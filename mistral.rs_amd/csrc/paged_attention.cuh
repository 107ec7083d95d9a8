// paged_attention.cuh -- decode attention over the paged KV cache, gfx950 / wave64.
//
// Semantics: mistralrs-paged-attn/src/cuda/pagedattention.cuh:110-486 (v1/v2 kernel) and :516-660
// (v2 reduce): one query token per sequence; f32 logits; softmax with inv = 1/(sum + 1e-6);
// masked logits zeroed; optional ALiBi, softcap (1.0 = off), attention sinks; probabilities are
// rounded to the query dtype before P.V (ROUND_P) exactly like the reference's from_float(logits_vec).
//
// MI355X design (differs from the CUDA kernel on purpose):
//   * GQA-aware: one workgroup serves G query heads that share a KV head, so every K/V byte is read
//     from HBM once per group instead of once per query head (the reference launches one block per
//     query head: 4x (8B) .. 8x (70B) more KV traffic);
//   * wave64 mapping: a wave owns one 16/32-token KV block per iteration.  For Q.K^T lane l handles
//     token l % BS and every LPT-th 16-byte chunk of the head dim (the K layout
//     [blk][kvh][hd/x][BS][x] makes the 32 lanes of a chunk read one contiguous 512-byte span);
//     for P.V lane l handles V row l / LPR (+ 64/LPR per step) and 16 bytes of tokens, so one wave
//     load covers 1 KiB of contiguous V rows;
//   * q for the G heads lives in LDS as f32 and is re-read per chunk (broadcast ds_read_b128);
//     logits for the G heads live in LDS; softmax reductions use LDS + xor butterflies.
#pragma once
#include "common.cuh"
#include <float.h>

namespace mrs {

template <class CT> __device__ __forceinline__ void unpack16(const int4 &raw, float *out);
template <> __device__ __forceinline__ void unpack16<bf16_t>(const int4 &raw, float *o) {
  const unsigned w[4] = {(unsigned)raw.x, (unsigned)raw.y, (unsigned)raw.z, (unsigned)raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const int4 &raw, float *o) {
  const unsigned w[4] = {(unsigned)raw.x, (unsigned)raw.y, (unsigned)raw.z, (unsigned)raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = half_bits_to_float((uint16_t)(w[i] & 0xffff)); o[2 * i + 1] = half_bits_to_float((uint16_t)(w[i] >> 16)); }
}
template <> __device__ __forceinline__ void unpack16<float>(const int4 &raw, float *o) {
  o[0] = __int_as_float(raw.x); o[1] = __int_as_float(raw.y); o[2] = __int_as_float(raw.z); o[3] = __int_as_float(raw.w);
}

template <> __device__ __forceinline__ void unpack16<fp8_t>(const int4 &raw, float *o) {  // 16 E4M3 values, exact
  const unsigned w[4] = {(unsigned)raw.x, (unsigned)raw.y, (unsigned)raw.z, (unsigned)raw.w};
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = fp8_e4m3_to_float((uint8_t)(w[i >> 2] >> (8 * (i & 3))));
}
template <class CT> struct is_fp8 { static constexpr bool value = false; };
template <> struct is_fp8<fp8_t> { static constexpr bool value = true; };

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

struct PagedAttnArgs {
  float *exp_sums;    // v2: [seqs, heads, max_parts]
  float *max_logits;  // v2
  void *out;          // v1: [seqs, heads, hd] ; v2: tmp_out [seqs, heads, max_parts, hd]
  const void *q;      // [seqs, heads, hd] with row stride q_stride
  const void *k_cache, *v_cache;
  const uint32_t *block_tables, *context_lens;
  const float *alibi_slopes, *sinks;
  const float *k_scale, *v_scale;  // fp8 cache: one f32 each (device); element = T(float(fp8) * scale) (quant_utils.cuh:24-29,79-88,135-145)
  int num_heads, num_kv_heads, max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride;
  int logits_stride;  // padded tokens per head in LDS
  float scale, softcapping;
};

// NW waves per workgroup
template <class T, class CT, int HD, int BS, int G, int PART, bool ROUND_P, int NW = 4>
__global__ void __launch_bounds__(NW * 64) paged_attention_kernel(const PagedAttnArgs a) {
  constexpr int X = 16 / sizeof(CT);        // elements per 16 bytes
  constexpr int NCH = HD / X;               // 16-byte chunks of the head dim
  constexpr int LPT = 64 / BS;              // lanes per token in Q.K^T
  constexpr int LPR = BS / X;               // lanes per V row
  constexpr int RPI = 64 / LPR;             // V rows per wave step
  constexpr int NI = (HD + RPI - 1) / RPI;  // steps over the head dim
  static_assert(BS * LPT == 64 && LPR * X == BS && LPR >= 1, "unsupported block size");

  const int seq = blockIdx.y, part = blockIdx.z, max_parts = gridDim.z;
  const uint32_t ctx = a.context_lens[seq];
  if (PART > 0 && (uint32_t)(part * PART) >= ctx) return;
  const int num_ctx_blocks = (ctx + BS - 1) / BS;
  const int blocks_per_part = PART > 0 ? PART / BS : num_ctx_blocks;
  const int start_block = PART > 0 ? part * blocks_per_part : 0;
  const int end_block = min(start_block + blocks_per_part, num_ctx_blocks);
  const int start_tok = start_block * BS;
  const int ntok = min(start_tok + (end_block - start_block) * BS, (int)ctx) - start_tok;

  const int head0 = blockIdx.x * G;
  const int kvh = head0 / (a.num_heads / a.num_kv_heads);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *q_s = (float *)smem;                   // [G][HD]
  float *logits = q_s + G * HD;                 // [G][logits_stride]   (reused as out_s[NW][G][HD])
  float *red = logits + (size_t)max(G * a.logits_stride, NW * G * HD);  // [2][G][NW]
  const int LS = a.logits_stride;

  const T *q = (const T *)a.q + (size_t)seq * a.q_stride + (size_t)head0 * HD;
  for (int i = tid; i < G * HD; i += NW * 64) q_s[i] = to_f<T>(q[i]);
  __syncthreads();

  const uint32_t *block_table = a.block_tables + (size_t)seq * a.max_num_blocks_per_seq;
  const CT *kc = (const CT *)a.k_cache + (size_t)kvh * a.kv_head_stride;
  const CT *vc = (const CT *)a.v_cache + (size_t)kvh * a.kv_head_stride;
  constexpr bool FP8 = is_fp8<CT>::value;
  float ks = 1.0f, vs = 1.0f;
  if constexpr (FP8) { ks = *a.k_scale; vs = *a.v_scale; }

  // ---------------------------------------------------------------- Q.K^T
  float qk_max[G];
#pragma unroll
  for (int g = 0; g < G; ++g) qk_max[g] = -FLT_MAX;
  const int tok_in_blk = lane % BS, cpart = lane / BS;
  for (int b = start_block + wave; b < end_block; b += NW) {
    const CT *kb = kc + (size_t)block_table[b] * a.kv_block_stride + tok_in_blk * X;
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += LPT) {
      const int c = c0 + cpart;
      if (NCH % LPT == 0 || c < NCH) {
        float kf[X];
        unpack16<CT>(ld16_a16(kb + (size_t)c * BS * X), kf);
        if constexpr (FP8) {
#pragma unroll
          for (int j = 0; j < X; ++j) kf[j] = round_to<T>(kf[j] * ks);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float *qg = q_s + g * HD + c * X;
#pragma unroll
          for (int j = 0; j < X; ++j) acc[g] = fmaf(qg[j], kf[j], acc[g]);
        }
      }
    }
    const int token = b * BS + tok_in_blk;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = acc[g];
#pragma unroll
      for (int m = BS; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
      float qk = a.scale * v;
      if (a.softcapping != 1.0f) qk = tanhf(qk / a.softcapping) * a.softcapping;
      // reference quirk kept for parity: context_len is uint32_t there (pagedattention.cuh:138), so `token_idx - context_len + 1` (:283) wraps:
      // 0 for the last token, 2^32 - k for the k-th token before it (upstream vLLM: int, i.e. -k)
      if (a.alibi_slopes) { const float s = a.alibi_slopes[head0 + g]; qk += (s != 0.f) ? s * (float)((uint32_t)token - ctx + 1u) : 0.f; }
      const bool masked = token >= (int)ctx;
      if (cpart == 0) logits[g * LS + token - start_tok] = masked ? 0.f : qk;
      if (!masked) qk_max[g] = fmaxf(qk_max[g], qk);
    }
  }
  // max over the workgroup
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float m = wave_max(qk_max[g]);
    if (lane == 0) red[g * NW + wave] = m;
  }
  __syncthreads();
  float gmax[G], esum[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float m = red[g * NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[g * NW + w]);
    if (PART == 0 && a.sinks) m = fmaxf(m, a.sinks[head0 + g]);
    gmax[g] = m;
    float s = 0.f;
    for (int i = tid; i < ntok; i += NW * 64) { const float e = __expf(logits[g * LS + i] - m); logits[g * LS + i] = e; s += e; }
    s = wave_sum(s);
    if (lane == 0) red[G * NW + g * NW + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[G * NW + g * NW + w];
    if (PART == 0 && a.sinks) s += __expf(a.sinks[head0 + g] - gmax[g]);
    esum[g] = s;
    const float inv = 1.0f / (s + 1e-6f);
    for (int i = tid; i < ntok; i += NW * 64) logits[g * LS + i] *= inv;
  }
  __syncthreads();
  if (PART > 0 && tid < G) {
    const size_t o = ((size_t)seq * a.num_heads + head0 + tid) * max_parts + part;
    float m = 0.f, s = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) if (g == tid) { m = gmax[g]; s = esum[g]; }
    a.max_logits[o] = m;
    a.exp_sums[o] = s;
  }

  // ---------------------------------------------------------------- P.V
  float acc[G][NI];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[g][i] = 0.f;
  const int vrow0 = lane / LPR, vtok0 = (lane % LPR) * X;
  for (int b = start_block + wave; b < end_block; b += NW) {
    const CT *vb = vc + (size_t)block_table[b] * a.kv_block_stride + vtok0;
    const int token0 = b * BS + vtok0;
    float p[G][X];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < X; ++j) {
        const float pv = logits[g * LS + token0 - start_tok + j];
        p[g][j] = ROUND_P ? round_to<T>(pv) : pv;
      }
    const bool last = (b == num_ctx_blocks - 1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = vrow0 + i * RPI;
      if (HD % RPI == 0 || row < HD) {
        float vf[X];
        unpack16<CT>(ld16_a16(vb + (size_t)row * BS), vf);
        if constexpr (FP8) {
#pragma unroll
          for (int j = 0; j < X; ++j) vf[j] = round_to<T>(vf[j] * vs);
        }
        if (last) {
#pragma unroll
          for (int j = 0; j < X; ++j) vf[j] = (token0 + j < (int)ctx) ? vf[j] : 0.f;  // stale slots may hold NaNs
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < X; ++j) acc[g][i] = fmaf(p[g][j], vf[j], acc[g][i]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v = acc[g][i];
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) v += __shfl_xor(v, m, 64);
      acc[g][i] = v;
    }
  __syncthreads();  // logits are dead: reuse as out_s[NW][G][HD]
  float *out_s = logits;
  if (lane % LPR == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = vrow0 + i * RPI;
        if (HD % RPI == 0 || row < HD) out_s[(wave * G + g) * HD + row] = acc[g][i];
      }
  }
  __syncthreads();
  T *out = (T *)a.out;
  for (int i = tid; i < G * HD; i += NW * 64) {
    const int g = i / HD, d = i % HD;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += out_s[(w * G + g) * HD + d];
    const size_t o = PART > 0 ? (((size_t)seq * a.num_heads + head0 + g) * max_parts + part) * HD + d
                              : ((size_t)seq * a.num_heads + head0 + g) * HD + d;
    out[o] = from_f<T>(s);
  }
}

// v2 reduce: merge the per-partition partial outputs (pagedattention.cuh:516-660). grid (heads, seqs)
template <class T, int HD, int PART>
__global__ void __launch_bounds__(128) paged_attention_reduce_kernel(T *__restrict__ out, const float *__restrict__ exp_sums,
                                                                     const float *__restrict__ max_logits, const T *__restrict__ tmp_out,
                                                                     const uint32_t *__restrict__ context_lens, int max_parts,
                                                                     const float *__restrict__ sinks) {
  const int num_heads = gridDim.x, head = blockIdx.x, seq = blockIdx.y;
  const int nparts = (context_lens[seq] + PART - 1) / PART;
  const T *tmp = tmp_out + ((size_t)seq * num_heads + head) * max_parts * HD;
  T *o = out + ((size_t)seq * num_heads + head) * HD;
  if (nparts == 1 && sinks == nullptr) {
    for (int i = threadIdx.x; i < HD; i += blockDim.x) o[i] = tmp[i];
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_max = (float *)smem, *s_sum = s_max + nparts;
  __shared__ float red[4];
  const float *ml = max_logits + ((size_t)seq * num_heads + head) * max_parts;
  const float *es = exp_sums + ((size_t)seq * num_heads + head) * max_parts;
  float m = -FLT_MAX;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { s_max[i] = ml[i]; m = fmaxf(m, ml[i]); }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);
  if (sinks) m = fmaxf(m, sinks[head]);
  float gs = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { const float r = es[i] * expf(s_max[i] - m); s_sum[i] = r; gs += r; }
  gs = wave_sum(gs);
  if ((threadIdx.x & 63) == 0) red[2 + (threadIdx.x >> 6)] = gs;
  __syncthreads();
  gs = red[2] + red[3];
  if (sinks) gs += __expf(sinks[head] - m);
  const float inv = 1.0f / (gs + 1e-6f);
  for (int i = threadIdx.x; i < HD; i += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < nparts; ++j) acc += to_f<T>(tmp[(size_t)j * HD + i]) * s_sum[j] * inv;
    o[i] = from_f<T>(acc);
  }
}

// MI355X decode path: merge the split-KV partials (as paged_attention_reduce_kernel) and emit the result directly as
// Q8_1 blocks (the o_proj GEMV's activation format: same bytes as launch_mmvq_gguf_quantize_q8_1_f32 on the f32
// attention output).  grid (heads, seqs), block HD threads (HD % 32 == 0); y: [seq][stride_blocks] blocks.
template <int HD, int PART>
__global__ void __launch_bounds__(HD) paged_attention_reduce_q8_1_kernel(uint8_t *__restrict__ y, int stride_blocks, const float *__restrict__ exp_sums,
                                                                         const float *__restrict__ max_logits, const float *__restrict__ tmp_out,
                                                                         const uint32_t *__restrict__ context_lens, int max_parts) {
  const int num_heads = gridDim.x, head = blockIdx.x, seq = blockIdx.y, i = threadIdx.x;
  const int nparts = (context_lens[seq] + PART - 1) / PART;
  const float *tmp = tmp_out + ((size_t)seq * num_heads + head) * max_parts * HD;
  const float *ml = max_logits + ((size_t)seq * num_heads + head) * max_parts;
  const float *es = exp_sums + ((size_t)seq * num_heads + head) * max_parts;
  float v;
  if (nparts == 1) {
    v = tmp[i];
  } else {
    float m = -FLT_MAX;
    for (int j = 0; j < nparts; ++j) m = fmaxf(m, ml[j]);  // every thread walks the (few) partitions: no barrier needed
    float gs = 0.f, acc = 0.f;
    for (int j = 0; j < nparts; ++j) {
      const float r = es[j] * expf(ml[j] - m);
      gs += r;
      acc += tmp[(size_t)j * HD + i] * r;
    }
    v = acc * (1.0f / (gs + 1e-6f));
  }
  float amax = fabsf(v), sum = v;
#pragma unroll
  for (int mk = 16; mk > 0; mk >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, mk, 64)); sum += __shfl_xor(sum, mk, 64); }
  const float d = amax / 127.0f;
  const int e = head * HD + i;
  uint8_t *blk = y + ((size_t)seq * stride_blocks + e / 32) * 36;
  ((int8_t *)(blk + 4))[e & 31] = amax == 0.0f ? (int8_t)0 : (int8_t)roundf(v / d);
  if ((e & 31) == 0) { ((uint16_t *)blk)[0] = float_to_half_bits(d); ((uint16_t *)blk)[1] = float_to_half_bits(sum); }
}

// =====================================================================================================
// MI355X decode attention v3 ("wave per KV chunk"): the batch-1 decode step at a few hundred tokens of context is
// a latency chain, not a bandwidth problem (3 MB of KV for Llama-3-8B at ctx 768), so the kernel has NO workgroup
// barrier and no LDS logits: every wave owns `bpw` consecutive 32-token KV blocks of one (sequence, kv-head) and
// runs an online-softmax over them for all G query heads of the GQA group, reading each K/V byte once.
//   * Q.K^T: lane (t = lane & 31, half = lane >> 5) owns token t and 64 of the 128 head dims (8 coalesced 16-byte
//     chunks of the [hd/8][32][8] K layout); q (f32, G heads) is staged once per wave in LDS and read as broadcasts;
//   * P.V: lane owns V rows d = lane and lane + 64 ([hd][32] layout: one 64-byte row = 4 x 16 B), probabilities go
//     through a 512-byte wave-private LDS tile (written by the token lanes, read back as broadcasts);
//   * K and V loads of a block are issued together, before any arithmetic;
//   * output: un-normalised partial o[g][d], running max m[g] and sum l[g] per (seq, head, split) for the merge kernel.
// Semantics = pagedattention.cuh:110-486 with f32 probabilities rounded to the KV dtype before P.V (ROUND_P).
// CT: element type of the pages (bf16_t / f16_t); ROUND_P: probabilities rounded to CT before P.V (the reference GPU kernels, pagedattention.cuh:381-384)
// or kept in f32 (the reference CPU path, attention/backends/cpu/single_q.rs -- the decode engine's arithmetic)
template <int G, class CT = bf16_t, bool ROUND_P = true>
__global__ void __launch_bounds__(256) decode_attn_wave_kernel(const float *__restrict__ q, const uint16_t *__restrict__ k_cache,
                                                               const uint16_t *__restrict__ v_cache, const uint32_t *__restrict__ block_tables,
                                                               const uint32_t *__restrict__ context_lens, float *__restrict__ part_o,
                                                               float *__restrict__ part_m, float *__restrict__ part_l, int num_heads,
                                                               int num_kv_heads, int max_blocks_per_seq, int q_stride, int kv_block_stride,
                                                               int kv_head_stride, int bpw, int max_splits, float scale) {
  constexpr int HD = 128, BS = 32;
  __shared__ __attribute__((aligned(16))) float q_s[4][G * HD];  // per wave
  __shared__ __attribute__((aligned(16))) float p_s[4][G * BS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kvh = blockIdx.x, seq = blockIdx.y;
  const int split = blockIdx.z * 4 + wave;
  const int ctx = (int)context_lens[seq];
  const int nblk = (ctx + BS - 1) / BS;
  const int b0 = split * bpw, b1 = min(b0 + bpw, nblk);
  if (b0 >= nblk) return;  // wave-uniform; no barriers anywhere in this kernel
  const int head0 = kvh * G;
  // q -> LDS (wave-private)
  const float *qg = q + (size_t)seq * q_stride + (size_t)head0 * HD;
  for (int i = lane * 4; i < G * HD; i += 256) *(float4 *)(q_s[wave] + i) = *(const float4 *)(qg + i);
  const uint32_t *bt = block_tables + (size_t)seq * max_blocks_per_seq;
  const int t = lane & 31, half = lane >> 5;
  float m[G], l[G], o0[G], o1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { m[g] = -FLT_MAX; l[g] = 0.f; o0[g] = 0.f; o1[g] = 0.f; }
  for (int b = b0; b < b1; ++b) {
    const size_t base = (size_t)bt[b] * kv_block_stride + (size_t)kvh * kv_head_stride;
    const uint16_t *kb = k_cache + base + (size_t)(half * 8) * BS * 8 + t * 8;
    const uint16_t *vb = v_cache + base;
    int4 kr[8], vr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kr[c] = *(const int4 *)(kb + (size_t)c * BS * 8);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) vr[r * 4 + c] = *(const int4 *)(vb + (size_t)(lane + 64 * r) * BS + c * 8);
    // ---- scores for token t, all G heads
    float s[G];
#pragma unroll
    for (int g = 0; g < G; ++g) s[g] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float kf[8];
      unpack16<CT>(kr[c], kf);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 qa = *(const float4 *)(q_s[wave] + g * HD + (half * 8 + c) * 8);
        const float4 qb = *(const float4 *)(q_s[wave] + g * HD + (half * 8 + c) * 8 + 4);
        s[g] = fmaf(qa.x, kf[0], s[g]); s[g] = fmaf(qa.y, kf[1], s[g]); s[g] = fmaf(qa.z, kf[2], s[g]); s[g] = fmaf(qa.w, kf[3], s[g]);
        s[g] = fmaf(qb.x, kf[4], s[g]); s[g] = fmaf(qb.y, kf[5], s[g]); s[g] = fmaf(qb.z, kf[6], s[g]); s[g] = fmaf(qb.w, kf[7], s[g]);
      }
    }
    const bool valid = b * BS + t < ctx;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = s[g] + __shfl_xor(s[g], 32, 64);  // the two dim-halves of token t
      v = valid ? v * scale : -FLT_MAX;
      // block max over the 32 tokens (all DPP inside 16 lanes, one swizzle across the two rows of 16)
      float mx = v;
      mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx)); mx = fmaxf(mx, dpp_f<0x140>(mx));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      const float mn = fmaxf(m[g], mx);
      const float p = valid ? __expf(v - mn) : 0.f;
      const float pr = ROUND_P ? round_to<CT>(p) : p;  // reference GPU kernels: probabilities cast to the KV dtype before P.V
      float ps = p;
      ps += dpp_f<0xB1>(ps); ps += dpp_f<0x4E>(ps); ps += dpp_f<0x141>(ps); ps += dpp_f<0x140>(ps);
      ps += __shfl_xor(ps, 16, 64);
      const float alpha = __expf(m[g] - mn);
      l[g] = l[g] * alpha + ps;
      o0[g] *= alpha; o1[g] *= alpha;
      m[g] = mn;
      if (half == 0) p_s[wave][g * BS + t] = pr;
    }
    // ---- P.V for rows d = lane, lane + 64 (wave-private LDS: same-wave LDS ops are ordered, no barrier needed)
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
          const float4 pa = *(const float4 *)(p_s[wave] + g * BS + c * 8);
          const float4 pb = *(const float4 *)(p_s[wave] + g * BS + c * 8 + 4);
          float acc = r == 0 ? o0[g] : o1[g];
          acc = fmaf(pa.x, vf[0], acc); acc = fmaf(pa.y, vf[1], acc); acc = fmaf(pa.z, vf[2], acc); acc = fmaf(pa.w, vf[3], acc);
          acc = fmaf(pb.x, vf[4], acc); acc = fmaf(pb.y, vf[5], acc); acc = fmaf(pb.z, vf[6], acc); acc = fmaf(pb.w, vf[7], acc);
          if (r == 0) o0[g] = acc; else o1[g] = acc;
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t pi = ((size_t)seq * num_heads + head0 + g) * max_splits + split;
    part_o[pi * HD + lane] = o0[g];
    part_o[pi * HD + lane + 64] = o1[g];
    if (lane == 0) { part_m[pi] = m[g]; part_l[pi] = l[g]; }
  }
}

// merge of the per-split partials + Q8_1 quantisation of the attention output (o_proj's activation format).
// grid (heads, seqs), 128 threads (thread = head dim)
// F32OUT: y = float [seqs][heads * HD] (decode engine: the o_proj prologue quantizes); else Q8_1 blocks
template <int HD, bool F32OUT = false>
__global__ void __launch_bounds__(HD) decode_attn_merge_q8_1_kernel(uint8_t *__restrict__ y, int stride_blocks, const float *__restrict__ part_o,
                                                                     const float *__restrict__ part_m, const float *__restrict__ part_l,
                                                                     const uint32_t *__restrict__ context_lens, int bpw, int max_splits) {
  const int num_heads = gridDim.x, head = blockIdx.x, seq = blockIdx.y, i = threadIdx.x;
  const int nblk = ((int)context_lens[seq] + 31) / 32;
  const int ns = (nblk + bpw - 1) / bpw;  // <= 64 (DEC_MAX_SPLITS)
  const size_t p0 = ((size_t)seq * num_heads + head) * max_splits;
  // split weights r_j = exp(m_j - max m) once per workgroup (lane j <-> split j), then ONE pass over the partial outputs
  // with independent loads in flight (the naive per-thread loop is a chain of ns dependent L2 round trips)
  __shared__ float r_s[64];
  __shared__ float gs_s;
  if (i < 64) {
    const float mj = i < ns ? part_m[p0 + i] : -FLT_MAX;
    const float lj = i < ns ? part_l[p0 + i] : 0.f;
    const float mx = wave_max(mj);
    const float r = i < ns ? __expf(mj - mx) : 0.f;
    r_s[i] = r;
    const float g = wave_sum(lj * r);
    if (i == 0) gs_s = g;
  }
  __syncthreads();
  const float gs = gs_s;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float *po = part_o + p0 * HD + i;
  int j = 0;
  for (; j + 4 <= ns; j += 4) {
    const float v0 = po[(size_t)(j + 0) * HD], v1 = po[(size_t)(j + 1) * HD], v2 = po[(size_t)(j + 2) * HD], v3 = po[(size_t)(j + 3) * HD];
    a0 = fmaf(v0, r_s[j], a0); a1 = fmaf(v1, r_s[j + 1], a1); a2 = fmaf(v2, r_s[j + 2], a2); a3 = fmaf(v3, r_s[j + 3], a3);
  }
  for (; j < ns; ++j) a0 = fmaf(po[(size_t)j * HD], r_s[j], a0);
  const float acc = (a0 + a1) + (a2 + a3);
  const float v = acc * (1.0f / (gs + 1e-6f));
  if constexpr (F32OUT) {
    ((float *)y)[((size_t)seq * num_heads + head) * HD + i] = v;
    return;
  }
  float amax = fabsf(v), sum = v;
#pragma unroll
  for (int mk = 16; mk > 0; mk >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, mk, 64)); sum += __shfl_xor(sum, mk, 64); }
  const float d = amax / 127.0f;
  const int e = head * HD + i;
  uint8_t *blk = y + ((size_t)seq * stride_blocks + e / 32) * 36;
  ((int8_t *)(blk + 4))[e & 31] = amax == 0.0f ? (int8_t)0 : (int8_t)roundf(v / d);
  if ((e & 31) == 0) { ((uint16_t *)blk)[0] = float_to_half_bits(d); ((uint16_t *)blk)[1] = float_to_half_bits(sum); }
}

}  // namespace mrs
