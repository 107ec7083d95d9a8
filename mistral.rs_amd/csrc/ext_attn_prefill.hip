// ext_attn_prefill.hip -- causal prompt attention on the bf16 matrix cores, reading K / V straight from the paged cache.
//
// Role in the reference: Sdpa::run_attention on the prompt chunk (mistralrs-core/src/attention/mod.rs:254-372; FlashAttention-2 on
// CUDA) inside PagedAttention::forward's prompt branch (paged_attention/layers/paged_attention.rs:1413-1475); semantics
// softmax(scale * Q K^T + causal mask) V with GQA, f32 softmax.  Here the chunk's K / V have already been scattered into the pages
// (reshape_and_cache, bf16), so the same kernel serves a first chunk and later chunks with a cached prefix.
//
// MI355X design: one wave = 32 queries of one head; a workgroup = the G query heads of one KV head (GQA: the waves read the same
// K / V lines).  The paged layouts ARE MFMA operand layouts:
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: A = K block (32 keys x 16 dims): lane (key t, half) needs 8 consecutive dims of
//     key t = ONE 16-byte piece of the [hd/8][32][8] K layout, loaded from HBM/L2 directly into the operand registers; B = Q^T kept
//     in registers for the whole key loop.  In the result a lane holds 16 keys of ONE query, so the online softmax is register math
//     plus one exchange with lane ^ 32;
//   * O += P V: the 8 keys a lane holds per half-block, taken in the order they sit in the S^T registers, ARE a valid A fragment (the
//     key order inside a 16-key MFMA step is free as long as V uses the same order), and the matching B fragment is two 8-byte runs
//     of the [hd][32] V layout.  No LDS staging of K, V or P; LDS only broadcasts the per-query rescale factors.
#include "common.cuh"
#include <float.h>

namespace mrs {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

struct PrefillAttnArgs {
  const float *q;             // [T][q_stride] f32 (RoPE applied)
  const uint16_t *k_cache;    // bf16 [blocks][kvh][hd/8][32][8]
  const uint16_t *v_cache;    // bf16 [blocks][kvh][hd][32]
  const uint32_t *block_table;  // [max_blocks] of this sequence
  float *out;                 // [T][o_stride] f32
  int T, start_pos, num_heads, num_kv_heads, q_stride, o_stride, kv_block_stride, kv_head_stride;
  float scale;
  int window;  // > 0: sliding window (key <= pos - window is masked, paged_attention/layers/paged_attention.rs:551-553); 0 = causal only
};

template <int G>
__global__ void __launch_bounds__(64 * G) prefill_attn_kernel(const PrefillAttnArgs a) {
  constexpr int HD = 128, BS = 32;
  __shared__ __attribute__((aligned(16))) float bc[G][32];  // per-wave broadcast of per-query factors
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qt = blockIdx.x, kvh = blockIdx.y;
  const int head = kvh * G + wave;
  const int ql = lane & 31, kh = lane >> 5;
  const int t_q = min(qt * 32 + ql, a.T - 1);        // query token of this lane's S^T column (clamped; masked at the store)
  const int p_q = a.start_pos + t_q;                  // its absolute position
  const int total_len = a.start_pos + a.T;            // keys [0, total_len) are valid in the cache
  const int last_q_pos = a.start_pos + min(qt * 32 + 31, a.T - 1);
  const int nkb = last_q_pos / BS + 1;                // key blocks this tile attends
  const int first_q_pos = a.start_pos + qt * 32;
  const int kb0 = a.window > 0 && first_q_pos >= a.window ? (first_q_pos - a.window + 1) / BS : 0;  // blocks before it are too old for every query of the tile

  // Q^T fragments: 8 d-steps of 16 dims; lane (query ql, half kh) holds dims dstep*16 + kh*8 .. +8
  bf16x8 qf[8];
  {
    const float *qp = a.q + (size_t)t_q * a.q_stride + (size_t)head * HD + kh * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      const float4 x = *(const float4 *)(qp + ds * 16), y = *(const float4 *)(qp + ds * 16 + 4);
      const unsigned w0 = pk_bf16(x.x, x.y), w1 = pk_bf16(x.z, x.w), w2 = pk_bf16(y.x, y.y), w3 = pk_bf16(y.z, y.w);
      qf[ds] = __builtin_bit_cast(bf16x8, make_uint4(w0, w1, w2, w3));
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -FLT_MAX, l = 0.f;

  for (int kb = kb0; kb < nkb; ++kb) {
    const size_t base = (size_t)a.block_table[kb] * a.kv_block_stride + (size_t)kvh * a.kv_head_stride;
    // ---- S^T = K Q^T (32 keys x 32 queries)
    const uint16_t *kp = a.k_cache + base + (size_t)(kh * BS + ql) * 8;
    int4 kr[8];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) kr[ds] = *(const int4 *)(kp + (size_t)ds * 2 * BS * 8);
    // V fragments for this block (issued now, consumed after the softmax): lane (d = dt*32 + ql, kh), PV step s
    const uint16_t *vp = a.v_cache + base + (size_t)ql * BS + 4 * kh;
    int2 vr[4][2][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        vr[dt][s][0] = *(const int2 *)(vp + (size_t)dt * 32 * BS + 16 * s);
        vr[dt][s][1] = *(const int2 *)(vp + (size_t)dt * 32 * BS + 16 * s + 8);
      }
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kr[ds]), qf[ds], sacc, 0, 0, 0);
    // ---- online softmax for query column ql: this lane holds keys kr_ = (r & 3) + 8 (r >> 2) + 4 kh of the block
    float sv[16], mx = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kpos = kb * BS + (r & 3) + 8 * (r >> 2) + 4 * kh;
      sv[r] = kpos <= p_q && (a.window <= 0 || kpos + a.window > p_q) ? sacc[r] * a.scale : -FLT_MAX;
      mx = fmaxf(mx, sv[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.f, p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = sv[r] > -FLT_MAX ? __expf(sv[r] - mn) : 0.f; ps += p[r]; }
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mn;
    // ---- rescale O: accumulator register r of a lane belongs to query row (r & 3) + 8 (r >> 2) + 4 kh
    if (kh == 0) bc[wave][ql] = alpha;
    float ar[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 t4 = *(const float4 *)(&bc[wave][8 * g4 + 4 * kh]);
      ar[4 * g4] = t4.x; ar[4 * g4 + 1] = t4.y; ar[4 * g4 + 2] = t4.z; ar[4 * g4 + 3] = t4.w;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= ar[r];
    // ---- O += P V : P fragments are the S^T registers in place (rounded to bf16 = the KV dtype, as the reference does)
    const bool tail = (kb + 1) * BS > total_len;  // block with slots past the end of the sequence: their V may be stale (NaN)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pk_bf16(p[8 * s], p[8 * s + 1]), pk_bf16(p[8 * s + 2], p[8 * s + 3]),
                                                              pk_bf16(p[8 * s + 4], p[8 * s + 5]), pk_bf16(p[8 * s + 6], p[8 * s + 7])));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        int2 v0 = vr[dt][s][0], v1 = vr[dt][s][1];
        if (tail) {  // wave-uniform; zero the key slots >= total_len (4 keys per int2: tokens t0 .. t0+3)
          const int t0 = kb * BS + 16 * s + 4 * kh;
          auto zap = [&](int2 v, int tk) {
            if (tk + 0 >= total_len) v.x &= 0xffff0000; if (tk + 1 >= total_len) v.x &= 0x0000ffff;
            if (tk + 2 >= total_len) v.y &= 0xffff0000; if (tk + 3 >= total_len) v.y &= 0x0000ffff;
            return v;
          };
          v0 = zap(v0, t0); v1 = zap(v1, t0 + 8);
        }
        const bf16x8 vf = __builtin_bit_cast(bf16x8, make_int4(v0.x, v0.y, v1.x, v1.y));
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o[dt], 0, 0, 0);
      }
    }
  }
  // ---- normalise and store: out[query row][head*128 + dt*32 + ql]
  if (kh == 0) bc[wave][ql] = 1.0f / (l + 1e-6f);
  float inv[16];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 t4 = *(const float4 *)(&bc[wave][8 * g4 + 4 * kh]);
    inv[4 * g4] = t4.x; inv[4 * g4 + 1] = t4.y; inv[4 * g4 + 2] = t4.z; inv[4 * g4 + 3] = t4.w;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (t < a.T) {
      float *op = a.out + (size_t)t * a.o_stride + (size_t)head * HD + ql;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) op[dt * 32] = o[dt][r] * inv[r];
    }
  }
}

}  // namespace mrs

// q [T][q_stride] f32 over a bf16 paged cache (block 32, head_dim 128), positions start_pos .. start_pos+T-1, keys 0 .. start_pos+T-1
// already in the cache.  out [T][o_stride] f32.  Returns 0, -1 for unsupported shapes (caller falls back to paged_attention).
extern "C" int mrs_prefill_attention_window_f32_bf16(const float *q, const void *key_cache, const void *value_cache, const uint32_t *block_table,
                                                     float *out, int T, int start_pos, int num_heads, int num_kv_heads, int head_size, int block_size,
                                                     int q_stride, int o_stride, int kv_block_stride, int kv_head_stride, float scale, int sliding_window, void *stream);
extern "C" int mrs_prefill_attention_f32_bf16(const float *q, const void *key_cache, const void *value_cache, const uint32_t *block_table,
                                              float *out, int T, int start_pos, int num_heads, int num_kv_heads, int head_size, int block_size,
                                              int q_stride, int o_stride, int kv_block_stride, int kv_head_stride, float scale, void *stream) {
  return mrs_prefill_attention_window_f32_bf16(q, key_cache, value_cache, block_table, out, T, start_pos, num_heads, num_kv_heads, head_size, block_size, q_stride, o_stride,
                                               kv_block_stride, kv_head_stride, scale, 0, stream);
}
// sliding_window > 0: query at position p attends keys (p - W, p] only (Mistral; eager_attention_mask / the prompt mask of
// mistralrs-core/src/paged_attention/layers/paged_attention.rs:551-553); key blocks that are too old for every query of a tile are skipped
extern "C" int mrs_prefill_attention_window_f32_bf16(const float *q, const void *key_cache, const void *value_cache, const uint32_t *block_table,
                                                     float *out, int T, int start_pos, int num_heads, int num_kv_heads, int head_size, int block_size,
                                                     int q_stride, int o_stride, int kv_block_stride, int kv_head_stride, float scale, int sliding_window, void *stream) {
  using namespace mrs;
  if (T <= 0) return 0;
  if (head_size != 128 || block_size != 32 || num_heads % num_kv_heads) return -1;
  const int G = num_heads / num_kv_heads;
  PrefillAttnArgs a{q, (const uint16_t *)key_cache, (const uint16_t *)value_cache, block_table, out, T, start_pos, num_heads, num_kv_heads,
                    q_stride, o_stride, kv_block_stride, kv_head_stride, scale, sliding_window > 0 ? sliding_window : 0};
  const dim3 grid((T + 31) / 32, num_kv_heads);
  hipStream_t s = (hipStream_t)stream;
  switch (G) {
  case 1: hipLaunchKernelGGL(prefill_attn_kernel<1>, grid, dim3(64), 0, s, a); break;
  case 2: hipLaunchKernelGGL(prefill_attn_kernel<2>, grid, dim3(128), 0, s, a); break;
  case 4: hipLaunchKernelGGL(prefill_attn_kernel<4>, grid, dim3(256), 0, s, a); break;
  case 8: hipLaunchKernelGGL(prefill_attn_kernel<8>, grid, dim3(512), 0, s, a); break;
  default: return -1;
  }
  return 0;
}
