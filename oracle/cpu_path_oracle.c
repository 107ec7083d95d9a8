/*
 * oracle/cpu_path_oracle.c -- TEST INFRASTRUCTURE ONLY (see ggml_oracle.h).
 *
 * Round 3: the parts of the reference CPU decode path that are IN /root/reference, restated line by line, plus the decode
 * engine's own f32 summation orders written down as plain C so that "engine == this file" can be asserted bit for bit.
 *
 * (1) "cpu" order -- what mistralrs-core runs on a CPU device:
 *       attention   mistralrs-core/src/attention/backends/cpu/single_q.rs:57-157 (run_barrier: per kv chunk partials + merge),
 *                   :161-275 (compute_group_range: TILE = 128 positions, tile max, fast_exp correction, softmax tile, P.V by mad),
 *                   elem.rs:417-433 (fast_exp, Cephes polynomial), elem.rs:366-381 / 398-413 (portable dot: chunks of four
 *                   products summed left to right, then added to the running sum), elem.rs:383-396 (scale / mad).
 *                   The x86 / aarch64 builds replace dot / mad / max / softmax_row by SIMD kernels (avx.rs, neon.rs) whose lane
 *                   counts fix OTHER f32 association orders; the portable bodies restated here are the ones elem.rs itself
 *                   carries, and the number of kv chunks depends on the host's thread count (single_q.rs:79-83) -- it is a
 *                   parameter here.  So the reference's own result is reproducible only up to f32 summation order; that spread
 *                   is what tests and bench.py report next to the engine's distance.
 *       rms_norm    candle_nn::ops::rms_norm CPU kernel (candle-nn 0.9 ops.rs, CustomOp2 RmsNorm::cpu_fwd; candle is a git
 *                   dependency, not vendored: restated from its published source): sum of squares in f32 in element order,
 *                   m = sqrt(sum / d + eps), y_i = x_i / m * w_i.   Call site: mistralrs-core/src/layers.rs:403-414.
 *       SiLU        x / (1 + exp(-x)) with libm expf (mistralrs-quant/src/utils/ops.rs:2601-2612) -- ggml_oracle.c orc_glu_act.
 * (2) "engine" order -- the same operations with the summation trees of mistral.rs_amd/csrc (dec_core2.cuh act_finish,
 *     dec_attn2.cuh): every function below states the order in its comment; tests/test_dec_engine.py and test_dec_model.py assert
 *     bit equality between the HIP kernels and these functions (GEMV: orc_gemv_engine at the end of this file).
 */
#include "ggml_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ elem.rs:417-433 */
float orc_fast_exp(float x) {
  const float LOG2E = 1.44269504088896340736f; /* std::f32::consts::LOG2_E */
  const float C0 = 0.6933594f, C1 = -2.1219444e-4f; /* 0.693_359_4, -2.121_944_4e-4 */
  x = x < -87.0f ? -87.0f : (x > 87.0f ? 87.0f : x);
  const float z = roundf(x * LOG2E); /* f32::round = half away from zero */
  const float r = x - z * C0 - z * C1;
  const float r2 = r * r;
  const float p = r + r2 * (0.5f + r * (0.16666546f + r * (0.041665795f + r * (0.00833345f + r * 0.0013920345f))));
  const int32_t zi = (int32_t)z;
  uint32_t bits = (uint32_t)((zi + 127) << 23);
  float e;
  memcpy(&e, &bits, 4);
  return e * (1.0f + p);
}

/* elem.rs:366-381: dot_f32, portable body */
static float dot_f32_portable(const float *a, const float *b, int n) {
  float sum = 0.0f;
  const int chunks = n / 4;
  for (int c = 0; c < chunks; ++c) {
    const int i = c * 4;
    sum += a[i] * b[i] + a[i + 1] * b[i + 1] + a[i + 2] * b[i + 2] + a[i + 3] * b[i + 3];
  }
  for (int i = chunks * 4; i < n; ++i) sum += a[i] * b[i];
  return sum;
}

/* single_q.rs:161-275 compute_group_range for the `group` query rows h0 .. h0+group-1 of one kv head over positions [kv_start, kv_end).
 * rows: group x (dv + 2) floats: vkq[dv], running max, running sum.   k, v: [kv_len][KVH][hd] f32. */
static void compute_group_range(const float *q, const float *k, const float *v, int H, int KVH, int hd, float scale, int h0, int group,
                                int kv_start, int kv_end, float *rows) {
  enum { TILE = 128, MAXG = 8 };
  const int stride = hd + 2, rk2 = H / KVH, k_head = h0 / rk2;
  float m[MAXG], s[MAXG];
  float s_tile[MAXG * TILE];
  for (int j = 0; j < group; ++j) { m[j] = -INFINITY; s[j] = 0.0f; }
  for (int bs = kv_start; bs < kv_end; bs += TILE) {
    const int be = kv_end < bs + TILE ? kv_end : bs + TILE, bn = be - bs;
    for (int j = 0; j < group; ++j) {
      const float *q_row = q + (size_t)(h0 + j) * hd;
      float *tile = s_tile + j * TILE;
      for (int p = bs; p < be; ++p) tile[p - bs] = dot_f32_portable(q_row, k + ((size_t)p * KVH + k_head) * hd, hd) * scale;
      float bmax = -INFINITY;
      for (int i = 0; i < bn; ++i) bmax = tile[i] > bmax ? tile[i] : bmax; /* elem.rs:15: fold(NEG_INFINITY, f32::max) */
      float *vkq = rows + (size_t)j * stride;
      if (bmax > m[j]) {
        if (m[j] != -INFINITY) {
          const float corr = orc_fast_exp(m[j] - bmax);
          for (int d = 0; d < hd; ++d) vkq[d] *= corr; /* scale_acc */
          s[j] *= corr;
        }
        m[j] = bmax;
      }
      float local = 0.0f; /* elem.rs:26-33 simd_softmax_row_f32, portable body */
      for (int i = 0; i < bn; ++i) { tile[i] = orc_fast_exp(tile[i] - m[j]); local += tile[i]; }
      s[j] += local;
    }
    /* f32: pv_tile returns false -> per position, per row: vkq += v_row * p   (elem.rs:391-396 mad_f32) */
    for (int p = bs; p < be; ++p) {
      const float *v_row = v + ((size_t)p * KVH + k_head) * hd;
      for (int j = 0; j < group; ++j) {
        const float pr = s_tile[j * TILE + (p - bs)];
        float *vkq = rows + (size_t)j * stride;
        for (int d = 0; d < hd; ++d) vkq[d] += v_row[d] * pr;
      }
    }
  }
  for (int j = 0; j < group; ++j) { rows[(size_t)j * stride + hd] = m[j]; rows[(size_t)j * stride + hd + 1] = s[j]; }
}

/* single_q.rs:57-157 run_barrier for one sequence: q [H][hd], k / v [kv_len][KVH][hd] (already rounded to the cache dtype), out [H][hd].
 * n_kv_chunks: the reference derives it from its thread pool (UNITS_PER_THREAD * threads / groups, clamped to [1, ceil(kv_len / 256)]); 1 = one pass. */
void orc_attention_single_q_cpu(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, int hd, float scale,
                                int n_kv_chunks) {
  const int rk2 = H / KVH, stride = hd + 2;
  int group = 1;
  for (int g = 1; g <= (rk2 < 8 ? rk2 : 8); ++g)
    if (rk2 % g == 0) group = g;
  const int n_groups = H / group;
  const int max_chunks = (kv_len + 255) / 256 > 1 ? (kv_len + 255) / 256 : 1;
  if (n_kv_chunks < 1) n_kv_chunks = 1;
  if (n_kv_chunks > max_chunks) n_kv_chunks = max_chunks;
  const int kv_chunk = (kv_len + n_kv_chunks - 1) / n_kv_chunks > 1 ? (kv_len + n_kv_chunks - 1) / n_kv_chunks : 1;
  float *partials = calloc((size_t)H * n_kv_chunks * stride, sizeof(float));
#pragma omp parallel for schedule(static)
  for (int unit = 0; unit < n_groups * n_kv_chunks; ++unit) {
    const int gi = unit / n_kv_chunks, ci = unit % n_kv_chunks;
    const int kv_start = ci * kv_chunk, kv_end = n_kv_chunks == 1 ? kv_len : (kv_len < kv_start + kv_chunk ? kv_len : kv_start + kv_chunk);
    if (kv_start >= kv_end) { /* empty tail chunk: vkq = 0, m = -inf, s = 0 */
      for (int j = 0; j < group; ++j) partials[((size_t)ci * H + gi * group + j) * stride + hd] = -INFINITY;
      continue;
    }
    compute_group_range(q, k, v, H, KVH, hd, scale, gi * group, group, kv_start, kv_end, partials + ((size_t)ci * H + gi * group) * stride);
  }
  for (int row = 0; row < H; ++row) {
    float m_all = -INFINITY;
    for (int c = 0; c < n_kv_chunks; ++c) { const float mc = partials[((size_t)c * H + row) * stride + hd]; m_all = mc > m_all ? mc : m_all; }
    float *o = out + (size_t)row * hd;
    if (n_kv_chunks == 1) {
      const float *p = partials + (size_t)row * stride;
      const float inv_s = 1.0f / p[hd + 1];
      for (int d = 0; d < hd; ++d) o[d] = p[d] * inv_s;
      continue;
    }
    float s_all = 0.0f;
    for (int d = 0; d < hd; ++d) o[d] = 0.0f;
    for (int c = 0; c < n_kv_chunks; ++c) {
      const float *p = partials + ((size_t)c * H + row) * stride;
      const float m_c = p[hd], s_c = p[hd + 1];
      if (s_c == 0.0f || m_c == -INFINITY) continue;
      const float w = expf(m_c - m_all); /* f32::exp (libm) in the merge, fast_exp only in the hot loop */
      s_all += s_c * w;
      for (int d = 0; d < hd; ++d) o[d] += p[d] * w;
    }
    const float inv_s = 1.0f / s_all;
    for (int d = 0; d < hd; ++d) o[d] = o[d] * inv_s;
  }
  free(partials);
}

/* candle_nn::ops::rms_norm, CPU f32 kernel */
void orc_rms_norm_candle(const float *x, const float *w, float *out, int rows, int d, float eps) {
  for (int r = 0; r < rows; ++r) {
    const float *xr = x + (size_t)r * d;
    float sum2 = 0.0f;
    for (int i = 0; i < d; ++i) sum2 += xr[i] * xr[i];
    const float m = sqrtf(sum2 / (float)d + eps);
    for (int i = 0; i < d; ++i) out[(size_t)r * d + i] = xr[i] / m * w[i];
  }
}

/* full.rs:118-196 run (f32: USE_BARRIER_POOL, logit_softcap == 0 -> compute_tiled_qblock) + full.rs:198-421 compute_tiled_qblock for the prompt of
 * ONE sequence: q [q_len][H][hd], k / v [kv_len][KVH][hd] (already rounded to the cache dtype), out [q_len][H][hd].
 * The mask is the causal (+ sliding window) 0 / -inf matrix of attention/mod.rs:74-103 eager_attention_mask, which full.rs:80-117 binary_mask_range
 * classifies as Binary{start, end} for every row (mask.rs:13-31 row_with_id: one contiguous kv row per query) -- so no mask value is ever added to a
 * score and each row only scores its live range [start, end): start = query_pos - window + 1 (when query_pos >= window), end = query_pos + 1 with
 * query_pos = kv_len - q_len + row.  Q_BLOCK = 8 query rows share the K / V stream, KV_BLOCK = 128 positions per tile, ONE online-softmax correction
 * per tile (fast_exp), P.V by mad over the whole tile span with dead slots zeroed (p == 0 skipped).  f32: EXPAND_SCORE = false (score_rows:
 * dot4 / dot -> the portable dot body, see the header note on SIMD orders), pv_tile = false. */
void orc_attention_full_cpu(const float *q, const float *k, const float *v, float *out, int q_len, int kv_len, int H, int KVH, int hd, float scale,
                            int window) {
  enum { Q_BLOCK = 8, KV_BLOCK = 128 };
  const int rk2 = H / KVH, n_q_blocks = (q_len + Q_BLOCK - 1) / Q_BLOCK, prefix = kv_len > q_len ? kv_len - q_len : 0;
#pragma omp parallel for schedule(dynamic)
  for (int unit = 0; unit < H * n_q_blocks; ++unit) {
    const int q_block_idx = unit % n_q_blocks, h_i = unit / n_q_blocks, k_head = h_i / rk2;
    const int q_start = q_block_idx * Q_BLOCK, q_end = q_len < q_start + Q_BLOCK ? q_len : q_start + Q_BLOCK, nq = q_end - q_start;
    int row_start[Q_BLOCK], row_end[Q_BLOCK], kv_lo = kv_len, kv_hi = 0;
    for (int j = 0; j < nq; ++j) {
      const int qp = prefix + q_start + j;
      row_start[j] = window > 0 && qp >= window ? qp - window + 1 : 0;
      row_end[j] = qp + 1 < kv_len ? qp + 1 : kv_len;
      kv_lo = row_start[j] < kv_lo ? row_start[j] : kv_lo;
      kv_hi = row_end[j] > kv_hi ? row_end[j] : kv_hi;
    }
    float *acc = calloc((size_t)nq * hd, sizeof(float));
    float m[Q_BLOCK], s[Q_BLOCK], s_tile[Q_BLOCK * KV_BLOCK];
    for (int j = 0; j < Q_BLOCK; ++j) { m[j] = -INFINITY; s[j] = 0.0f; }
    for (int bs = kv_lo; bs < kv_hi; bs += KV_BLOCK) {
      const int be = kv_hi < bs + KV_BLOCK ? kv_hi : bs + KV_BLOCK;
      for (int j = 0; j < nq; ++j) { /* A: score the tile (live range only) */
        const int lo = row_start[j] > bs ? row_start[j] : bs, hi = row_end[j] < be ? row_end[j] : be;
        if (lo >= hi) continue;
        const float *q_row = q + ((size_t)(q_start + j) * H + h_i) * hd;
        for (int p = lo; p < hi; ++p) s_tile[j * KV_BLOCK + p - bs] = dot_f32_portable(q_row, k + ((size_t)p * KVH + k_head) * hd, hd) * scale;
      }
      for (int j = 0; j < nq; ++j) { /* B: one online-softmax correction per tile */
        const int lo = row_start[j] > bs ? row_start[j] : bs, hi = row_end[j] < be ? row_end[j] : be;
        if (lo >= hi) continue;
        float *live = s_tile + j * KV_BLOCK + (lo - bs);
        float bmax = -INFINITY;
        for (int i = 0; i < hi - lo; ++i) bmax = live[i] > bmax ? live[i] : bmax;
        if (bmax > m[j]) {
          if (m[j] != -INFINITY) {
            const float corr = orc_fast_exp(m[j] - bmax);
            for (int d = 0; d < hd; ++d) acc[(size_t)j * hd + d] *= corr;
            s[j] *= corr;
          }
          m[j] = bmax;
        }
        float local = 0.0f;
        for (int i = 0; i < hi - lo; ++i) { live[i] = orc_fast_exp(live[i] - m[j]); local += live[i]; }
        s[j] += local;
      }
      for (int j = 0; j < nq; ++j) { /* dead tile slots = 0 */
        const int lo = row_start[j] > bs ? row_start[j] : bs, hi = row_end[j] < be ? row_end[j] : be;
        float *row = s_tile + j * KV_BLOCK;
        if (lo >= hi) { for (int i = 0; i < be - bs; ++i) row[i] = 0.0f; continue; }
        for (int i = 0; i < lo - bs; ++i) row[i] = 0.0f;
        for (int i = hi - bs; i < be - bs; ++i) row[i] = 0.0f;
      }
      for (int p = bs; p < be; ++p) { /* C: P.V, each v row shared by the q block (groups of 4 rows do not change any row's order) */
        const float *v_row = v + ((size_t)p * KVH + k_head) * hd;
        for (int j = 0; j < nq; ++j) {
          const float pr = s_tile[j * KV_BLOCK + (p - bs)];
          if (pr != 0.0f)
            for (int d = 0; d < hd; ++d) acc[(size_t)j * hd + d] += v_row[d] * pr;
        }
      }
    }
    for (int j = 0; j < nq; ++j) {
      const float inv_s = 1.0f / s[j];
      float *o = out + ((size_t)(q_start + j) * H + h_i) * hd;
      for (int d = 0; d < hd; ++d) o[d] = acc[(size_t)j * hd + d] * inv_s;
    }
    free(acc);
  }
}


/* ================================================================== engine order */
/* the 64-lane all-reduce of dec_core2.cuh wave_sum_all: v += v[i^1]; v += v[i^2]; v += v[mirror within 8]; v += v[mirror within 16];
 * result = (lane0 + lane16) + (lane32 + lane48) */
static float wave_sum_all_64(const float *in) {
  float a[64], b[64];
  memcpy(a, in, sizeof a);
  for (int i = 0; i < 64; ++i) b[i] = a[i] + a[i ^ 1];
  for (int i = 0; i < 64; ++i) a[i] = b[i] + b[i ^ 2];
  for (int i = 0; i < 64; ++i) b[i] = a[i] + a[(i & ~7) | (7 - (i & 7))];
  for (int i = 0; i < 64; ++i) a[i] = b[i] + b[(i & ~15) | (15 - (i & 15))];
  return (a[0] + a[16]) + (a[32] + a[48]);
}

/* RmsNorm of the engine's GEMV prologue (dec_core2.cuh act_finish / ActStager, 512-thread workgroups): thread t sums the squares of its float4 pieces
 * e = 4 t + 2048 j (j ascending, x, y, z, w in order, fma), wave sums by wave_sum_all, the 8 wave sums as ((0+1)+(2+3))+((4+5)+(6+7));
 * m = sqrt(tot / d + eps); y_i = x_i / m * w_i  (candle's expression; the device computes the correctly rounded quotient with one
 * reciprocal and two fma per element, checked against `/` in tests/test_dec_engine.py). */
void orc_rms_norm_engine(const float *x, const float *w, float *out, int rows, int d, float eps) {
  for (int r = 0; r < rows; ++r) {
    const float *xr = x + (size_t)r * d;
    float part[512];
    for (int t = 0; t < 512; ++t) {
      float ss = 0.0f;
      for (int e = t * 4; e < d; e += 2048)
        for (int c = 0; c < 4; ++c) ss = fmaf(xr[e + c], xr[e + c], ss);
      part[t] = ss;
    }
    float ws[8];
    for (int wv = 0; wv < 8; ++wv) ws[wv] = wave_sum_all_64(part + wv * 64);
    const float tot = ((ws[0] + ws[1]) + (ws[2] + ws[3])) + ((ws[4] + ws[5]) + (ws[6] + ws[7]));
    const float m = sqrtf(tot / (float)d + eps);
    for (int i = 0; i < d; ++i) out[(size_t)r * d + i] = xr[i] / m * w[i];
  }
}

/* SiLU of the engine's gate / up epilogue: x / (1 + fast_exp(-x)) -- the reference's own Cephes exp (elem.rs:417-433) in place of libm's, so that
 * host and device evaluate the same expression bit for bit */
float orc_silu_engine(float x) { return x / (1.0f + orc_fast_exp(-x)); }
void orc_fused_glu_engine(const float *a, const float *b, float *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_silu_engine(a[i]) * b[i];
}

/* Decode attention of the engine (mistral.rs_amd/csrc/dec_attn2.cuh) for one sequence; head size 128, 32-token pages.
 * A split = bpw consecutive 32-token blocks, processed in order with an online softmax (state m, l, o[128] per query head):
 *   score(t)  = (chain over dims 0..63) + (chain over dims 64..127), each an fma chain in dim order from 0;  * scale
 *   mn = max(m, max_t score);  p_t = fast_exp(score_t - mn)  (0 beyond the context);  ps = the 32-value tree below
 *   alpha = fast_exp(m - mn);  l = l * alpha + ps;  o[d] = o[d] * alpha;  then o[d] = fma(p_t, v[t][d], o[d]) for t = 0..31 in order
 * merge (the order of single_q.rs run_barrier): m_all = max_j m_j;  s += l_j * w_j, acc[d] += o_j[d] * w_j with w_j = fast_exp(m_j - m_all),
 * j ascending, separate multiply and add;  out[d] = acc[d] * (1 / s). */
static float tree32(const float *in) {
  float a[32], b[32];
  memcpy(a, in, sizeof a);
  for (int i = 0; i < 32; ++i) b[i] = a[i] + a[i ^ 1];
  for (int i = 0; i < 32; ++i) a[i] = b[i] + b[i ^ 2];
  for (int i = 0; i < 32; ++i) b[i] = a[i] + a[(i & ~7) | (7 - (i & 7))];
  for (int i = 0; i < 32; ++i) a[i] = b[i] + b[(i & ~15) | (15 - (i & 15))];
  return a[0] + a[16];
}
void orc_attention_engine_w(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, float scale, int bpw, int window);
void orc_attention_engine(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, float scale, int bpw) {
  orc_attention_engine_w(q, k, v, out, kv_len, H, KVH, scale, bpw, 0);
}
/* window > 0: positions below kv_len - window are masked like positions past the context (score -FLT_MAX, p = 0); a split that lies entirely before the
 * window keeps (m, l, o) = (-FLT_MAX, 0, 0) -- what the masked pass computes, and what the kernel publishes without reading K / V */
void orc_attention_engine_w(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, float scale, int bpw, int window) {
  enum { HD = 128, BS = 32 };
  const int lo = window > 0 && kv_len > window ? kv_len - window : 0;
  const int G = H / KVH, nblk = (kv_len + BS - 1) / BS;
  if (bpw < 1) bpw = 1;
  const int ns = (nblk + bpw - 1) / bpw;
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h) {
    const int kvh = h / G;
    const float *qh = q + (size_t)h * HD;
    float *pm = malloc(sizeof(float) * ns), *pl = malloc(sizeof(float) * ns), *po = malloc(sizeof(float) * ns * HD);
    for (int sp = 0; sp < ns; ++sp) {
      float m = -FLT_MAX, l = 0.0f, o[HD];
      for (int d = 0; d < HD; ++d) o[d] = 0.0f;
      const int b1 = (sp + 1) * bpw < nblk ? (sp + 1) * bpw : nblk;
      for (int b = sp * bpw; b < b1; ++b) {
        float sc[BS], p[BS], mx = -FLT_MAX;
        for (int t = 0; t < BS; ++t) {
          const int pos = b * BS + t;
          if (pos < kv_len && pos >= lo) {
            const float *kr = k + ((size_t)pos * KVH + kvh) * HD;
            float s0 = 0.0f, s1 = 0.0f;
            for (int d = 0; d < 64; ++d) s0 = fmaf(qh[d], kr[d], s0);
            for (int d = 64; d < 128; ++d) s1 = fmaf(qh[d], kr[d], s1);
            sc[t] = (s0 + s1) * scale;
          } else {
            sc[t] = -FLT_MAX;
          }
          mx = fmaxf(mx, sc[t]);
        }
        const float mn = fmaxf(m, mx);
        for (int t = 0; t < BS; ++t) p[t] = b * BS + t < kv_len && b * BS + t >= lo ? orc_fast_exp(sc[t] - mn) : 0.0f;
        const float ps = tree32(p);
        const float alpha = orc_fast_exp(m - mn);
        l = l * alpha + ps;
        m = mn;
        for (int d = 0; d < HD; ++d) {
          float acc = o[d] * alpha;
          for (int t = 0; t < BS; ++t) {
            const int pos = b * BS + t;
            acc = fmaf(p[t], pos < kv_len ? v[((size_t)pos * KVH + kvh) * HD + d] : 0.0f, acc);
          }
          o[d] = acc;
        }
      }
      pm[sp] = m; pl[sp] = l;
      memcpy(po + (size_t)sp * HD, o, sizeof o);
    }
    float m_all = -FLT_MAX, s_all = 0.0f, acc[HD];
    for (int j = 0; j < ns; ++j) m_all = fmaxf(m_all, pm[j]);
    for (int d = 0; d < HD; ++d) acc[d] = 0.0f;
    for (int j = 0; j < ns; ++j) {
      const float w = orc_fast_exp(pm[j] - m_all);
      s_all = s_all + pl[j] * w;
      for (int d = 0; d < HD; ++d) acc[d] = acc[d] + po[(size_t)j * HD + d] * w;
    }
    const float inv = 1.0f / s_all;
    for (int d = 0; d < HD; ++d) out[(size_t)h * HD + d] = acc[d] * inv;
    free(pm); free(pl); free(po);
  }
}

/* The device's quotient x / m: y = 1 / m correctly rounded, q0 = x y, r = fma(-m, q0, x), q = fma(r, y, q0) (dec_core2.cuh div_by).
 * Returns the number of (x, m) pairs for which it differs from the IEEE quotient: tests/test_oracle.py holds it to 0. */
int64_t orc_div_by_mismatches(const float *x, const float *m, int64_t n) {
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float y = 1.0f / m[i], q0 = x[i] * y, r = fmaf(-m[i], q0, x[i]), q = fmaf(r, y, q0);
    if (q != x[i] / m[i]) ++bad;
  }
  return bad;
}

/* ================================================================== the engine's GEMV order "ORD-U" (mistral.rs_amd/csrc/dec_core2.cuh, ext_gemm_qi.hip)
 * Same integers and the same f32 products as the reference CPU matvec (activation row -> Q8_K / Q8_0, integer dots per sub-block, d_w * d_x scaling;
 * orc_matmul_cpu / orc_gemv_cpu_fast).  The f32 combination is ONE order shared by the batch-1 GEMV, the batched GEMV and the prompt GEMM on the matrix cores:
 *     T_sb  one f32 term per 256-value superblock from its EXACT integer sums:
 *             Q4_K / Q5_K:  fma(d_w d_x, (float)isum, -((dmin_w d_x) (float)msum))      isum = sum_j sc_j <q_j, a_j>,  msum = sum_j m_j sum(a_j)
 *             Q6_K:         (d_w d_x) (float)isum                                       isum = sum_r sc_r <q_r - 32, a_r>
 *             Q8_0:         t = ((float)isum_0 dw_0) dx_0;  t = t + ((float)isum_b dw_b) dx_b  for the 8 blocks of the 256 values in order
 *     the row's S superblocks are cut into 4 runs of Cs = ceil(S / 4); c_p = the T of run p added left to right (an empty run is +0)
 *     row = ((c_0 + c_1) + c_2) + c_3
 * Written from the format definitions, not from the kernel source. */
static void k4sm(int j, const uint8_t *p, int *sc, int *mn) {
  if (j < 4) { *sc = p[j] & 63; *mn = p[j + 4] & 63; }
  else { *sc = (p[j + 4] & 0xF) | ((p[j - 4] >> 6) << 4); *mn = (p[j + 4] >> 4) | ((p[j] >> 6) << 4); }
}
static inline float h2f_(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return orc_fp16_to_fp32(v); }

/* exact integer sums of one superblock against one Q8_K activation block (292 B: d f32, 256 int8, 16 int16 sums) */
static float term_q45K(int type, const uint8_t *b, const uint8_t *yb) {
  float yd; memcpy(&yd, yb, 4);
  const int8_t *q8 = (const int8_t *)(yb + 4);
  const uint8_t *qh = b + 16, *qs = b + (type == ORC_Q4_K ? 16 : 48);
  int32_t isum = 0, msum = 0;
  for (int j = 0; j < 8; ++j) {  /* sub-block j: 32 values at 32 j */
    int sc, mn; k4sm(j, b + 4, &sc, &mn);
    int32_t dj = 0, bs = 0;
    for (int l = 0; l < 32; ++l) {
      int q = (j & 1) ? (qs[(j >> 1) * 32 + l] >> 4) : (qs[(j >> 1) * 32 + l] & 0xF);
      if (type == ORC_Q5_K) q |= ((qh[l] >> j) & 1) << 4;
      dj += q * q8[32 * j + l];
      bs += q8[32 * j + l];
    }
    isum += sc * dj;
    msum += mn * bs;
  }
  const float d = h2f_(b), dmin = h2f_(b + 2);
  return fmaf(d * yd, (float)isum, -((dmin * yd) * (float)msum));
}
static float term_q6K(const uint8_t *b, const uint8_t *yb) {
  float yd; memcpy(&yd, yb, 4);
  const int8_t *q8 = (const int8_t *)(yb + 4);
  const uint8_t *ql = b, *qhh = b + 128;
  const int8_t *sc = (const int8_t *)(b + 192);
  int32_t isum = 0;
  for (int r = 0; r < 16; ++r) {
    int32_t dsum = 0;
    for (int i = 0; i < 16; ++i) {
      const int e = r * 16 + i, hh = e / 128, pos = e % 32, qt = (e % 128) / 32, ii = hh * 64 + pos + (qt % 2) * 32;
      const int lo = qt < 2 ? (ql[ii] & 15) : (ql[ii] >> 4), hi = (qhh[hh * 32 + pos] >> (qt * 2)) & 3;
      dsum += ((lo | (hi << 4)) - 32) * q8[e];
    }
    isum += sc[r] * dsum;
  }
  return (h2f_(b + 208) * yd) * (float)isum;
}
static float term_q8_0(const uint8_t *w /* 8 blocks of 34 B */, const uint8_t *y /* 8 Q8_0 blocks of 34 B */) {
  float t = 0.0f;
  for (int blk = 0; blk < 8; ++blk) {
    const int8_t *a = (const int8_t *)(w + (size_t)blk * 34 + 2), *b = (const int8_t *)(y + (size_t)blk * 34 + 2);
    int32_t s = 0;
    for (int l = 0; l < 32; ++l) s += a[l] * b[l];
    const float p = (float)s * h2f_(w + (size_t)blk * 34) * h2f_(y + (size_t)blk * 34);
    t = blk == 0 ? p : t + p;
  }
  return t;
}
/* combine the S superblock terms of a row in ORD-U order */
float orc_ordu_combine(const float *T, int S) {
  const int Cs = (S + 3) / 4;
  float c[4];
  for (int p = 0; p < 4; ++p) {
    const int a = p * Cs, e = a + Cs < S ? a + Cs : S;
    float v = 0.0f;
    for (int sb = a; sb < e; ++sb) v = sb == a ? T[sb] : v + T[sb];
    c[p] = v;
  }
  return ((c[0] + c[1]) + c[2]) + c[3];
}
static float row_engine(int type, const uint8_t *w, int S, const uint8_t *y) {
  float T[1024];
  for (int sb = 0; sb < S; ++sb)
    T[sb] = type == ORC_Q6_K ? term_q6K(w + (size_t)sb * 210, y + (size_t)sb * 292)
          : type == ORC_Q8_0 ? term_q8_0(w + (size_t)sb * 272, y + (size_t)sb * 272)
          : term_q45K(type, w + (size_t)sb * (type == ORC_Q4_K ? 144 : 176), y + (size_t)sb * 292);
  return orc_ordu_combine(T, S);
}

/* out[N] = W[N,K] . x[K] in the engine's order.  Returns 0, or -1 for a type / shape the engine does not take (K must be a multiple of 256). */
int orc_gemv_engine(int type, const void *W, int N, int K, const float *x, float *out) {
  if ((type != ORC_Q4_K && type != ORC_Q5_K && type != ORC_Q6_K && type != ORC_Q8_0) || K % 256 || K / 256 > 1024) return -1;
  const int kq = type != ORC_Q8_0;
  const size_t row_bytes = (size_t)(K / orc_block_size(type)) * orc_type_size(type);
  uint8_t *y = malloc(kq ? (size_t)(K / 256) * 292 : (size_t)(K / 32) * 34);
  if (kq) orc_quantize_q8_K(x, y, K); else orc_quantize_row(ORC_Q8_0, x, y, K);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) out[n] = row_engine(type, (const uint8_t *)W + (size_t)n * row_bytes, K / 256, y);
  free(y);
  return 0;
}

/* trunc(x + copysign(0.49999997f, x)) (the device's round-half-away, dec_core2.cuh round_away / common.cuh fast_exp_ref) against roundf over EVERY
 * float with |x| <= limit: returns the number of mismatches (tests/test_oracle.py holds it to 0 for limit = 129) */
int64_t orc_round_trick_mismatches(float limit) {
  uint32_t top;
  memcpy(&top, &limit, 4);
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (uint32_t b = 0; b <= top; ++b) {
    float x;
    memcpy(&x, &b, 4);
    if (truncf(x + copysignf(0.49999997f, x)) != roundf(x)) ++bad;
    if (truncf(-x + copysignf(0.49999997f, -x)) != roundf(-x)) ++bad;
  }
  return bad;
}
