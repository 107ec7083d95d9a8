/*
 * oracle/llama_oracle.c -- TEST INFRASTRUCTURE ONLY (see ggml_oracle.h).
 *
 * Throughput-oriented restatement of the reference CPU matvec (candle QMatMul::forward, b = 1):
 * activation row -> Q8_K (K-quants) / Q8_0 (Q8_0), integer dot products, f32 accumulation, rows spread
 * over OpenMP threads.  Same arithmetic as orc_matmul_cpu (ggml_oracle.c) -- the integer parts are
 * identical, the f32 accumulation is per 32-/16-group instead of ggml's 8-lane order -- but written as
 * plain loops gcc vectorises (-O3 -march=native), so that bench.py's cpu_baseline ("port") is a fair
 * stand-in for the reference's SIMD CPU path, which cannot be built here (no Rust toolchain).
 */
#include "ggml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float h2f(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return orc_fp16_to_fp32(v); }

static float dot_q4_K_fast(const uint8_t *w, int nblk, const uint8_t *y) {
  float sumf = 0;
  for (int ib = 0; ib < nblk; ++ib, w += 144, y += 292) {
    float yd; memcpy(&yd, y, 4);
    const int8_t *q8 = (const int8_t *)(y + 4);
    int16_t bsums[16]; memcpy(bsums, y + 260, 32);
    const uint8_t *sc12 = w + 4, *qs = w + 16;
    int sc[8], mn[8];
    for (int j = 0; j < 4; ++j) { sc[j] = sc12[j] & 63; mn[j] = sc12[j + 4] & 63; }
    for (int j = 4; j < 8; ++j) { sc[j] = (sc12[j + 4] & 0xF) | ((sc12[j - 4] >> 6) << 4); mn[j] = (sc12[j + 4] >> 4) | ((sc12[j] >> 6) << 4); }
    int summ = 0;
    for (int j = 0; j < 8; ++j) summ += mn[j] * (bsums[2 * j] + bsums[2 * j + 1]);
    int sumi = 0;
    for (int c = 0; c < 4; ++c) {
      int s0 = 0, s1 = 0;
      for (int l = 0; l < 32; ++l) { s0 += q8[c * 64 + l] * (qs[c * 32 + l] & 0xF); s1 += q8[c * 64 + 32 + l] * (qs[c * 32 + l] >> 4); }
      sumi += sc[2 * c] * s0 + sc[2 * c + 1] * s1;
    }
    sumf += h2f(w) * yd * (float)sumi - h2f(w + 2) * yd * (float)summ;
  }
  return sumf;
}

static float dot_q5_K_fast(const uint8_t *w, int nblk, const uint8_t *y) {
  float sumf = 0;
  for (int ib = 0; ib < nblk; ++ib, w += 176, y += 292) {
    float yd; memcpy(&yd, y, 4);
    const int8_t *q8 = (const int8_t *)(y + 4);
    int16_t bsums[16]; memcpy(bsums, y + 260, 32);
    const uint8_t *sc12 = w + 4, *qh = w + 16, *qs = w + 48;
    int sc[8], mn[8];
    for (int j = 0; j < 4; ++j) { sc[j] = sc12[j] & 63; mn[j] = sc12[j + 4] & 63; }
    for (int j = 4; j < 8; ++j) { sc[j] = (sc12[j + 4] & 0xF) | ((sc12[j - 4] >> 6) << 4); mn[j] = (sc12[j + 4] >> 4) | ((sc12[j] >> 6) << 4); }
    int summ = 0;
    for (int j = 0; j < 8; ++j) summ += mn[j] * (bsums[2 * j] + bsums[2 * j + 1]);
    int sumi = 0;
    for (int c = 0; c < 4; ++c) {
      int s0 = 0, s1 = 0;
      for (int l = 0; l < 32; ++l) {
        const int lo = (qs[c * 32 + l] & 0xF) | (((qh[l] >> (2 * c)) & 1) << 4);
        const int hi = (qs[c * 32 + l] >> 4) | (((qh[l] >> (2 * c + 1)) & 1) << 4);
        s0 += q8[c * 64 + l] * lo; s1 += q8[c * 64 + 32 + l] * hi;
      }
      sumi += sc[2 * c] * s0 + sc[2 * c + 1] * s1;
    }
    sumf += h2f(w) * yd * (float)sumi - h2f(w + 2) * yd * (float)summ;
  }
  return sumf;
}

static float dot_q6_K_fast(const uint8_t *w, int nblk, const uint8_t *y) {
  float sumf = 0;
  for (int ib = 0; ib < nblk; ++ib, w += 210, y += 292) {
    float yd; memcpy(&yd, y, 4);
    const int8_t *q8 = (const int8_t *)(y + 4);
    const uint8_t *ql = w, *qh = w + 128;
    const int8_t *sc = (const int8_t *)(w + 192);
    int sumi = 0;
    for (int h = 0; h < 2; ++h) {
      int s[8] = {0};
      for (int l = 0; l < 32; ++l) {
        const int q1 = ((ql[h * 64 + l] & 0xF) | (((qh[h * 32 + l] >> 0) & 3) << 4)) - 32;
        const int q2 = ((ql[h * 64 + 32 + l] & 0xF) | (((qh[h * 32 + l] >> 2) & 3) << 4)) - 32;
        const int q3 = ((ql[h * 64 + l] >> 4) | (((qh[h * 32 + l] >> 4) & 3) << 4)) - 32;
        const int q4 = ((ql[h * 64 + 32 + l] >> 4) | (((qh[h * 32 + l] >> 6) & 3) << 4)) - 32;
        const int g = l / 16;
        s[0 + g] += q8[h * 128 + l] * q1;
        s[2 + g] += q8[h * 128 + 32 + l] * q2;
        s[4 + g] += q8[h * 128 + 64 + l] * q3;
        s[6 + g] += q8[h * 128 + 96 + l] * q4;
      }
      for (int g = 0; g < 8; ++g) sumi += sc[h * 8 + g] * s[g];
    }
    sumf += h2f(w + 208) * yd * (float)sumi;
  }
  return sumf;
}

static float dot_q8_0_fast(const uint8_t *w, int nblk, const uint8_t *y /* Q8_0 34 B */) {
  float sumf = 0;
  for (int ib = 0; ib < nblk; ++ib, w += 34, y += 34) {
    const int8_t *a = (const int8_t *)(w + 2), *b = (const int8_t *)(y + 2);
    int s = 0;
    for (int l = 0; l < 32; ++l) s += a[l] * b[l];
    sumf += (float)s * h2f(w) * h2f(y);
  }
  return sumf;
}

/* out[N] = W[N,K] . x[K]   (b = 1).  Returns 0, or -1 when the type has no fast path. */
int orc_gemv_cpu_fast(int type, const void *W, int N, int K, const float *x, float *out) {
  if (type != ORC_Q4_K && type != ORC_Q5_K && type != ORC_Q6_K && type != ORC_Q8_0) return -1;
  const int kq = type != ORC_Q8_0;
  const size_t row_bytes = (size_t)(K / orc_block_size(type)) * orc_type_size(type);
  uint8_t *y = malloc(kq ? (size_t)(K / 256) * 292 : (size_t)(K / 32) * 34);
  if (kq) orc_quantize_q8_K(x, y, K); else orc_quantize_row(ORC_Q8_0, x, y, K);
  const int nblk = K / orc_block_size(type);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const uint8_t *w = (const uint8_t *)W + (size_t)n * row_bytes;
    out[n] = type == ORC_Q4_K ? dot_q4_K_fast(w, nblk, y) : type == ORC_Q5_K ? dot_q5_K_fast(w, nblk, y)
           : type == ORC_Q6_K ? dot_q6_K_fast(w, nblk, y) : dot_q8_0_fast(w, nblk, y);
  }
  free(y);
  return 0;
}
