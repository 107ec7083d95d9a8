/*
 * oracle/ggml_oracle.c -- TEST INFRASTRUCTURE ONLY (see ggml_oracle.h).
 *
 * Plain-C restatement of the GGML block arithmetic used by the reference hot path.
 * Sources followed (no code copied; the layouts are the public GGML definitions):
 *   block structs / sizes ........ mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:134-226,
 *                                  kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu:44-110
 *   per-element decode ........... marlin_gguf_affine_repack.cu:141-278 (get_quant,
 *                                  get_scale_min_k4, get_affine_params)
 *   Q8_1 activation quantizer .... mmvq_gguf.cu:1220-1250 (GPU semantics)
 *   MMVQ dot products ............ mmvq_gguf.cu:240-700 (vec_dot_*_q8_1)
 *   CPU dot products / Q8_K ...... candle k_quants (git dep, NOT in tree) == public GGML
 *                                  generic `ggml_vec_dot_*_q8_K` -- "parity unpinned"
 *   weight quantizers ............ public GGML `quantize_row_*_ref` (make_qkx2_quants,
 *                                  make_qx_quants); only used to synthesise weights
 */
#include "ggml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QK_K 256

/* ------------------------------------------------------------------ fp16 / bf16 */
float orc_fp16_to_fp32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal */
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t orc_fp32_to_fp16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xff;
  uint32_t man = x & 0x7fffffu;
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0));
  int e = (int)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign; /* underflow to zero */
    man |= 0x800000u;
    int shift = 14 - e; /* 14..24 */
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1))) half++;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++; /* may carry into exp: ok */
  return (uint16_t)(sign | half);
}

float orc_bf16_to_fp32(uint16_t h) {
  uint32_t bits = (uint32_t)h << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t orc_fp32_to_bf16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40); /* quiet NaN */
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7fffu + lsb;
  return (uint16_t)(x >> 16);
}

/* ------------------------------------------------------------------ type table */
int orc_block_size(int t) {
  switch (t) {
  case ORC_F32: case ORC_F16: case ORC_BF16: return 1;
  case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_Q8_0: case ORC_Q8_1: return 32;
  case ORC_Q2_K: case ORC_Q3_K: case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: case ORC_Q8_K: return 256;
  default: return 0;
  }
}
int orc_type_size(int t) {
  switch (t) {
  case ORC_F32: return 4; case ORC_F16: case ORC_BF16: return 2;
  case ORC_Q4_0: return 18; case ORC_Q4_1: return 20; case ORC_Q5_0: return 22; case ORC_Q5_1: return 24;
  case ORC_Q8_0: return 34; case ORC_Q8_1: return 36;
  case ORC_Q2_K: return 84; case ORC_Q3_K: return 110; case ORC_Q4_K: return 144;
  case ORC_Q5_K: return 176; case ORC_Q6_K: return 210; case ORC_Q8_K: return 292;
  default: return 0;
  }
}

static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void st16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
static inline float ldf32(const uint8_t *p) { float v; memcpy(&v, p, 4); return v; }

/* 6-bit (scale, min) pair j of a K-quant 12-byte scale field */
static inline void k4_scale_min(int j, const uint8_t *q, uint8_t *sc, uint8_t *m) {
  if (j < 4) {
    *sc = q[j] & 63;
    *m = q[j + 4] & 63;
  } else {
    *sc = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
    *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4));
  }
}

/* Q3_K: 16 signed 6-bit scales (already minus 32) */
static void q3k_scales(const uint8_t *s12, int8_t *out16) {
  for (int j = 0; j < 16; ++j) {
    int lo = (j < 8) ? (s12[j] & 0xF) : (s12[j - 8] >> 4);
    int hi = (s12[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
    out16[j] = (int8_t)((lo | (hi << 4)) - 32);
  }
}

/* ------------------------------------------------------------------ integer views
 * Every format decodes as  w = scale_g * q - offset_g  per group; these helpers expose the
 * integer q (0..2^bits-1, or signed for Q8_0) so dequantisation and the integer dot products
 * share one decode. */
static void block_ints(int type, const uint8_t *b, int *q /*[blk]*/) {
  switch (type) {
  case ORC_Q4_0: case ORC_Q4_1: {
    const uint8_t *qs = b + (type == ORC_Q4_0 ? 2 : 4);
    for (int j = 0; j < 16; ++j) { q[j] = qs[j] & 0xF; q[j + 16] = qs[j] >> 4; }
  } break;
  case ORC_Q5_0: case ORC_Q5_1: {
    const uint8_t *qhp = b + (type == ORC_Q5_0 ? 2 : 4);
    const uint8_t *qs = qhp + 4;
    uint32_t qh; memcpy(&qh, qhp, 4);
    for (int j = 0; j < 16; ++j) {
      q[j] = (qs[j] & 0xF) | (((qh >> j) & 1) << 4);
      q[j + 16] = (qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
    }
  } break;
  case ORC_Q8_0: { const int8_t *qs = (const int8_t *)(b + 2); for (int j = 0; j < 32; ++j) q[j] = qs[j]; } break;
  case ORC_Q8_1: { const int8_t *qs = (const int8_t *)(b + 4); for (int j = 0; j < 32; ++j) q[j] = qs[j]; } break;
  case ORC_Q2_K: {
    const uint8_t *qs = b + 16;
    for (int e = 0; e < 256; ++e) q[e] = (qs[(e / 128) * 32 + (e % 32)] >> (((e % 128) / 32) * 2)) & 3;
  } break;
  case ORC_Q3_K: {
    const uint8_t *hm = b, *qs = b + 32;
    for (int e = 0; e < 256; ++e) {
      int lo = (qs[(e / 128) * 32 + (e % 32)] >> (((e % 128) / 32) * 2)) & 3;
      int hi = (hm[e % 32] >> (e / 32)) & 1;
      q[e] = lo | (hi << 2);
    }
  } break;
  case ORC_Q4_K: {
    const uint8_t *qs = b + 16;
    for (int e = 0; e < 256; ++e) {
      int c = e / 64, p = e % 64;
      uint8_t v = qs[c * 32 + (p % 32)];
      q[e] = p < 32 ? (v & 0xF) : (v >> 4);
    }
  } break;
  case ORC_Q5_K: {
    const uint8_t *qh = b + 16, *qs = b + 48;
    for (int e = 0; e < 256; ++e) {
      int c = e / 64, p = e % 64;
      uint8_t v = qs[c * 32 + (p % 32)];
      int lo = p < 32 ? (v & 0xF) : (v >> 4);
      int hi = (qh[p % 32] >> (c * 2 + p / 32)) & 1;
      q[e] = lo | (hi << 4);
    }
  } break;
  case ORC_Q6_K: {
    const uint8_t *ql = b, *qh = b + 128;
    for (int e = 0; e < 256; ++e) {
      int h = e / 128, pos = e % 32, qt = (e % 128) / 32;
      int i = h * 64 + pos + (qt % 2) * 32;
      int lo = qt < 2 ? (ql[i] & 0xF) : (ql[i] >> 4);
      int hi = (qh[h * 32 + pos] >> (qt * 2)) & 3;
      q[e] = lo | (hi << 4);
    }
  } break;
  case ORC_Q8_K: { const int8_t *qs = (const int8_t *)(b + 4); for (int e = 0; e < 256; ++e) q[e] = qs[e]; } break;
  default: break;
  }
}

/* group affine params: w = scale[g]*q - offset[g]; returns group size */
static int block_affine(int type, const uint8_t *b, float *scale, float *offset) {
  switch (type) {
  case ORC_Q4_0: { float d = orc_fp16_to_fp32(ld16(b)); scale[0] = d; offset[0] = 8.0f * d; return 32; }
  case ORC_Q4_1: { scale[0] = orc_fp16_to_fp32(ld16(b)); offset[0] = -orc_fp16_to_fp32(ld16(b + 2)); return 32; }
  case ORC_Q5_0: { float d = orc_fp16_to_fp32(ld16(b)); scale[0] = d; offset[0] = 16.0f * d; return 32; }
  case ORC_Q5_1: { scale[0] = orc_fp16_to_fp32(ld16(b)); offset[0] = -orc_fp16_to_fp32(ld16(b + 2)); return 32; }
  case ORC_Q8_0: { scale[0] = orc_fp16_to_fp32(ld16(b)); offset[0] = 0.0f; return 32; }
  case ORC_Q8_1: { scale[0] = orc_fp16_to_fp32(ld16(b)); offset[0] = 0.0f; return 32; }
  case ORC_Q2_K: {
    float d = orc_fp16_to_fp32(ld16(b + 80)), dmin = orc_fp16_to_fp32(ld16(b + 82));
    for (int g = 0; g < 16; ++g) { scale[g] = d * (b[g] & 0xF); offset[g] = dmin * (b[g] >> 4); }
    return 16;
  }
  case ORC_Q3_K: {
    float d = orc_fp16_to_fp32(ld16(b + 108));
    int8_t sc[16]; q3k_scales(b + 96, sc);
    for (int g = 0; g < 16; ++g) { scale[g] = d * sc[g]; offset[g] = 4.0f * scale[g]; }
    return 16;
  }
  case ORC_Q4_K: case ORC_Q5_K: {
    float d = orc_fp16_to_fp32(ld16(b)), dmin = orc_fp16_to_fp32(ld16(b + 2));
    for (int g = 0; g < 8; ++g) { uint8_t sc, m; k4_scale_min(g, b + 4, &sc, &m); scale[g] = d * sc; offset[g] = dmin * m; }
    return 32;
  }
  case ORC_Q6_K: {
    float d = orc_fp16_to_fp32(ld16(b + 208));
    const int8_t *sc = (const int8_t *)(b + 192);
    for (int g = 0; g < 16; ++g) { scale[g] = d * sc[g]; offset[g] = 32.0f * scale[g]; }
    return 16;
  }
  case ORC_Q8_K: { scale[0] = ldf32(b); offset[0] = 0.0f; return 256; }
  default: return 0;
  }
}

void orc_dequantize_row(int type, const void *blocks, float *out, int64_t k) {
  if (type == ORC_F32) { memcpy(out, blocks, (size_t)k * 4); return; }
  if (type == ORC_F16) { const uint16_t *p = blocks; for (int64_t i = 0; i < k; ++i) out[i] = orc_fp16_to_fp32(p[i]); return; }
  if (type == ORC_BF16) { const uint16_t *p = blocks; for (int64_t i = 0; i < k; ++i) out[i] = orc_bf16_to_fp32(p[i]); return; }
  const int blk = orc_block_size(type), ts = orc_type_size(type);
  const uint8_t *b = blocks;
  int q[256]; float sc[16], off[16];
  for (int64_t ib = 0; ib < k / blk; ++ib, b += ts) {
    block_ints(type, b, q);
    int gs = block_affine(type, b, sc, off);
    for (int e = 0; e < blk; ++e) out[ib * blk + e] = sc[e / gs] * (float)q[e] - off[e / gs];
  }
}

/* ------------------------------------------------------------------ weight quantizers */
static inline int nearest_int(float f) { return (int)lrintf(f); } /* RNE like ggml's magic-number trick */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static float make_qkx2_quants(int n, int nmax, const float *x, const float *w, uint8_t *L, float *the_min,
                              uint8_t *Laux, float rmin, float rdelta, int nstep, int use_mad) {
  float min = x[0], max = x[0], sum_w = w[0], sum_x = sum_w * x[0];
  for (int i = 1; i < n; ++i) {
    if (x[i] < min) min = x[i];
    if (x[i] > max) max = x[i];
    sum_w += w[i]; sum_x += w[i] * x[i];
  }
  if (min > 0) min = 0;
  if (max == min) { for (int i = 0; i < n; ++i) L[i] = 0; *the_min = -min; return 0.f; }
  float iscale = nmax / (max - min), scale = 1 / iscale, best_mad = 0;
  for (int i = 0; i < n; ++i) {
    int l = nearest_int(iscale * (x[i] - min));
    L[i] = (uint8_t)imax(0, imin(nmax, l));
    float diff = scale * L[i] + min - x[i];
    diff = use_mad ? fabsf(diff) : diff * diff;
    best_mad += w[i] * diff;
  }
  if (nstep < 1) { *the_min = -min; return scale; }
  for (int is = 0; is <= nstep; ++is) {
    iscale = (rmin + rdelta * is + nmax) / (max - min);
    float sum_l = 0, sum_l2 = 0, sum_xl = 0;
    for (int i = 0; i < n; ++i) {
      int l = nearest_int(iscale * (x[i] - min));
      l = imax(0, imin(nmax, l));
      Laux[i] = (uint8_t)l;
      sum_l += w[i] * l; sum_l2 += w[i] * l * l; sum_xl += w[i] * l * x[i];
    }
    float D = sum_w * sum_l2 - sum_l * sum_l;
    if (D > 0) {
      float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D;
      float this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;
      if (this_min > 0) { this_min = 0; this_scale = sum_xl / sum_l2; }
      float mad = 0;
      for (int i = 0; i < n; ++i) {
        float diff = this_scale * Laux[i] + this_min - x[i];
        diff = use_mad ? fabsf(diff) : diff * diff;
        mad += w[i] * diff;
      }
      if (mad < best_mad) {
        memcpy(L, Laux, (size_t)n);
        best_mad = mad; scale = this_scale; min = this_min;
      }
    }
  }
  *the_min = -min;
  return scale;
}

static float make_qx_quants(int n, int nmax, const float *x, int8_t *L) { /* rmse_type 1, no weights */
  float max = 0, amax = 0;
  for (int i = 0; i < n; ++i) { float ax = fabsf(x[i]); if (ax > amax) { amax = ax; max = x[i]; } }
  if (amax < 1e-15f) { for (int i = 0; i < n; ++i) L[i] = 0; return 0.f; }
  float iscale = -nmax / max, sumlx = 0, suml2 = 0;
  for (int i = 0; i < n; ++i) {
    int l = nearest_int(iscale * x[i]);
    l = imax(-nmax, imin(nmax - 1, l));
    L[i] = (int8_t)(l + nmax);
    float w = x[i] * x[i];
    sumlx += w * x[i] * l; suml2 += w * l * l;
  }
  float scale = suml2 ? sumlx / suml2 : 0.0f;
  float best = scale * sumlx;
  for (int is = -9; is <= 9; ++is) {
    if (is == 0) continue;
    iscale = -(nmax + 0.1f * is) / max;
    sumlx = suml2 = 0;
    for (int i = 0; i < n; ++i) {
      int l = nearest_int(iscale * x[i]);
      l = imax(-nmax, imin(nmax - 1, l));
      float w = x[i] * x[i];
      sumlx += w * x[i] * l; suml2 += w * l * l;
    }
    if (suml2 > 0 && sumlx * sumlx > best * suml2) {
      for (int i = 0; i < n; ++i) {
        int l = nearest_int(iscale * x[i]);
        L[i] = (int8_t)(nmax + imax(-nmax, imin(nmax - 1, l)));
      }
      scale = sumlx / suml2; best = scale * sumlx;
    }
  }
  return scale;
}

static void quantize_q4_5_K(int type, const float *x, uint8_t *y, int64_t k) {
  const int five = type == ORC_Q5_K;
  const int nmax = five ? 31 : 15;
  const int ts = orc_type_size(type);
  uint8_t L[QK_K], Laux[32];
  float weights[32], mins[8], scales[8];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += ts) {
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < 8; ++j) {
      float sum_x2 = 0;
      for (int l = 0; l < 32; ++l) sum_x2 += x[32 * j + l] * x[32 * j + l];
      float av_x = sqrtf(sum_x2 / 32);
      for (int l = 0; l < 32; ++l) weights[l] = av_x + fabsf(x[32 * j + l]);
      scales[j] = five ? make_qkx2_quants(32, 31, x + 32 * j, weights, L + 32 * j, &mins[j], Laux, -0.5f, 0.1f, 15, 0)
                       : make_qkx2_quants(32, 15, x + 32 * j, weights, L + 32 * j, &mins[j], Laux, -1.f, 0.1f, 20, 0);
      if (scales[j] > max_scale) max_scale = scales[j];
      if (mins[j] > max_min) max_min = mins[j];
    }
    float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f;
    float inv_min = max_min > 0 ? 63.f / max_min : 0.f;
    uint8_t *sc12 = y + 4;
    memset(sc12, 0, 12);
    for (int j = 0; j < 8; ++j) {
      uint8_t ls = (uint8_t)imin(63, nearest_int(inv_scale * scales[j]));
      uint8_t lm = (uint8_t)imin(63, nearest_int(inv_min * mins[j]));
      if (j < 4) { sc12[j] = ls; sc12[j + 4] = lm; }
      else {
        sc12[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
        sc12[j - 4] |= (uint8_t)((ls >> 4) << 6);
        sc12[j] |= (uint8_t)((lm >> 4) << 6);
      }
    }
    st16(y, orc_fp32_to_fp16(max_scale / 63.f));
    st16(y + 2, orc_fp32_to_fp16(max_min / 63.f));
    float dd = orc_fp16_to_fp32(ld16(y)), dmin = orc_fp16_to_fp32(ld16(y + 2));
    for (int j = 0; j < 8; ++j) {
      uint8_t sc, m; k4_scale_min(j, sc12, &sc, &m);
      float d = dd * sc;
      if (!d) { for (int ii = 0; ii < 32; ++ii) L[32 * j + ii] = 0; continue; }
      float dm = dmin * m;
      for (int ii = 0; ii < 32; ++ii) {
        int l = nearest_int((x[32 * j + ii] + dm) / d);
        L[32 * j + ii] = (uint8_t)imax(0, imin(nmax, l));
      }
    }
    if (!five) {
      uint8_t *q = y + 16;
      for (int j = 0; j < QK_K; j += 64, q += 32)
        for (int l = 0; l < 32; ++l) q[l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 4));
    } else {
      uint8_t *qh = y + 16, *ql = y + 48;
      memset(qh, 0, 32);
      uint8_t m1 = 1, m2 = 2;
      for (int n = 0; n < QK_K; n += 64, ql += 32, m1 <<= 2, m2 <<= 2)
        for (int j = 0; j < 32; ++j) {
          int l1 = L[n + j], l2 = L[n + j + 32];
          if (l1 > 15) { l1 -= 16; qh[j] |= m1; }
          if (l2 > 15) { l2 -= 16; qh[j] |= m2; }
          ql[j] = (uint8_t)(l1 | (l2 << 4));
        }
    }
  }
}

static void quantize_q6_K(const float *x, uint8_t *y, int64_t k) {
  int8_t L[QK_K];
  float scales[16];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 210) {
    float max_scale = 0, max_abs = 0;
    for (int ib = 0; ib < 16; ++ib) {
      float s = make_qx_quants(16, 32, x + 16 * ib, L + 16 * ib);
      scales[ib] = s;
      if (fabsf(s) > max_abs) { max_abs = fabsf(s); max_scale = s; }
    }
    if (max_abs < 1e-15f) { memset(y, 0, 210); continue; }
    float iscale = -128.f / max_scale;
    st16(y + 208, orc_fp32_to_fp16(1 / iscale));
    int8_t *sc = (int8_t *)(y + 192);
    for (int ib = 0; ib < 16; ++ib) sc[ib] = (int8_t)imin(127, nearest_int(iscale * scales[ib]));
    float dall = orc_fp16_to_fp32(ld16(y + 208));
    for (int j = 0; j < 16; ++j) {
      float d = dall * sc[j];
      if (!d) continue;
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int(x[16 * j + ii] / d);
        L[16 * j + ii] = (int8_t)(imax(-32, imin(31, l)) + 32);
      }
    }
    uint8_t *ql = y, *qh = y + 128;
    for (int j = 0; j < QK_K; j += 128, ql += 64, qh += 32)
      for (int l = 0; l < 32; ++l) {
        uint8_t q1 = L[j + l] & 0xF, q2 = L[j + l + 32] & 0xF, q3 = L[j + l + 64] & 0xF, q4 = L[j + l + 96] & 0xF;
        ql[l] = (uint8_t)(q1 | (q3 << 4));
        ql[l + 32] = (uint8_t)(q2 | (q4 << 4));
        qh[l] = (uint8_t)((L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6));
      }
  }
}

/* ---- Q2_K / Q3_K: public GGML quantize_row_q2_K_ref / quantize_row_q3_K_ref (make_qkx2_quants with |x| weights and mean-absolute error;
 * make_q3_quants with its 5 refinement sweeps).  ISQ targets IsqType::Q2K / Q3K (mistralrs-quant/src/lib.rs:818-819, utils/isq.rs generate_isq!).
 * As the other quantizers: a restatement of the published algorithm behind a dependency that is not in the tree ("parity unpinned"). */
static void quantize_q2_K(const float *x, uint8_t *y, int64_t k) {
  uint8_t L[QK_K], Laux[16];
  float weights[16], mins[16], scales[16];
  const float q4scale = 15.f;
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 84) {
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < 16; ++j) {
      for (int l = 0; l < 16; ++l) weights[l] = fabsf(x[16 * j + l]);
      scales[j] = make_qkx2_quants(16, 3, x + 16 * j, weights, L + 16 * j, &mins[j], Laux, -0.5f, 0.1f, 15, 1);
      if (scales[j] > max_scale) max_scale = scales[j];
      if (mins[j] > max_min) max_min = mins[j];
    }
    uint8_t *sc = y, *qs = y + 16;
    if (max_scale > 0) {
      float iscale = q4scale / max_scale;
      for (int j = 0; j < 16; ++j) sc[j] = (uint8_t)nearest_int(iscale * scales[j]);
      st16(y + 80, orc_fp32_to_fp16(max_scale / q4scale));
    } else {
      for (int j = 0; j < 16; ++j) sc[j] = 0;
      st16(y + 80, orc_fp32_to_fp16(0.f));
    }
    if (max_min > 0) {
      float iscale = q4scale / max_min;
      for (int j = 0; j < 16; ++j) sc[j] |= (uint8_t)(nearest_int(iscale * mins[j]) << 4);
      st16(y + 82, orc_fp32_to_fp16(max_min / q4scale));
    } else {
      st16(y + 82, orc_fp32_to_fp16(0.f));
    }
    const float dd = orc_fp16_to_fp32(ld16(y + 80)), dmin = orc_fp16_to_fp32(ld16(y + 82));
    for (int j = 0; j < 16; ++j) {
      const float d = dd * (sc[j] & 0xF);
      if (!d) { for (int ii = 0; ii < 16; ++ii) L[16 * j + ii] = 0; continue; }  /* as quantize_q4_5_K here: a scale that rounds to 0 stores zeros */
      const float dm = dmin * (sc[j] >> 4);
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int((x[16 * j + ii] + dm) / d);
        L[16 * j + ii] = (uint8_t)imax(0, imin(3, l));
      }
    }
    for (int j = 0; j < QK_K; j += 128)
      for (int l = 0; l < 32; ++l) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}

static float make_q3_quants(int n, int nmax, const float *x, int8_t *L) { /* do_rmse = true */
  float max = 0, amax = 0;
  for (int i = 0; i < n; ++i) { float ax = fabsf(x[i]); if (ax > amax) { amax = ax; max = x[i]; } }
  if (amax < 1e-15f) { for (int i = 0; i < n; ++i) L[i] = 0; return 0.f; }
  float iscale = -nmax / max, sumlx = 0, suml2 = 0;
  for (int i = 0; i < n; ++i) {
    int l = nearest_int(iscale * x[i]);
    l = imax(-nmax, imin(nmax - 1, l));
    L[i] = (int8_t)l;
    float w = x[i] * x[i];
    sumlx += w * x[i] * l; suml2 += w * l * l;
  }
  for (int itry = 0; itry < 5; ++itry) {
    int n_changed = 0;
    for (int i = 0; i < n; ++i) {
      float w = x[i] * x[i];
      float slx = sumlx - w * x[i] * L[i];
      if (slx > 0) {
        float sl2 = suml2 - w * L[i] * L[i];
        int new_l = nearest_int(x[i] * sl2 / slx);
        new_l = imax(-nmax, imin(nmax - 1, new_l));
        if (new_l != L[i]) {
          slx += w * x[i] * new_l; sl2 += w * new_l * new_l;
          if (sl2 > 0 && slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = (int8_t)new_l; sumlx = slx; suml2 = sl2; ++n_changed; }
        }
      }
    }
    if (!n_changed) break;
  }
  for (int i = 0; i < n; ++i) L[i] = (int8_t)(L[i] + nmax);
  return sumlx / suml2;
}

static void quantize_q3_K(const float *x, uint8_t *y, int64_t k) {
  int8_t L[QK_K];
  float scales[16];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 110) {
    float max_scale = 0, amax = 0;
    for (int j = 0; j < 16; ++j) {
      scales[j] = make_q3_quants(16, 4, x + 16 * j, L + 16 * j);
      float a = fabsf(scales[j]);
      if (a > amax) { amax = a; max_scale = scales[j]; }
    }
    uint8_t *hmask = y, *qs = y + 32, *sc12 = y + 96;
    memset(sc12, 0, 12);
    if (max_scale) {
      float iscale = -32.f / max_scale;
      for (int j = 0; j < 16; ++j) {
        int l = nearest_int(iscale * scales[j]);
        l = imax(-32, imin(31, l)) + 32;
        if (j < 8) sc12[j] = (uint8_t)(l & 0xF);
        else sc12[j - 8] |= (uint8_t)((l & 0xF) << 4);
        l >>= 4;
        sc12[j % 4 + 8] |= (uint8_t)(l << (2 * (j / 4)));
      }
      st16(y + 108, orc_fp32_to_fp16(1 / iscale));
    } else {
      st16(y + 108, orc_fp32_to_fp16(0.f));
    }
    int8_t sc16[16];
    q3k_scales(sc12, sc16);
    const float dd = orc_fp16_to_fp32(ld16(y + 108));
    for (int j = 0; j < 16; ++j) {
      float d = dd * sc16[j];
      if (!d) continue;  /* the quants of make_q3_quants (0..7) stay */
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int(x[16 * j + ii] / d);
        L[16 * j + ii] = (int8_t)(imax(-4, imin(3, l)) + 4);
      }
    }
    memset(hmask, 0, 32);
    int m = 0; uint8_t hm = 1;  /* the high bit of the first 32 quants goes to bit 0, of the next 32 to bit 1, ... */
    for (int j = 0; j < QK_K; ++j) {
      if (L[j] > 3) { hmask[m] |= hm; L[j] -= 4; }
      if (++m == 32) { m = 0; hm <<= 1; }
    }
    for (int j = 0; j < QK_K; j += 128)
      for (int l = 0; l < 32; ++l) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}

/* ---- importance-weighted K-quant quantizers: public GGML `quantize_row_q{4,5,6}_K_impl` with quant_weights (what candle's
 * QTensor::quantize_imatrix runs for these formats; candle is a git dependency that is not in the tree: "parity unpinned").  Call sites in the
 * reference: mistralrs-quant/src/gguf/mod.rs:238-252 (expert stacks), utils/isq.rs generate_isq_imatrix!.  qw has one entry per input column
 * (k of them), shared by every row. */
static float make_qp_quants(int n, int nmax, const float *x, uint8_t *L, const float *qw) {
  float max = 0;
  for (int i = 0; i < n; ++i) if (x[i] > max) max = x[i];
  if (!max) { for (int i = 0; i < n; ++i) L[i] = 0; return 0.f; }
  float iscale = nmax / max;
  for (int i = 0; i < n; ++i) L[i] = (uint8_t)nearest_int(iscale * x[i]);
  float scale = 1 / iscale, best_mse = 0;
  for (int i = 0; i < n; ++i) { float diff = x[i] - scale * L[i]; best_mse += qw[i] * diff * diff; }
  for (int is = -4; is <= 4; ++is) {
    if (is == 0) continue;
    float iscale_is = (0.1f * is + nmax) / max, scale_is = 1 / iscale_is, mse = 0;
    for (int i = 0; i < n; ++i) {
      int l = imin(nmax, nearest_int(iscale_is * x[i]));
      float diff = x[i] - scale_is * l;
      mse += qw[i] * diff * diff;
    }
    if (mse < best_mse) { best_mse = mse; iscale = iscale_is; }
  }
  float sumlx = 0, suml2 = 0;
  for (int i = 0; i < n; ++i) {
    int l = imin(nmax, nearest_int(iscale * x[i]));
    L[i] = (uint8_t)l;
    sumlx += qw[i] * x[i] * l; suml2 += qw[i] * l * l;
  }
  for (int itry = 0; itry < 5; ++itry) {
    int n_changed = 0;
    for (int i = 0; i < n; ++i) {
      float w = qw[i], slx = sumlx - w * x[i] * L[i], sl2 = suml2 - w * L[i] * L[i];
      if (slx > 0 && sl2 > 0) {
        int new_l = imin(nmax, nearest_int(x[i] * sl2 / slx));
        if (new_l != L[i]) {
          slx += w * x[i] * new_l; sl2 += w * new_l * new_l;
          if (slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = (uint8_t)new_l; sumlx = slx; suml2 = sl2; ++n_changed; }
        }
      }
    }
    if (!n_changed) break;
  }
  return sumlx / suml2;
}

static void quantize_q4_5_K_imatrix(int type, const float *x, uint8_t *y, int64_t k, const float *qw_row) {
  const int five = type == ORC_Q5_K;
  const int nmax = five ? 31 : 15;
  const int ts = orc_type_size(type);
  uint8_t L[QK_K], Laux[32], Ls[8], Lm[8];
  float weights[32], mins[8], scales[8], sw[8];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += ts) {
    float sum_x2 = 0;
    for (int l = 0; l < QK_K; ++l) sum_x2 += x[l] * x[l];
    const float sigma2 = 2 * sum_x2 / QK_K;
    for (int j = 0; j < 8; ++j) {
      const float *qw = qw_row + QK_K * i + 32 * j;
      for (int l = 0; l < 32; ++l) weights[l] = qw[l] * sqrtf(sigma2 + x[32 * j + l] * x[32 * j + l]);
      float sumw = 0;
      for (int l = 0; l < 32; ++l) sumw += weights[l];
      sw[j] = sumw;
      scales[j] = make_qkx2_quants(32, nmax, x + 32 * j, weights, L + 32 * j, &mins[j], Laux, -0.9f, 0.05f, 36, 0); /* make_qkx3_quants with weights given */
    }
    const float d_block = make_qp_quants(8, 63, scales, Ls, sw), m_block = make_qp_quants(8, 63, mins, Lm, sw);
    uint8_t *sc12 = y + 4;
    memset(sc12, 0, 12);
    for (int j = 0; j < 8; ++j) {
      uint8_t ls = Ls[j], lm = Lm[j];
      if (j < 4) { sc12[j] = ls; sc12[j + 4] = lm; }
      else {
        sc12[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
        sc12[j - 4] |= (uint8_t)((ls >> 4) << 6);
        sc12[j] |= (uint8_t)((lm >> 4) << 6);
      }
    }
    st16(y, orc_fp32_to_fp16(d_block));
    st16(y + 2, orc_fp32_to_fp16(m_block));
    float dd = orc_fp16_to_fp32(ld16(y)), dmin = orc_fp16_to_fp32(ld16(y + 2));
    for (int j = 0; j < 8; ++j) {
      uint8_t sc, m; k4_scale_min(j, sc12, &sc, &m);
      float d = dd * sc;
      if (!d) { for (int ii = 0; ii < 32; ++ii) L[32 * j + ii] = 0; continue; }
      float dm = dmin * m;
      for (int ii = 0; ii < 32; ++ii) {
        int l = nearest_int((x[32 * j + ii] + dm) / d);
        L[32 * j + ii] = (uint8_t)imax(0, imin(nmax, l));
      }
    }
    if (!five) {
      uint8_t *q = y + 16;
      for (int j = 0; j < QK_K; j += 64, q += 32)
        for (int l = 0; l < 32; ++l) q[l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 4));
    } else {
      uint8_t *qh = y + 16, *ql = y + 48;
      memset(qh, 0, 32);
      uint8_t m1 = 1, m2 = 2;
      for (int n = 0; n < QK_K; n += 64, ql += 32, m1 <<= 2, m2 <<= 2)
        for (int j = 0; j < 32; ++j) {
          int l1 = L[n + j], l2 = L[n + j + 32];
          if (l1 > 15) { l1 -= 16; qh[j] |= m1; }
          if (l2 > 15) { l2 -= 16; qh[j] |= m2; }
          ql[j] = (uint8_t)(l1 | (l2 << 4));
        }
    }
  }
}

static float make_qx_quants_w(int n, int nmax, const float *x, int8_t *L, const float *qw) { /* rmse_type 1 with the weights given */
  float max = 0, amax = 0;
  for (int i = 0; i < n; ++i) { float ax = fabsf(x[i]); if (ax > amax) { amax = ax; max = x[i]; } }
  if (amax < 1e-15f) { for (int i = 0; i < n; ++i) L[i] = 0; return 0.f; }
  float iscale = -nmax / max, sumlx = 0, suml2 = 0;
  for (int i = 0; i < n; ++i) {
    int l = nearest_int(iscale * x[i]);
    l = imax(-nmax, imin(nmax - 1, l));
    L[i] = (int8_t)(l + nmax);
    sumlx += qw[i] * x[i] * l; suml2 += qw[i] * l * l;
  }
  float scale = suml2 ? sumlx / suml2 : 0.0f;
  float best = scale * sumlx;
  for (int is = -9; is <= 9; ++is) {
    if (is == 0) continue;
    iscale = -(nmax + 0.1f * is) / max;
    sumlx = suml2 = 0;
    for (int i = 0; i < n; ++i) {
      int l = nearest_int(iscale * x[i]);
      l = imax(-nmax, imin(nmax - 1, l));
      sumlx += qw[i] * x[i] * l; suml2 += qw[i] * l * l;
    }
    if (suml2 > 0 && sumlx * sumlx > best * suml2) {
      for (int i = 0; i < n; ++i) {
        int l = nearest_int(iscale * x[i]);
        L[i] = (int8_t)(nmax + imax(-nmax, imin(nmax - 1, l)));
      }
      scale = sumlx / suml2; best = scale * sumlx;
    }
  }
  return scale;
}

static void quantize_q6_K_imatrix(const float *x, uint8_t *y, int64_t k, const float *qw_row) {
  int8_t L[QK_K];
  float scales[16];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 210) {
    float max_scale = 0, max_abs = 0;
    for (int ib = 0; ib < 16; ++ib) {
      float s = make_qx_quants_w(16, 32, x + 16 * ib, L + 16 * ib, qw_row + QK_K * i + 16 * ib); /* the importance values themselves are the weights */
      scales[ib] = s;
      if (fabsf(s) > max_abs) { max_abs = fabsf(s); max_scale = s; }
    }
    if (max_abs < 1e-15f) { memset(y, 0, 210); continue; }
    float iscale = -128.f / max_scale;
    st16(y + 208, orc_fp32_to_fp16(1 / iscale));
    int8_t *sc = (int8_t *)(y + 192);
    for (int ib = 0; ib < 16; ++ib) sc[ib] = (int8_t)imin(127, nearest_int(iscale * scales[ib]));
    float dall = orc_fp16_to_fp32(ld16(y + 208));
    for (int j = 0; j < 16; ++j) {
      float d = dall * sc[j];
      if (!d) continue;
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int(x[16 * j + ii] / d);
        L[16 * j + ii] = (int8_t)(imax(-32, imin(31, l)) + 32);
      }
    }
    uint8_t *ql = y, *qh = y + 128;
    for (int j = 0; j < QK_K; j += 128, ql += 64, qh += 32)
      for (int l = 0; l < 32; ++l) {
        uint8_t q1 = L[j + l] & 0xF, q2 = L[j + l + 32] & 0xF, q3 = L[j + l + 64] & 0xF, q4 = L[j + l + 96] & 0xF;
        ql[l] = (uint8_t)(q1 | (q3 << 4));
        ql[l + 32] = (uint8_t)(q2 | (q4 << 4));
        qh[l] = (uint8_t)((L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6));
      }
  }
}

/* Q2_K / Q3_K with importance weights: GGML quantize_row_q2_K_impl / quantize_row_q3_K_impl (as above: published algorithm, unpinned). */
static void quantize_q2_K_imatrix(const float *x, uint8_t *y, int64_t k, const float *qw_row) {
  uint8_t L[QK_K], Laux[16], Ls[16], Lm[16];
  float weight[16], mins[16], scales[16], sw[16];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 84) {
    float sumx2 = 0;
    for (int j = 0; j < QK_K; ++j) sumx2 += x[j] * x[j];
    const float sigma2 = sumx2 / QK_K;
    for (int j = 0; j < 16; ++j) {
      const float *qw = qw_row + QK_K * i + 16 * j;
      for (int l = 0; l < 16; ++l) weight[l] = qw[l] * sqrtf(sigma2 + x[16 * j + l] * x[16 * j + l]);
      sw[j] = 0;
      for (int l = 0; l < 16; ++l) sw[j] += weight[l];
      scales[j] = make_qkx2_quants(16, 3, x + 16 * j, weight, L + 16 * j, &mins[j], Laux, -0.9f, 0.05f, 36, 0);
    }
    float dm = make_qp_quants(16, 15, scales, Ls, sw), mm = make_qp_quants(16, 15, mins, Lm, sw);
    st16(y + 80, orc_fp32_to_fp16(dm));
    st16(y + 82, orc_fp32_to_fp16(mm));
    dm = orc_fp16_to_fp32(ld16(y + 80));
    mm = orc_fp16_to_fp32(ld16(y + 82));
    uint8_t *sc = y, *qs = y + 16;
    for (int j = 0; j < 16; ++j) sc[j] = (uint8_t)(Ls[j] | (Lm[j] << 4));
    for (int j = 0; j < 16; ++j) {
      const float d = dm * (sc[j] & 0xF);
      if (!d) { for (int ii = 0; ii < 16; ++ii) L[16 * j + ii] = 0; continue; }  /* as the plain quantizer here */
      const float m = mm * (sc[j] >> 4);
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int((x[16 * j + ii] + m) / d);
        L[16 * j + ii] = (uint8_t)imax(0, imin(3, l));
      }
    }
    for (int j = 0; j < QK_K; j += 128)
      for (int l = 0; l < 32; ++l) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}

static void quantize_q3_K_imatrix(const float *x, uint8_t *y, int64_t k, const float *qw_row) {
  int8_t L[QK_K], Ls[16];
  float scales[16], weight[16], sw[16];
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 110) {
    float sumx2 = 0;
    for (int j = 0; j < QK_K; ++j) sumx2 += x[j] * x[j];
    const float sigma2 = 2 * sumx2 / QK_K;
    for (int j = 0; j < 16; ++j) {
      const float *qw = qw_row + QK_K * i + 16 * j;
      for (int l = 0; l < 16; ++l) weight[l] = qw[l] * sqrtf(sigma2 + x[16 * j + l] * x[16 * j + l]);
      float sumw = 0;
      for (int l = 0; l < 16; ++l) sumw += weight[l];
      sw[j] = sumw;
      scales[j] = make_qx_quants_w(16, 4, x + 16 * j, L + 16 * j, weight);
    }
    uint8_t *hmask = y, *qs = y + 32, *sc12 = y + 96;
    memset(sc12, 0, 12);
    const float d_block = make_qx_quants_w(16, 32, scales, Ls, sw);
    for (int j = 0; j < 16; ++j) {
      int l = Ls[j];
      if (j < 8) sc12[j] = (uint8_t)(l & 0xF);
      else sc12[j - 8] |= (uint8_t)((l & 0xF) << 4);
      l >>= 4;
      sc12[j % 4 + 8] |= (uint8_t)(l << (2 * (j / 4)));
    }
    st16(y + 108, orc_fp32_to_fp16(d_block));
    int8_t sc16[16];
    q3k_scales(sc12, sc16);
    const float dd = orc_fp16_to_fp32(ld16(y + 108));
    for (int j = 0; j < 16; ++j) {
      float d = dd * sc16[j];
      if (!d) continue;
      for (int ii = 0; ii < 16; ++ii) {
        int l = nearest_int(x[16 * j + ii] / d);
        L[16 * j + ii] = (int8_t)(imax(-4, imin(3, l)) + 4);
      }
    }
    memset(hmask, 0, 32);
    int m = 0; uint8_t hm = 1;
    for (int j = 0; j < QK_K; ++j) {
      if (L[j] > 3) { hmask[m] |= hm; L[j] -= 4; }
      if (++m == 32) { m = 0; hm <<= 1; }
    }
    for (int j = 0; j < QK_K; j += 128)
      for (int l = 0; l < 32; ++l) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}

/* one row of k values with the importance vector qw[k]; 0 ok, -1 for a type without a weighted quantizer here */
int orc_quantize_row_imatrix(int type, const float *x, void *blocks, int64_t k, const float *qw) {
  switch (type) {
  case ORC_Q4_K: case ORC_Q5_K: quantize_q4_5_K_imatrix(type, x, (uint8_t *)blocks, k, qw); return 0;
  case ORC_Q6_K: quantize_q6_K_imatrix(x, (uint8_t *)blocks, k, qw); return 0;
  case ORC_Q2_K: quantize_q2_K_imatrix(x, (uint8_t *)blocks, k, qw); return 0;
  case ORC_Q3_K: quantize_q3_K_imatrix(x, (uint8_t *)blocks, k, qw); return 0;
  default: return -1;
  }
}

static void quantize_legacy(int type, const float *x, uint8_t *y, int64_t k) {
  const int ts = orc_type_size(type);
  for (int64_t i = 0; i < k / 32; ++i, x += 32, y += ts) {
    if (type == ORC_Q8_0) {
      float amax = 0;
      for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(x[j]));
      float d = amax / 127.f, id = d ? 1.f / d : 0.f;
      st16(y, orc_fp32_to_fp16(d));
      for (int j = 0; j < 32; ++j) ((int8_t *)(y + 2))[j] = (int8_t)roundf(x[j] * id);
    } else if (type == ORC_Q4_0 || type == ORC_Q5_0) {
      const int half_range = type == ORC_Q4_0 ? 8 : 16, top = 2 * half_range - 1;
      float amax = 0, max = 0;
      for (int j = 0; j < 32; ++j) if (fabsf(x[j]) > amax) { amax = fabsf(x[j]); max = x[j]; }
      float d = max / -(float)half_range, id = d ? 1.f / d : 0.f;
      st16(y, orc_fp32_to_fp16(d));
      uint8_t *qs = y + (type == ORC_Q4_0 ? 2 : 6);
      uint32_t qh = 0;
      for (int j = 0; j < 16; ++j) {
        int x0 = imin(top, (int)(int8_t)(x[j] * id + (half_range + 0.5f)));
        int x1 = imin(top, (int)(int8_t)(x[j + 16] * id + (half_range + 0.5f)));
        qs[j] = (uint8_t)((x0 & 0xF) | ((x1 & 0xF) << 4));
        qh |= (uint32_t)((x0 & 0x10) >> 4) << j;
        qh |= (uint32_t)((x1 & 0x10) >> 4) << (j + 16);
      }
      if (type == ORC_Q5_0) memcpy(y + 2, &qh, 4);
    } else { /* Q4_1 / Q5_1 */
      const int top = type == ORC_Q4_1 ? 15 : 31;
      float mn = x[0], mx = x[0];
      for (int j = 1; j < 32; ++j) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; } /* comparisons as GGML's quantize_row_q4_1_ref: the first of +0 / -0 stays (fminf leaves that open) */
      float d = (mx - mn) / top, id = d ? 1.f / d : 0.f;
      st16(y, orc_fp32_to_fp16(d));
      st16(y + 2, orc_fp32_to_fp16(mn));
      uint8_t *qs = y + (type == ORC_Q4_1 ? 4 : 8);
      uint32_t qh = 0;
      for (int j = 0; j < 16; ++j) {
        int x0 = imin(top, (int)(uint8_t)((x[j] - mn) * id + 0.5f));
        int x1 = imin(top, (int)(uint8_t)((x[j + 16] - mn) * id + 0.5f));
        qs[j] = (uint8_t)((x0 & 0xF) | ((x1 & 0xF) << 4));
        qh |= (uint32_t)((x0 & 0x10) >> 4) << j;
        qh |= (uint32_t)((x1 & 0x10) >> 4) << (j + 16);
      }
      if (type == ORC_Q5_1) memcpy(y + 4, &qh, 4);
    }
  }
}

int orc_quantize_row(int type, const float *x, void *blocks, int64_t k) {
  switch (type) {
  case ORC_F32: memcpy(blocks, x, (size_t)k * 4); return 0;
  case ORC_F16: for (int64_t i = 0; i < k; ++i) ((uint16_t *)blocks)[i] = orc_fp32_to_fp16(x[i]); return 0;
  case ORC_BF16: for (int64_t i = 0; i < k; ++i) ((uint16_t *)blocks)[i] = orc_fp32_to_bf16(x[i]); return 0;
  case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_Q8_0: quantize_legacy(type, x, blocks, k); return 0;
  case ORC_Q2_K: quantize_q2_K(x, blocks, k); return 0;
  case ORC_Q3_K: quantize_q3_K(x, blocks, k); return 0;
  case ORC_Q4_K: case ORC_Q5_K: quantize_q4_5_K(type, x, blocks, k); return 0;
  case ORC_Q6_K: quantize_q6_K(x, blocks, k); return 0;
  case ORC_Q8_K: orc_quantize_q8_K(x, blocks, k); return 0;
  default: return -1;
  }
}

/* splitmix64 */
static inline uint64_t rng_next(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void orc_random_blocks(int type, void *blocks, int64_t n_blocks, uint64_t seed, float d_scale) {
  const int ts = orc_type_size(type);
  uint8_t *b = blocks;
  uint64_t s = seed * 0x2545F4914F6CDD1Dull + 1;
  for (int64_t i = 0; i < n_blocks; ++i, b += ts) {
    for (int j = 0; j < ts; j += 8) { uint64_t r = rng_next(&s); memcpy(b + j, &r, (size_t)(ts - j < 8 ? ts - j : 8)); }
    uint64_t r = rng_next(&s);
    float d = d_scale * (0.5f + (float)(r & 0xffff) / 65536.f);
    float m = d_scale * (0.5f + (float)((r >> 16) & 0xffff) / 65536.f);
    if ((r >> 40) & 1) d = -d; /* negative super-scales are legal and do occur */
    switch (type) {
    case ORC_Q4_0: case ORC_Q5_0: case ORC_Q8_0: st16(b, orc_fp32_to_fp16(d)); break;
    case ORC_Q4_1: case ORC_Q5_1: st16(b, orc_fp32_to_fp16(fabsf(d))); st16(b + 2, orc_fp32_to_fp16(-8.f * m)); break;
    case ORC_Q8_1: { /* a consistent block: s = d * sum(q), as the quantizer writes it */
      int sm = 0; for (int j = 0; j < 32; ++j) sm += (int8_t)b[4 + j];
      st16(b, orc_fp32_to_fp16(d)); st16(b + 2, orc_fp32_to_fp16(d * (float)sm)); } break;
    case ORC_Q2_K: st16(b + 80, orc_fp32_to_fp16(fabsf(d))); st16(b + 82, orc_fp32_to_fp16(m)); break;
    case ORC_Q3_K: st16(b + 108, orc_fp32_to_fp16(d / 16.f)); break;
    case ORC_Q4_K: case ORC_Q5_K: st16(b, orc_fp32_to_fp16(fabsf(d) / 32.f)); st16(b + 2, orc_fp32_to_fp16(m / 4.f)); break;
    case ORC_Q6_K: st16(b + 208, orc_fp32_to_fp16(d / 64.f)); break;
    case ORC_Q8_K: { float df = d / 64.f; memcpy(b, &df, 4);
      int q[256]; block_ints(type, b, q);
      for (int g = 0; g < 16; ++g) { int sm = 0; for (int e = 0; e < 16; ++e) sm += q[g * 16 + e]; int16_t v = (int16_t)sm; memcpy(b + 260 + 2 * g, &v, 2); } } break;
    default: break;
    }
  }
}

/* ------------------------------------------------------------------ activation quantizers */
void orc_quantize_q8_1(const float *x, void *vy, int kx, int kx_padded, int rows) {
  uint8_t *y = vy;
  for (int r = 0; r < rows; ++r)
    for (int ib = 0; ib < kx_padded / 32; ++ib) {
      uint8_t *blk = y + ((size_t)r * (kx_padded / 32) + ib) * 36;
      float v[32], amax = 0, sum = 0;
      for (int j = 0; j < 32; ++j) {
        int ix = ib * 32 + j;
        v[j] = ix < kx ? x[(size_t)r * kx + ix] : 0.0f;
        amax = fmaxf(amax, fabsf(v[j]));
      }
      /* reference sums with a 32-lane xor butterfly (mmvq_gguf.cu:26-33): reproduce that order */
      float t[32];
      memcpy(t, v, sizeof t);
      for (int mask = 16; mask > 0; mask >>= 1) { float u[32]; for (int l = 0; l < 32; ++l) u[l] = t[l] + t[l ^ mask]; memcpy(t, u, sizeof t); }
      sum = t[0];
      const float d = amax / 127.0f;
      for (int j = 0; j < 32; ++j) ((int8_t *)(blk + 4))[j] = amax == 0.0f ? 0 : (int8_t)roundf(v[j] / d);
      st16(blk, orc_fp32_to_fp16(d));
      st16(blk + 2, orc_fp32_to_fp16(sum));
    }
}

/* block_q8_1_mmq quantizer (reference: mistralrs-quant/kernels/mmq_gguf/mmq_quantize.cu:104-198 quantize_mmq_q8_1, launch geometry :328-357;
 * block layout mmq_gguf.cuh:70-89).  144-byte blocks of 128 values: 16 header bytes + 128 int8; block index = (i0/128)*ne1 + i1 (k-block major,
 * token minor).  layout 0 = D4 (4 x f32 d), 1 = DS4 (4 x (half d, half sum)), 2 = D2S6 (half d per 64 values, half sum per 16 values for the first 96).
 * d_inv = 127/amax, q = roundf(x*d_inv), d = 1/d_inv; sums: 4 values per thread left to right, then the xor butterfly (offsets n/8 .. 1) over the
 * threads of the group.  amax == 0: q = 0, d = 0 (the device's float->int conversion of NaN). */
static float mmq_butterfly_sum(const float *t4, int nthreads) { /* t4: per-thread partial sums */
  float t[8], u[8];
  memcpy(t, t4, sizeof(float) * nthreads);
  for (int off = nthreads / 2; off > 0; off >>= 1) { for (int l = 0; l < nthreads; ++l) u[l] = t[l] + t[l ^ off]; memcpy(t, u, sizeof(float) * nthreads); }
  return t[0];
}
void orc_quantize_q8_1_mmq(const float *x, const int32_t *ids, void *vy, int layout, int64_t ne00, int64_t s01, int64_t ne0, int64_t ne1) {
  uint8_t *y = vy;
  const int per_scale = layout == 2 ? 64 : 32, per_sum = layout == 2 ? 16 : 32;
  for (int64_t i1 = 0; i1 < ne1; ++i1) {
    const int64_t row = ids ? ids[i1] : i1;
    for (int64_t kb = 0; kb < (ne0 + 127) / 128; ++kb) {
      uint8_t *blk = y + (size_t)(kb * ne1 + i1) * 144;
      float v[128];
      for (int j = 0; j < 128; ++j) { const int64_t i0 = kb * 128 + j; v[j] = (i0 < ne00 && i0 < ne0) ? x[row * s01 + i0] : 0.0f; }
      for (int g = 0; g < 128 / per_scale; ++g) {
        float amax = 0.0f;
        for (int j = 0; j < per_scale; ++j) amax = fmaxf(amax, fabsf(v[g * per_scale + j]));
        const float d_inv = 127.0f / amax, d = amax == 0.0f ? 0.0f : 1.0f / d_inv;
        for (int j = 0; j < per_scale; ++j) ((int8_t *)(blk + 16))[g * per_scale + j] = amax == 0.0f ? 0 : (int8_t)roundf(v[g * per_scale + j] * d_inv);
        if (layout == 0) memcpy(blk + 4 * g, &d, 4);
        else st16(blk + (layout == 1 ? 4 * g : 2 * g), orc_fp32_to_fp16(d));
      }
      if (layout != 0)
        for (int g = 0; g < (layout == 2 ? 6 : 4); ++g) {
          float t4[8];
          for (int t = 0; t < per_sum / 4; ++t) { const float *q = v + g * per_sum + 4 * t; t4[t] = q[0] + q[1] + q[2] + q[3]; }
          const float sum = mmq_butterfly_sum(t4, per_sum / 4);
          st16(blk + (layout == 1 ? 4 * g + 2 : 4 + 2 * g), orc_fp32_to_fp16(sum));
        }
    }
  }
}

/* MMQ maths (reference: mmq_vecdotq.cuh vec_dot_*_q8_1_impl_mmq / _dp4a, tile loaders of mmq_gguf.cuh): integer dots of the weight ints with the
 * block_q8_1_mmq ints, scaled by (weight scale * activation d); the weight offsets meet the STORED partial sums where the layout carries one
 * (DS4: half(sum x) of the 32-block; D2S6: half(sum x) of the 16-value run for the first 96 values of a 128-block) and d*SUM(u) otherwise.
 * f64 combination: order-free reference for any f32 summation order.  out[col*N + row]. */
static int mmq_layout_of(int type) {
  switch (type) {
  case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_1: case ORC_Q4_K: case ORC_Q5_K: return 1;
  case ORC_Q2_K: return 2;
  default: return 0;
  }
}
int orc_mmq_layout(int type) { return mmq_layout_of(type); }
static double row_dot_q8_1_mmq(int type, const uint8_t *wrow, int K, const uint8_t *y, int64_t ncols_y, int64_t col, double *mag) {
  const int blk = orc_block_size(type), ts = orc_type_size(type), layout = mmq_layout_of(type);
  int q[256]; float sc[16], off[16];
  double acc = 0, m = 0;
  for (int ib = 0; ib < K / blk; ++ib) {
    const uint8_t *b = wrow + (size_t)ib * ts;
    block_ints(type, b, q);
    const int gs = block_affine(type, b, sc, off);
    const int step = gs < 16 ? gs : 16; /* 16-value runs: the finest granularity any layout stores a sum at */
    for (int s = 0; s < blk / step; ++s) {
      const int e0 = s * step, g = e0 / gs;
      const int64_t e = (int64_t)ib * blk + e0;
      const uint8_t *yb = y + (size_t)((e / 128) * ncols_y + col) * 144;
      const int r = (int)(e % 128);
      const int8_t *u = (const int8_t *)(yb + 16) + r;
      float d8;
      if (layout == 0) d8 = ldf32(yb + 4 * (r / 32));
      else d8 = orc_fp16_to_fp32(ld16(yb + (layout == 1 ? 4 * (r / 32) : 2 * (r / 64))));
      int dot = 0, su = 0;
      for (int j = 0; j < step; ++j) { dot += q[e0 + j] * u[j]; su += u[j]; }
      double so; /* the offset's partner */
      if (layout == 1) so = 0.5 * (double)orc_fp16_to_fp32(ld16(yb + 4 * (r / 32) + 2)); /* two 16-runs share the 32-block's stored sum */
      else if (layout == 2 && r < 96) so = (double)orc_fp16_to_fp32(ld16(yb + 4 + 2 * (r / 16)));
      else so = (double)d8 * su;
      const double a = (double)sc[g] * (double)d8 * dot, o = (double)off[g] * so;
      acc += a - o;
      m += fabs(a) + fabs(o);
    }
  }
  if (mag) *mag = m;
  return acc;
}
void orc_matmul_q8_1_mmq(int type, const void *W, int N, int K, int64_t stride_row_x, const void *y, int64_t ncols_y, float *out, float *mag) {
  const size_t row_bytes = (size_t)stride_row_x * orc_type_size(type);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int64_t c = 0; c < ncols_y; ++c) {
      double m;
      out[(size_t)c * N + n] = (float)row_dot_q8_1_mmq(type, (const uint8_t *)W + n * row_bytes, K, y, ncols_y, c, &m);
      if (mag) mag[(size_t)c * N + n] = (float)m;
    }
}

void orc_quantize_q8_K(const float *x, void *vy, int64_t k) {
  uint8_t *y = vy;
  for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K, y += 292) {
    float max = 0, amax = 0;
    for (int j = 0; j < QK_K; ++j) { float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
    if (amax == 0) { memset(y, 0, 292); continue; }
    const float iscale = -128.f / max;
    int8_t *qs = (int8_t *)(y + 4);
    for (int j = 0; j < QK_K; ++j) { float v = roundf(iscale * x[j]); qs[j] = (int8_t)(v > 127.f ? 127.f : v); }
    for (int j = 0; j < 16; ++j) { int s = 0; for (int l = 0; l < 16; ++l) s += qs[j * 16 + l]; int16_t v = (int16_t)s; memcpy(y + 260 + 2 * j, &v, 2); }
    float d = 1.0f / iscale;
    memcpy(y, &d, 4);
  }
}

/* ------------------------------------------------------------------ matmul oracles */
static int g_threads = 0;
void orc_set_threads(int n) { g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}
int orc_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_matmul_exact(int type, const void *W, int N, int K, const float *X, int B, float *out) {
  const size_t row_bytes = (size_t)(K / orc_block_size(type)) * orc_type_size(type);
#pragma omp parallel
  {
    float *w = malloc((size_t)K * sizeof(float));
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) {
      orc_dequantize_row(type, (const uint8_t *)W + n * row_bytes, w, K);
      for (int b = 0; b < B; ++b) {
        double acc = 0;
        const float *x = X + (size_t)b * K;
        for (int kk = 0; kk < K; ++kk) acc += (double)w[kk] * (double)x[kk];
        out[(size_t)b * N + n] = (float)acc;
      }
    }
    free(w);
  }
}

/* one weight row against one Q8_1 activation row, MMVQ maths, f64 combination.
 *   w = sc_g*q - off_g ; x ~ d8_j*u   (j = 32-block of the activation)
 *   sum = SUM_g,j sc_g*d8_j*<q,u>_{g∩j}  -  off_g*d8_j*SUM(u)_{g∩j}
 * which is algebraically what every vec_dot_*_q8_1 computes (mmvq_gguf.cu:240-700), except
 * that Q4_0/Q4_1/Q5_0/Q5_1 fold the offset through the stored s = half(sum x) term instead of
 * d8*SUM(u) (vec_dot_q4_0_q8_1_impl, :244-258): handled explicitly below.
 * Q8_1 as a weight format exists only in the MoE kernels (indexed_moe.cu:483-502, moe_grouped.cu:471-490): each of the QI8_1 / VDR = 4
 * vec_dot calls of a block returns d_w d_x sumi + s_w s_x, so a block contributes d_w d_x <q,u> + 4 s_w s_x (the stored sums multiplied,
 * four times over -- the reference's literal arithmetic, restated as offset -4 s_w against the stored s_x). */
static double row_dot_q8_1(int type, const uint8_t *wrow, int K, const uint8_t *y, double *mag) {
  const int blk = orc_block_size(type), ts = orc_type_size(type);
  int q[256]; float sc[16], off[16];
  double acc = 0, m = 0;
  for (int ib = 0; ib < K / blk; ++ib) {
    const uint8_t *b = wrow + (size_t)ib * ts;
    block_ints(type, b, q);
    const int gs = block_affine(type, b, sc, off);
    for (int s = 0; s < blk / (gs < 32 ? gs : 32); ++s) { /* step = min(group, 32) */
      const int step = gs < 32 ? gs : 32;
      const int e0 = s * step;
      const int g = e0 / gs;
      const uint8_t *yb = y + (size_t)((ib * blk + e0) / 32) * 36;
      const float d8 = orc_fp16_to_fp32(ld16(yb));
      const int8_t *u = (const int8_t *)(yb + 4) + (e0 % 32);
      int dot = 0, su = 0;
      for (int e = 0; e < step; ++e) { dot += q[e0 + e] * u[e]; su += u[e]; }
      if (type == ORC_Q8_1) off[g] = -4.0f * orc_fp16_to_fp32(ld16(b + 2)); /* Q8_1 WEIGHTS (MoE kernels only): see the note above */
      if (type == ORC_Q4_0 || type == ORC_Q5_0 || type == ORC_Q4_1 || type == ORC_Q5_1 || type == ORC_Q8_1) {
        const float s8 = orc_fp16_to_fp32(ld16(yb + 2));
        acc += (double)sc[g] * (double)d8 * dot - (double)off[g] * (double)s8;
        m += fabs((double)sc[g] * (double)d8 * dot) + fabs((double)off[g] * (double)s8);
      } else {
        acc += (double)sc[g] * (double)d8 * dot - (double)off[g] * (double)d8 * su;
        m += fabs((double)sc[g] * (double)d8 * dot) + fabs((double)off[g] * (double)d8 * su);
      }
    }
  }
  if (mag) *mag = m;
  return acc;
}

void orc_matmul_q8_1_ex(int type, const void *W, int N, int K, const void *y, int stride_col_y, int B, float *out,
                        float *mag) {
  const size_t row_bytes = (size_t)(K / orc_block_size(type)) * orc_type_size(type);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int b = 0; b < B; ++b) {
      double m;
      out[(size_t)b * N + n] = (float)row_dot_q8_1(type, (const uint8_t *)W + n * row_bytes, K,
                                                   (const uint8_t *)y + (size_t)b * stride_col_y * 36, &m);
      if (mag) mag[(size_t)b * N + n] = (float)m;
    }
}
void orc_matmul_q8_1(int type, const void *W, int N, int K, const void *y, int stride_col_y, int B, float *out) {
  orc_matmul_q8_1_ex(type, W, N, K, y, stride_col_y, B, out, NULL);
}

/* candle / ggml generic CPU dot products ------------------------------------------ */
static float dot_kquant_q8K(int type, const uint8_t *wrow, int K, const uint8_t *y) {
  /* ggml generic order: 8 f32 lanes `sums[l]` over aux32[l], mins folded through bsums */
  float sums[8] = {0}, sumf = 0;
  int q[256];
  for (int ib = 0; ib < K / QK_K; ++ib) {
    const uint8_t *b = wrow + (size_t)ib * orc_type_size(type);
    const uint8_t *yb = y + (size_t)ib * 292;
    const float yd = ldf32(yb);
    const int8_t *q8 = (const int8_t *)(yb + 4);
    int16_t bsums[16]; memcpy(bsums, yb + 260, 32);
    block_ints(type, b, q);
    int32_t aux32[8] = {0};
    if (type == ORC_Q4_K || type == ORC_Q5_K) {
      uint8_t sc[8], mn[8];
      for (int j = 0; j < 8; ++j) k4_scale_min(j, b + 4, &sc[j], &mn[j]);
      int sumi = 0;
      for (int j = 0; j < 16; ++j) sumi += bsums[j] * mn[j / 2];
      for (int j = 0; j < 8; ++j)
        for (int c = 0; c < 4; ++c)
          for (int l = 0; l < 8; ++l) aux32[l] += (int32_t)sc[j] * (int16_t)(q8[j * 32 + c * 8 + l] * q[j * 32 + c * 8 + l]);
      const float d = orc_fp16_to_fp32(ld16(b)) * yd;
      for (int l = 0; l < 8; ++l) sums[l] += d * (float)aux32[l];
      const float dmin = orc_fp16_to_fp32(ld16(b + 2)) * yd;
      sumf -= dmin * (float)sumi;
    } else if (type == ORC_Q6_K) {
      const int8_t *sc = (const int8_t *)(b + 192);
      for (int j = 0; j < 16; ++j)
        for (int c = 0; c < 2; ++c)
          for (int l = 0; l < 8; ++l) aux32[l] += (int32_t)sc[j] * (int16_t)(q8[j * 16 + c * 8 + l] * (q[j * 16 + c * 8 + l] - 32));
      const float d = orc_fp16_to_fp32(ld16(b + 208)) * yd;
      for (int l = 0; l < 8; ++l) sums[l] += d * (float)aux32[l];
    } else if (type == ORC_Q2_K) {
      int summs = 0, isum = 0;
      for (int j = 0; j < 16; ++j) summs += bsums[j] * (b[j] >> 4);
      for (int j = 0; j < 16; ++j) { int s = 0; for (int l = 0; l < 16; ++l) s += q8[j * 16 + l] * q[j * 16 + l]; isum += (b[j] & 0xF) * s; }
      const float dall = yd * orc_fp16_to_fp32(ld16(b + 80)), dmin = yd * orc_fp16_to_fp32(ld16(b + 82));
      sumf += dall * (float)isum - dmin * (float)summs;
    } else if (type == ORC_Q3_K) {
      int8_t sc[16]; q3k_scales(b + 96, sc);
      for (int j = 0; j < 16; ++j)
        for (int c = 0; c < 2; ++c)
          for (int l = 0; l < 8; ++l) aux32[l] += (int32_t)sc[j] * (int16_t)(q8[j * 16 + c * 8 + l] * (q[j * 16 + c * 8 + l] - 4));
      const float d = orc_fp16_to_fp32(ld16(b + 108)) * yd;
      for (int l = 0; l < 8; ++l) sums[l] += d * (float)aux32[l];
    }
  }
  for (int l = 0; l < 8; ++l) sumf += sums[l];
  return sumf;
}

static float dot_legacy_q8(int type, const uint8_t *wrow, int K, const uint8_t *y /* q8_0 (34B) or q8_1-cpu (36B: d, s=d*sum q) */) {
  const int ts = orc_type_size(type);
  const int with_s = (type == ORC_Q4_1 || type == ORC_Q5_1);
  int q[32];
  float sumf = 0;
  for (int ib = 0; ib < K / 32; ++ib) {
    const uint8_t *b = wrow + (size_t)ib * ts;
    const uint8_t *yb = y + (size_t)ib * (with_s ? 36 : 34);
    const int8_t *u = (const int8_t *)(yb + (with_s ? 4 : 2));
    block_ints(type, b, q);
    int sumi = 0;
    const int zero = type == ORC_Q4_0 ? 8 : type == ORC_Q5_0 ? 16 : 0;
    for (int e = 0; e < 32; ++e) sumi += (q[e] - zero) * u[e];
    const float dx = orc_fp16_to_fp32(ld16(b)), dy = orc_fp16_to_fp32(ld16(yb));
    if (with_s) sumf += (dx * dy) * (float)sumi + orc_fp16_to_fp32(ld16(b + 2)) * orc_fp16_to_fp32(ld16(yb + 2));
    else sumf += (float)sumi * dx * dy;
  }
  return sumf;
}

void orc_matmul_cpu(int type, const void *W, int N, int K, const float *X, int B, float *out) {
  const size_t row_bytes = (size_t)(K / orc_block_size(type)) * orc_type_size(type);
  const int kq = orc_block_size(type) == 256;
  const int with_s = (type == ORC_Q4_1 || type == ORC_Q5_1);
  const size_t ybytes = kq ? (size_t)(K / 256) * 292 : (size_t)(K / 32) * (with_s ? 36 : 34);
  uint8_t *Y = malloc(ybytes * (size_t)B);
  for (int b = 0; b < B; ++b) {
    const float *x = X + (size_t)b * K;
    uint8_t *y = Y + ybytes * b;
    if (kq) orc_quantize_q8_K(x, y, K);
    else if (!with_s) quantize_legacy(ORC_Q8_0, x, y, K);
    else
      for (int ib = 0; ib < K / 32; ++ib) { /* ggml CPU q8_1: d = amax/127, s = d * sum(q) */
        float amax = 0;
        for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(x[ib * 32 + j]));
        float d = amax / 127.f, id = d ? 1.f / d : 0.f;
        int sum = 0;
        for (int j = 0; j < 32; ++j) { int8_t v = (int8_t)roundf(x[ib * 32 + j] * id); ((int8_t *)(y + ib * 36 + 4))[j] = v; sum += v; }
        st16(y + ib * 36, orc_fp32_to_fp16(d));
        st16(y + ib * 36 + 2, orc_fp32_to_fp16(d * (float)sum));
      }
  }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int b = 0; b < B; ++b) {
      const uint8_t *wrow = (const uint8_t *)W + n * row_bytes;
      out[(size_t)b * N + n] = kq ? dot_kquant_q8K(type, wrow, K, Y + ybytes * b) : dot_legacy_q8(type, wrow, K, Y + ybytes * b);
    }
  free(Y);
}

/* ------------------------------------------------------------------ glue ops */
void orc_rms_norm(const float *x, const float *w, float *out, int rows, int d, float eps) {
  for (int r = 0; r < rows; ++r) {
    double ss = 0;
    for (int i = 0; i < d; ++i) ss += (double)x[(size_t)r * d + i] * x[(size_t)r * d + i];
    const float inv = (float)(1.0 / sqrt(ss / d + (double)eps));
    for (int i = 0; i < d; ++i) out[(size_t)r * d + i] = x[(size_t)r * d + i] * inv * w[i];
  }
}

/* reference: mistralrs-quant/src/rotary/mod.rs:308-402 (CPU inner), kernels/rotary/rotary.cu:9-33.
 * interleaved (GPT-J, neox=0): pairs (2i, 2i+1); neox=1: pairs (i, i+rot/2) */
void orc_rope(float *x, const float *cos_t, const float *sin_t, const int32_t *positions, int tokens, int heads,
              int head_dim, int rot_dim, int neox) {
  const int half = rot_dim / 2;
  for (int t = 0; t < tokens; ++t) {
    const float *c = cos_t + (size_t)positions[t] * half, *s = sin_t + (size_t)positions[t] * half;
    for (int h = 0; h < heads; ++h) {
      float *v = x + ((size_t)t * heads + h) * head_dim;
      for (int i = 0; i < half; ++i) {
        const int a = neox ? i : 2 * i, b = neox ? i + half : 2 * i + 1;
        const float xa = v[a], xb = v[b];
        v[a] = xa * c[i] - xb * s[i];
        v[b] = xb * c[i] + xa * s[i];
      }
    }
  }
}

/* reference: mistralrs-quant/src/utils/ops.rs:2601-2620, kernels/mmvq_gguf/mmvq_gguf.cu:44-85 */
float orc_glu_act(float x, int act) {
  switch (act) {
  case 1: { const float x3 = x * x * x; return 0.5f * x * (1.0f + tanhf(0.7978845608f * (x + 0.044715f * x3))); }
  case 2: return fmaxf(x, 0.0f);
  case 3: return x * 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  case 4: return 1.0f / (1.0f + expf(-x));
  default: return x / (1.0f + expf(-x));
  }
}
void orc_fused_glu(const float *a, const float *b, float *out, int64_t n, int act) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_glu_act(a[i], act) * b[i];
}

void orc_attention(const float *q, const float *k, const float *v, float *out, int T, int S, int H, int KVH, int hd,
                   float scale, float softcap) {
  const int grp = H / KVH;
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < T; ++t)
    for (int h = 0; h < H; ++h) {
      const int kvh = h / grp, n = S - T + t + 1;
      double *p = malloc((size_t)n * sizeof(double));
      double mx = -INFINITY;
      const float *qv = q + ((size_t)t * H + h) * hd;
      for (int s = 0; s < n; ++s) {
        const float *kv = k + ((size_t)s * KVH + kvh) * hd;
        double d = 0;
        for (int i = 0; i < hd; ++i) d += (double)qv[i] * kv[i];
        d *= scale;
        if (softcap != 1.0f && softcap != 0.0f) d = tanh(d / softcap) * softcap;
        p[s] = d;
        if (d > mx) mx = d;
      }
      double den = 0;
      for (int s = 0; s < n; ++s) { p[s] = exp(p[s] - mx); den += p[s]; }
      float *o = out + ((size_t)t * H + h) * hd;
      for (int i = 0; i < hd; ++i) {
        double acc = 0;
        for (int s = 0; s < n; ++s) acc += p[s] * v[((size_t)s * KVH + kvh) * hd + i];
        o[i] = (float)(acc / den);
      }
      free(p);
    }
}
