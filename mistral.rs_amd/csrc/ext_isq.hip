// ext_isq.hip -- in-situ quantization (ISQ) of dense weights to GGML Q8_0 blocks on the GPU.
// Reference: `generate_isq!` (mistralrs-quant/src/utils/isq.rs:323-361) calls candle's QTensor::quantize(w, Q8_0) on the CPU and
// uploads the blocks; here the same per-block rule (GGML quantize_row_q8_0: d = amax/127, id = d ? 1/d : 0, q = round(x*id),
// d stored as f16) runs on the device where the dense weight already lives: 288 GB of HBM hold the bf16 source and the Q8_0
// result side by side, so there is no host round trip.  Output = standard Q8_0 bytes [N][K/32][34] consumed by the same GEMV /
// GEMM kernels as GGUF Q8_0 files.  `K % 32 != 0` is refused (the reference falls back to another dtype: isq.rs:249-287).
#include "common.cuh"
#include "gguf_blocks.cuh"

namespace mrs {

template <class T>
__global__ void __launch_bounds__(256) isq_q8_0_kernel(const T *__restrict__ w, uint8_t *__restrict__ out, size_t nblocks) {
  // one 32-lane half-wave per block: lane l holds value l of the block
  const size_t blk = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (blk >= nblocks) return;  // whole half-waves leave together
  const int l = threadIdx.x & 31;
  const float x = to_f<T>(w[blk * 32 + l]);
  float amax = fabsf(x);
#pragma unroll
  for (int m = 16; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
  const float d = amax / 127.0f;
  const float id = d != 0.0f ? 1.0f / d : 0.0f;
  uint8_t *b = out + blk * 34;
  ((int8_t *)(b + 2))[l] = (int8_t)roundf(x * id);
  if (l == 0) *(uint16_t *)b = float_to_half_bits(d);
}


// ------------------------------------------------------------------------------------------------ K-quants (Q4_K / Q5_K / Q6_K) + legacy 4/5-bit
// The reference runs candle's QTensor::quantize (= GGML's quantize_row_q{4,5,6}_K_ref / q{4,5}_{0,1}_ref) on the host cores for every ISQ'd
// tensor (utils/isq.rs:323-361): minutes for an 8B model.  Here the same search runs where the dense weight lives.  The arithmetic is GGML's,
// operation for operation in f32 with contraction off (make_qkx2_quants: 21 / 16 candidate scales per 32-weight sub-block with the weights
// av_x + |x|; make_qx_quants: 19 candidates per 16-weight sub-block), so the blocks are bit-identical to the oracle's restatement
// (oracle/ggml_oracle.c: quantize_q4_5_K, quantize_q6_K, quantize_legacy; tests/test_isq.py).  Mapping: ONE LANE PER SUB-BLOCK -- the search over
// a sub-block is sequential by definition -- 8 (Q4_K / Q5_K) or 16 (Q6_K) neighbouring lanes form a superblock; the cross-sub-block steps
// (max scale / max min, 6-bit scale packing, nibble / high-bit interleaving) are wave exchanges inside that lane group.

__device__ __forceinline__ int isq_nearest_int(float f) { return (int)rintf(f); }  // round half to even, as lrintf / ggml's magic-number trick

// make_qkx2_quants(32, nmax, x, w, .., rmin, rdelta, nstep, use_mad = false): only the returned scale and *the_min matter to the callers
// (the quants are recomputed after the 6-bit scale rounding), so L / Laux are not materialised: a candidate's quants are re-derived where needed.
template <int N = 32, bool MAD = false>
__device__ __forceinline__ float isq_make_qkx2(const float (&x)[N], const float (&w)[N], int nmax, float rmin, float rdelta, int nstep, float &the_min) {
  float mn = x[0], mx = x[0], sum_w = w[0], sum_x = sum_w * x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) {
    if (x[i] < mn) mn = x[i];
    if (x[i] > mx) mx = x[i];
    sum_w += w[i];
    sum_x += w[i] * x[i];
  }
  if (mn > 0) mn = 0;
  if (mx == mn) { the_min = -mn; return 0.f; }
  float iscale = nmax / (mx - mn), scale = 1 / iscale, best_mad = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int l = max(0, min(nmax, isq_nearest_int(iscale * (x[i] - mn))));
    float diff = scale * l + mn - x[i];
    diff = MAD ? fabsf(diff) : diff * diff;
    best_mad += w[i] * diff;
  }
  for (int is = 0; is <= nstep; ++is) {
    iscale = (rmin + rdelta * is + nmax) / (mx - mn);
    float sum_l = 0, sum_l2 = 0, sum_xl = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = max(0, min(nmax, isq_nearest_int(iscale * (x[i] - mn))));
      sum_l += w[i] * l;
      sum_l2 += w[i] * l * l;
      sum_xl += w[i] * l * x[i];
    }
    const float D = sum_w * sum_l2 - sum_l * sum_l;
    if (D > 0) {
      float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D;
      float this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;
      if (this_min > 0) { this_min = 0; this_scale = sum_xl / sum_l2; }
      float mad = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = max(0, min(nmax, isq_nearest_int(iscale * (x[i] - mn))));  // Laux[i]
        float diff = this_scale * l + this_min - x[i];
        diff = MAD ? fabsf(diff) : diff * diff;
        mad += w[i] * diff;
      }
      if (mad < best_mad) { best_mad = mad; scale = this_scale; mn = this_min; }  // the following candidates use the updated min, as GGML does
    }
  }
  the_min = -mn;
  return scale;
}

// Q4_K (FIVE = false, 144 B) / Q5_K (FIVE = true, 176 B): 8 lanes per superblock, lane j = sub-block j
// make_qp_quants(8, 63, x, L, qw) of GGML's importance-weighted quantizers: the 6-bit super-scale search over the eight sub-block scales (or mins)
template <int N = 8, int NMAX = 63>
__device__ __forceinline__ float isq_make_qp8(const float (&x)[N], int (&L)[N], const float (&qw)[N]) {
  constexpr int nmax = NMAX;
  float mx = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) if (x[i] > mx) mx = x[i];
  if (!(mx != 0.f)) {
#pragma unroll
    for (int i = 0; i < N; ++i) L[i] = 0;
    return 0.f;
  }
  float iscale = nmax / mx;
#pragma unroll
  for (int i = 0; i < N; ++i) L[i] = isq_nearest_int(iscale * x[i]);
  const float scale = 1 / iscale;
  float best_mse = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) { const float diff = x[i] - scale * L[i]; best_mse += qw[i] * diff * diff; }
  for (int is = -4; is <= 4; ++is) {
    if (is == 0) continue;
    const float iscale_is = (0.1f * is + nmax) / mx, scale_is = 1 / iscale_is;
    float mse = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = min(nmax, isq_nearest_int(iscale_is * x[i]));
      const float diff = x[i] - scale_is * l;
      mse += qw[i] * diff * diff;
    }
    if (mse < best_mse) { best_mse = mse; iscale = iscale_is; }
  }
  float sumlx = 0, suml2 = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int l = min(nmax, isq_nearest_int(iscale * x[i]));
    L[i] = l;
    sumlx += qw[i] * x[i] * l;
    suml2 += qw[i] * l * l;
  }
  for (int itry = 0; itry < 5; ++itry) {
    int n_changed = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float w = qw[i];
      float slx = sumlx - w * x[i] * L[i], sl2 = suml2 - w * L[i] * L[i];
      if (slx > 0 && sl2 > 0) {
        const int new_l = min(nmax, isq_nearest_int(x[i] * sl2 / slx));
        if (new_l != L[i]) {
          slx += w * x[i] * new_l;
          sl2 += w * new_l * new_l;
          if (slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = new_l; sumlx = slx; suml2 = sl2; ++n_changed; }
        }
      }
    }
    if (!n_changed) break;
  }
  return sumlx / suml2;
}

// IM: GGML's quantize_row_q{4,5}_K_impl with quant_weights qw[k] (one importance value per input column, shared by the rows): weights
// qw * sqrt(sigma2 + x^2) with sigma2 = 2 * sum(x^2) / 256 over the superblock, 37 candidate scales per sub-block, 6-bit scales by make_qp_quants.
template <class T, bool FIVE, bool IM = false>
__global__ void __launch_bounds__(256) isq_q45_k_kernel(const T *__restrict__ src, uint8_t *__restrict__ out, size_t nsuper, const float *__restrict__ qw = nullptr,
                                                        int sb_per_row = 0) {
  constexpr int NMAX = FIVE ? 31 : 15, TS = FIVE ? 176 : 144;
  const int lane = threadIdx.x & 63, j = lane & 7, base = lane & ~7;
  const size_t sb_raw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const bool live = sb_raw < nsuper;           // whole 8-lane groups are live or not; dead groups compute on the last superblock
  const size_t sb = live ? sb_raw : nsuper - 1;
  float x[32], w[32];
  const T *p = src + sb * 256 + (size_t)j * 32;
  float sum_x2 = 0;
#pragma unroll
  for (int l = 0; l < 32; ++l) { x[l] = to_f<T>(p[l]); sum_x2 += x[l] * x[l]; }
  float mn_j, sc_j;
  int all[8];
  uint16_t dbits, mbits;
  if constexpr (IM) {
    // sigma2 over the whole superblock in element order (every lane of the group walks the 256 values: load-time work, and the order is the definition)
    float tot = 0;
    const T *p0 = src + sb * 256;
    for (int l = 0; l < 256; ++l) { const float v = to_f<T>(p0[l]); tot += v * v; }
    const float sigma2 = 2 * tot / 256;
    const float *q = qw + (sb % (size_t)sb_per_row) * 256 + (size_t)j * 32;
    float sumw = 0;
#pragma unroll
    for (int l = 0; l < 32; ++l) { w[l] = q[l] * sqrtf(sigma2 + x[l] * x[l]); sumw += w[l]; }
    sc_j = isq_make_qkx2(x, w, NMAX, -0.9f, 0.05f, 36, mn_j);  // make_qkx3_quants with the weights given
    float scs[8], mns[8], sws[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) { scs[g] = __shfl(sc_j, base + g, 64); mns[g] = __shfl(mn_j, base + g, 64); sws[g] = __shfl(sumw, base + g, 64); }
    int Ls[8], Lm[8];
    const float d_block = isq_make_qp8(scs, Ls, sws), m_block = isq_make_qp8(mns, Lm, sws);
#pragma unroll
    for (int g = 0; g < 8; ++g) all[g] = (Ls[g] & 0xff) | ((Lm[g] & 0xff) << 8);
    dbits = float_to_half_bits(d_block);
    mbits = float_to_half_bits(m_block);
  } else {
    const float av_x = sqrtf(sum_x2 / 32);
#pragma unroll
    for (int l = 0; l < 32; ++l) w[l] = av_x + fabsf(x[l]);
    sc_j = FIVE ? isq_make_qkx2(x, w, 31, -0.5f, 0.1f, 15, mn_j) : isq_make_qkx2(x, w, 15, -1.f, 0.1f, 20, mn_j);
    float max_scale = sc_j > 0 ? sc_j : 0.f, max_min = mn_j > 0 ? mn_j : 0.f;  // `if (scales[j] > max_scale)` from +0: never -0, never NaN
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { max_scale = fmaxf(max_scale, __shfl_xor(max_scale, m, 64)); max_min = fmaxf(max_min, __shfl_xor(max_min, m, 64)); }
    const float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f, inv_min = max_min > 0 ? 63.f / max_min : 0.f;
    const int ls = min(63, isq_nearest_int(inv_scale * sc_j)), lm = min(63, isq_nearest_int(inv_min * mn_j));
    // header (lane 0 of the group): half d, half dmin, 12 bytes of 6-bit scales / mins (get_scale_min_k4 layout)
    const int mine = (ls & 0xff) | ((lm & 0xff) << 8);
#pragma unroll
    for (int g = 0; g < 8; ++g) all[g] = __shfl(mine, base + g, 64);
    dbits = float_to_half_bits(max_scale / 63.f);
    mbits = float_to_half_bits(max_min / 63.f);
  }
  uint8_t *y = out + sb * TS;
  uint8_t sc12[12];  // every lane builds the packed header (cheap) so that its own (sc, m) come out of the bytes exactly as a reader decodes them
#pragma unroll
  for (int g = 0; g < 12; ++g) sc12[g] = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int gls = all[g] & 0xff, glm = (all[g] >> 8) & 0xff;
    if (g < 4) { sc12[g] = (uint8_t)gls; sc12[g + 4] = (uint8_t)glm; }
    else {
      sc12[g + 4] = (uint8_t)((gls & 0xF) | ((glm & 0xF) << 4));
      sc12[g - 4] |= (uint8_t)((gls >> 4) << 6);
      sc12[g] |= (uint8_t)((glm >> 4) << 6);
    }
  }
  if (live && j == 0) {
    uint32_t hw[4];
    hw[0] = (uint32_t)dbits | ((uint32_t)mbits << 16);
#pragma unroll
    for (int g = 0; g < 3; ++g) hw[g + 1] = (uint32_t)sc12[4 * g] | ((uint32_t)sc12[4 * g + 1] << 8) | ((uint32_t)sc12[4 * g + 2] << 16) | ((uint32_t)sc12[4 * g + 3] << 24);
#pragma unroll
    for (int g = 0; g < 4; ++g) ((uint32_t *)y)[g] = hw[g];  // blocks are 16-byte aligned (144 / 176 B strides)
  }
  // final quants against the ROUNDED scales (get_scale_min_k4 of the packed bytes): l = clamp(rne((x + dmin * m) / (d * sc)), 0, nmax); d * sc == 0 -> 0
  int ksc = 0, km = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g == j) {
      if (g < 4) { ksc = sc12[g] & 63; km = sc12[g + 4] & 63; }
      else { ksc = (sc12[g + 4] & 0xF) | ((sc12[g - 4] >> 6) << 4); km = (sc12[g + 4] >> 4) | ((sc12[g] >> 6) << 4); }
    }
  }
  const float dd = half_bits_to_float(dbits), dmin = half_bits_to_float(mbits);
  const float d = dd * (float)ksc, dm = dmin * (float)km;
  uint32_t lo[8], hi[8];  // 32 quants: low nibbles one per byte, bit 4 one per byte
#pragma unroll
  for (int g = 0; g < 8; ++g) { lo[g] = 0; hi[g] = 0; }
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    int l = 0;
    if (d != 0.f) l = max(0, min(NMAX, isq_nearest_int((x[i] + dm) / d)));
    lo[i >> 2] |= (uint32_t)(l & 0xF) << (8 * (i & 3));
    hi[i >> 2] |= (uint32_t)(l >> 4) << (8 * (i & 3));
  }
  // qs: byte l of pair c = L[2c][l] | L[2c+1][l] << 4: lanes 2c / 2c+1 swap their nibble words; the even lane writes bytes 0..15, the odd 16..31
  uint8_t *qs = y + (FIVE ? 48 : 16) + (j >> 1) * 32;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const uint32_t other = (uint32_t)__shfl_xor((int)lo[g], 1, 64);
    const uint32_t word = (j & 1) ? (other | (lo[g] << 4)) : (lo[g] | (other << 4));
    const bool mine_to_write = (j & 1) ? g >= 4 : g < 4;
    if (live && mine_to_write) ((uint32_t *)qs)[g] = word;
  }
  if constexpr (FIVE) {  // qh[l] bit j = bit 4 of L[j][l]: OR over the 8 lanes, lane g writes word g
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      uint32_t v = hi[g] << j;
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) v |= (uint32_t)__shfl_xor((int)v, m, 64);
      if (live && j == g) ((uint32_t *)(y + 16))[g] = v;
    }
  }
}

// Q6_K (210 B, 2-byte aligned): 16 lanes per superblock, lane ib = 16-weight sub-block ib
// IM: GGML's quantize_row_q6_K_impl with quant_weights -- the importance values themselves weight the 19-candidate scale search
template <class T, bool IM = false>
__global__ void __launch_bounds__(256) isq_q6_k_kernel(const T *__restrict__ src, uint8_t *__restrict__ out, size_t nsuper, const float *__restrict__ qw = nullptr,
                                                       int sb_per_row = 0) {
  const int lane = threadIdx.x & 63, ib = lane & 15, base = lane & ~15;
  const size_t sb_raw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = sb_raw < nsuper;
  const size_t sb = live ? sb_raw : nsuper - 1;
  float x[16];
  const T *p = src + sb * 256 + (size_t)ib * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = to_f<T>(p[i]);
  float wq[16];  // rmse_type 1: x^2, or the importance values
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if constexpr (IM) wq[i] = qw[(sb % (size_t)sb_per_row) * 256 + (size_t)ib * 16 + i];
    else wq[i] = x[i] * x[i];
  }
  // make_qx_quants(16, 32, x, L, 1, weights)
  int L[16];
  float scale;
  {
    float mx = 0, amax = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float ax = fabsf(x[i]); if (ax > amax) { amax = ax; mx = x[i]; } }
    if (amax < 1e-15f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L[i] = 0;
      scale = 0.f;
    } else {
      float iscale = -32 / mx, sumlx = 0, suml2 = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = max(-32, min(31, isq_nearest_int(iscale * x[i])));
        L[i] = l + 32;
        const float w = wq[i];
        sumlx += w * x[i] * l;
        suml2 += w * l * l;
      }
      scale = suml2 ? sumlx / suml2 : 0.0f;
      float best = scale * sumlx;
      for (int is = -9; is <= 9; ++is) {
        if (is == 0) continue;
        iscale = -(32 + 0.1f * is) / mx;
        sumlx = suml2 = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int l = max(-32, min(31, isq_nearest_int(iscale * x[i])));
          const float w = wq[i];
          sumlx += w * x[i] * l;
          suml2 += w * l * l;
        }
        if (suml2 > 0 && sumlx * sumlx > best * suml2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) L[i] = 32 + max(-32, min(31, isq_nearest_int(iscale * x[i])));
          scale = sumlx / suml2;
          best = scale * sumlx;
        }
      }
    }
  }
  // the scale of largest magnitude, first sub-block on ties (`if (fabsf(s) > max_abs)`)
  float best_abs = fabsf(scale), best_s = scale;
  int best_i = ib;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const float oa = __shfl_xor(best_abs, m, 64), os = __shfl_xor(best_s, m, 64);
    const int oi = __shfl_xor(best_i, m, 64);
    if (oa > best_abs || (oa == best_abs && oi < best_i)) { best_abs = oa; best_s = os; best_i = oi; }
  }
  uint8_t *y = out + sb * 210;
  const bool zero_block = best_abs < 1e-15f;  // all-zero superblock: 210 zero bytes.  No early return: the lane exchanges below stay wave-uniform
  const float iscale = -128.f / best_s;
  const uint16_t dbits = zero_block ? (uint16_t)0 : float_to_half_bits(1 / iscale);
  const int sc = zero_block ? 0 : min(127, isq_nearest_int(iscale * scale));
  const float d = half_bits_to_float(dbits) * (float)(int8_t)sc;
  if (zero_block) {
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = 0;
  } else if (d != 0.f) {
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = max(-32, min(31, isq_nearest_int(x[i] / d))) + 32;
  }
  if (live) {
    ((int8_t *)(y + 192))[ib] = (int8_t)sc;
    if (ib == 0) *(uint16_t *)(y + 208) = dbits;
  }
  // interleave: half h = ib / 8, quarter qt = (ib % 8) / 2, parity par = ib % 2 (which 16 of the quarter's 32 columns)
  //   ql[h*64 + (qt&1)*32 + par*16 + i] = lo4(L of quarter qt&1) | lo4(L of quarter (qt&1)+2) << 4   <- lanes ib and ib + 4
  //   qh[h*32 + par*16 + i]            = sum over qt of (L >> 4) << 2 qt                               <- lanes ib, ib+2, ib+4, ib+6
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { lo[g] = 0; hi[g] = 0; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo[i >> 2] |= (uint32_t)(L[i] & 0xF) << (8 * (i & 3));
    hi[i >> 2] |= (uint32_t)(L[i] >> 4) << (8 * (i & 3));
  }
  const int h = ib >> 3, qt = (ib & 7) >> 1, par = ib & 1;
  uint16_t *ql = (uint16_t *)(y + h * 64 + (qt & 1) * 32 + par * 16), *qh = (uint16_t *)(y + 128 + h * 32 + par * 16);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t up = (uint32_t)__shfl((int)lo[g], base + ((ib + 4) & 15), 64);   // quarter qt + 2 (meaningful for qt < 2)
    const uint32_t h1 = (uint32_t)__shfl((int)hi[g], base + ((ib + 2) & 15), 64), h2 = (uint32_t)__shfl((int)hi[g], base + ((ib + 4) & 15), 64),
                   h3 = (uint32_t)__shfl((int)hi[g], base + ((ib + 6) & 15), 64);
    if (live && qt < 2) { const uint32_t v = lo[g] | (up << 4); ql[2 * g] = (uint16_t)v; ql[2 * g + 1] = (uint16_t)(v >> 16); }
    if (live && qt == 0) { const uint32_t v = hi[g] | (h1 << 2) | (h2 << 4) | (h3 << 6); qh[2 * g] = (uint16_t)v; qh[2 * g + 1] = (uint16_t)(v >> 16); }
  }
}

// make_qx_quants(16, NMAX, x, L, rmse_type 1, weights) of GGML: L = quant + NMAX, returns the scale (19 candidate scales)
template <int NMAX>
__device__ __forceinline__ float isq_make_qx16(const float (&x)[16], const float (&wq)[16], int (&L)[16]) {
  float mx = 0, amax = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const float ax = fabsf(x[i]); if (ax > amax) { amax = ax; mx = x[i]; } }
  if (amax < 1e-15f) {
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = 0;
    return 0.f;
  }
  float iscale = -NMAX / mx, sumlx = 0, suml2 = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int l = max(-NMAX, min(NMAX - 1, isq_nearest_int(iscale * x[i])));
    L[i] = l + NMAX;
    sumlx += wq[i] * x[i] * l;
    suml2 += wq[i] * l * l;
  }
  float scale = suml2 ? sumlx / suml2 : 0.0f;
  float best = scale * sumlx;
  for (int is = -9; is <= 9; ++is) {
    if (is == 0) continue;
    iscale = -(NMAX + 0.1f * is) / mx;
    sumlx = suml2 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int l = max(-NMAX, min(NMAX - 1, isq_nearest_int(iscale * x[i])));
      sumlx += wq[i] * x[i] * l;
      suml2 += wq[i] * l * l;
    }
    if (suml2 > 0 && sumlx * sumlx > best * suml2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L[i] = NMAX + max(-NMAX, min(NMAX - 1, isq_nearest_int(iscale * x[i])));
      scale = sumlx / suml2;
      best = scale * sumlx;
    }
  }
  return scale;
}

// Q2_K (84 B: 16 x {4-bit scale | 4-bit min << 4}, 64 B of 2-bit quants, half d, half dmin) -- quantize_row_q2_K_ref: 16 lanes per superblock,
// lane ib = 16-weight sub-block; make_qkx2_quants(16, 3, x, |x|, .., -0.5, 0.1, 15, use_mad = true).
// IM: quantize_row_q2_K_impl -- weights qw * sqrt(sigma2 + x^2) with sigma2 = sum(x^2) / 256, make_qkx3_quants(16, 3, .., -0.9, 0.05, 36), 4-bit scales and
// minimums by make_qp_quants(16, 15, .., sw)
template <class T, bool IM = false>
__global__ void __launch_bounds__(256) isq_q2_k_kernel(const T *__restrict__ src, uint8_t *__restrict__ out, size_t nsuper, const float *__restrict__ qw = nullptr,
                                                       int sb_per_row = 0) {
  const int lane = threadIdx.x & 63, ib = lane & 15, base = lane & ~15;
  const size_t sb_raw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = sb_raw < nsuper;
  const size_t sb = live ? sb_raw : nsuper - 1;
  float x[16], w[16];
  const T *p = src + sb * 256 + (size_t)ib * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = to_f<T>(p[i]); w[i] = fabsf(x[i]); }
  float mn, sc;
  int ls = 0, lm = 0;
  uint16_t dbits = float_to_half_bits(0.f), mbits = float_to_half_bits(0.f);
  if constexpr (IM) {
    float tot = 0;  // sigma2 over the superblock in element order (every lane of the group walks the 256 values)
    const T *p0 = src + sb * 256;
    for (int l = 0; l < 256; ++l) { const float v = to_f<T>(p0[l]); tot += v * v; }
    const float sigma2 = tot / 256;
    const float *q = qw + (sb % (size_t)sb_per_row) * 256 + (size_t)ib * 16;
    float sumw = 0;
#pragma unroll
    for (int l = 0; l < 16; ++l) { w[l] = q[l] * sqrtf(sigma2 + x[l] * x[l]); sumw += w[l]; }
    sc = isq_make_qkx2<16, false>(x, w, 3, -0.9f, 0.05f, 36, mn);
    float scs[16], mns[16], sws[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) { scs[g] = __shfl(sc, base + g, 64); mns[g] = __shfl(mn, base + g, 64); sws[g] = __shfl(sumw, base + g, 64); }
    int Ls[16], Lm[16];
    const float dmv = isq_make_qp8<16, 15>(scs, Ls, sws), mmv = isq_make_qp8<16, 15>(mns, Lm, sws);
    dbits = float_to_half_bits(dmv);
    mbits = float_to_half_bits(mmv);
#pragma unroll
    for (int g = 0; g < 16; ++g) if (g == ib) { ls = Ls[g]; lm = Lm[g]; }
  } else {
    sc = isq_make_qkx2<16, true>(x, w, 3, -0.5f, 0.1f, 15, mn);
    float max_scale = sc > 0 ? sc : 0.f, max_min = mn > 0 ? mn : 0.f;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { max_scale = fmaxf(max_scale, __shfl_xor(max_scale, m, 64)); max_min = fmaxf(max_min, __shfl_xor(max_min, m, 64)); }
    if (max_scale > 0) { ls = isq_nearest_int((15.f / max_scale) * sc); dbits = float_to_half_bits(max_scale / 15.f); }
    if (max_min > 0) { lm = isq_nearest_int((15.f / max_min) * mn); mbits = float_to_half_bits(max_min / 15.f); }
  }
  const uint8_t scb = (uint8_t)((uint8_t)ls | (uint8_t)(lm << 4));  // exactly the byte a reader decodes: `scales[j] = l; scales[j] |= l << 4`
  const float d = half_bits_to_float(dbits) * (float)(scb & 0xF), dm = half_bits_to_float(mbits) * (float)(scb >> 4);
  uint32_t lo[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int l = 0;
    if (d != 0.f) l = max(0, min(3, isq_nearest_int((x[i] + dm) / d)));
    lo[i >> 2] |= (uint32_t)l << (8 * (i & 3));
  }
  uint8_t *y = out + sb * 84;
  if (live) {
    y[ib] = scb;
    if (ib == 0) { *(uint16_t *)(y + 80) = dbits; *(uint16_t *)(y + 82) = mbits; }
  }
  // qs[32 h + l] = L[q = 0][l] | L[1][l] << 2 | L[2][l] << 4 | L[3][l] << 6 over the quarters of half h: lanes ib, ib + 2, ib + 4, ib + 6
  const int qt = (ib >> 1) & 3, h = ib >> 3, par = ib & 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t q1 = (uint32_t)__shfl((int)lo[g], base + ((ib + 2) & 15), 64), q2 = (uint32_t)__shfl((int)lo[g], base + ((ib + 4) & 15), 64),
                   q3 = (uint32_t)__shfl((int)lo[g], base + ((ib + 6) & 15), 64);
    if (live && qt == 0) ((uint32_t *)(y + 16 + 32 * h + 16 * par))[g] = lo[g] | (q1 << 2) | (q2 << 4) | (q3 << 6);
  }
}

// Q3_K (110 B: 32 B high-bit mask, 64 B of 2-bit quants, 12 B of 6-bit scales, half d) -- quantize_row_q3_K_ref: make_q3_quants(16, 4, x, L, true)
// per sub-block (first guess + up to 5 refinement sweeps), super-scale -32 / (scale of largest magnitude).
// IM: quantize_row_q3_K_impl -- make_qx_quants(16, 4, x, L, 1, qw * sqrt(sigma2 + x^2)) with sigma2 = 2 sum(x^2) / 256 per sub-block, then the sixteen scales
// by make_qx_quants(16, 32, scales, Ls, 1, sw) (their 6-bit codes and the super-scale together)
template <class T, bool IM = false>
__global__ void __launch_bounds__(256) isq_q3_k_kernel(const T *__restrict__ src, uint8_t *__restrict__ out, size_t nsuper, const float *__restrict__ qw = nullptr,
                                                       int sb_per_row = 0) {
  const int lane = threadIdx.x & 63, ib = lane & 15, base = lane & ~15;
  const size_t sb_raw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = sb_raw < nsuper;
  const size_t sb = live ? sb_raw : nsuper - 1;
  float x[16];
  const T *p = src + sb * 256 + (size_t)ib * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = to_f<T>(p[i]);
  int L[16];
  float scale;
  float sumw_im = 0;
  if constexpr (IM) {
    float tot = 0;
    const T *p0 = src + sb * 256;
    for (int l = 0; l < 256; ++l) { const float v = to_f<T>(p0[l]); tot += v * v; }
    const float sigma2 = 2 * tot / 256;
    const float *q = qw + (sb % (size_t)sb_per_row) * 256 + (size_t)ib * 16;
    float wq[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) { wq[l] = q[l] * sqrtf(sigma2 + x[l] * x[l]); sumw_im += wq[l]; }
    scale = isq_make_qx16<4>(x, wq, L);
  } else {
    constexpr int nmax = 4;
    float mx = 0, amax = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float ax = fabsf(x[i]); if (ax > amax) { amax = ax; mx = x[i]; } }
    if (amax < 1e-15f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L[i] = 0;
      scale = 0.f;
    } else {
      const float iscale = -nmax / mx;
      float sumlx = 0, suml2 = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = max(-nmax, min(nmax - 1, isq_nearest_int(iscale * x[i])));
        L[i] = l;
        const float w = x[i] * x[i];
        sumlx += w * x[i] * l;
        suml2 += w * l * l;
      }
      for (int itry = 0; itry < 5; ++itry) {
        int n_changed = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float w = x[i] * x[i];
          float slx = sumlx - w * x[i] * L[i];
          if (slx > 0) {
            float sl2 = suml2 - w * L[i] * L[i];
            const int new_l = max(-nmax, min(nmax - 1, isq_nearest_int(x[i] * sl2 / slx)));
            if (new_l != L[i]) {
              slx += w * x[i] * new_l;
              sl2 += w * new_l * new_l;
              if (sl2 > 0 && slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = new_l; sumlx = slx; suml2 = sl2; ++n_changed; }
            }
          }
        }
        if (!n_changed) break;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) L[i] += nmax;
      scale = sumlx / suml2;
    }
  }
  // the scale of largest magnitude, first sub-block on ties (`if (scale > amax)` over |scale|)
  float best_abs = fabsf(scale), best_s = scale;
  int best_i = ib;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const float oa = __shfl_xor(best_abs, m, 64), os = __shfl_xor(best_s, m, 64);
    const int oi = __shfl_xor(best_i, m, 64);
    if (oa > best_abs || (oa == best_abs && oi < best_i)) { best_abs = oa; best_s = os; best_i = oi; }
  }
  int l6;
  uint16_t dbits;
  if constexpr (IM) {
    float scs[16], sws[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) { scs[g] = __shfl(scale, base + g, 64); sws[g] = __shfl(sumw_im, base + g, 64); }
    int Ls[16];
    const float d_block = isq_make_qx16<32>(scs, sws, Ls);  // codes already carry the + 32
    dbits = float_to_half_bits(d_block);
    l6 = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) if (g == ib) l6 = Ls[g];
  } else {
    const bool any = best_abs > 0.f;  // `if (max_scale)`: max_scale is only ever set to a scale whose magnitude exceeded 0
    const float iscale = any ? -32.f / best_s : 0.f;
    l6 = any ? max(-32, min(31, isq_nearest_int(iscale * scale))) + 32 : 0;  // 6-bit code; all twelve bytes stay 0 without a scale
    dbits = any ? float_to_half_bits(1 / iscale) : float_to_half_bits(0.f);
  }
  int all[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) all[g] = __shfl(l6, base + g, 64);
  uint8_t *y = out + sb * 110;
  if (live && ib < 12) {
    uint8_t b;
    if (ib < 8) b = (uint8_t)((all[ib] & 0xF) | ((all[ib + 8] & 0xF) << 4));
    else { const int m = ib - 8; b = (uint8_t)((all[m] >> 4) | ((all[m + 4] >> 4) << 2) | ((all[m + 8] >> 4) << 4) | ((all[m + 12] >> 4) << 6)); }
    y[96 + ib] = b;
  }
  if (live && ib == 0) *(uint16_t *)(y + 108) = dbits;
  const float d = half_bits_to_float(dbits) * (float)(l6 - 32);  // the reader's 6-bit scale minus 32 (-32 for an all-zero block: d = -0 -> the first quants stay)
  if (d != 0.f) {
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = max(-4, min(3, isq_nearest_int(x[i] / d))) + 4;
  }
  // high bits: element 16 ib + i -> hmask[16 (ib & 1) + i] bit ib >> 1; low two bits as Q2_K
  uint32_t lo[4] = {0, 0, 0, 0}, hb = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int l = L[i];
    if (l > 3) { hb |= 1u << i; l -= 4; }
    lo[i >> 2] |= (uint32_t)l << (8 * (i & 3));
  }
  const int qt = (ib >> 1) & 3, h = ib >> 3, par = ib & 1;
  uint32_t hm[4] = {0, 0, 0, 0};  // byte i (of this parity's 16) = sum over the 8 lanes of the parity of bit i << (their ib >> 1)
#pragma unroll
  for (int kq = 0; kq < 8; ++kq) {
    const uint32_t o = (uint32_t)__shfl((int)hb, base + 2 * kq + par, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) hm[i >> 2] |= ((o >> i) & 1u) << (8 * (i & 3) + kq);
  }
  if (live && ib < 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) { ((uint16_t *)(y + 16 * par))[2 * g] = (uint16_t)hm[g]; ((uint16_t *)(y + 16 * par))[2 * g + 1] = (uint16_t)(hm[g] >> 16); }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t q1 = (uint32_t)__shfl((int)lo[g], base + ((ib + 2) & 15), 64), q2 = (uint32_t)__shfl((int)lo[g], base + ((ib + 4) & 15), 64),
                   q3 = (uint32_t)__shfl((int)lo[g], base + ((ib + 6) & 15), 64);
    if (live && qt == 0) {
      const uint32_t v = lo[g] | (q1 << 2) | (q2 << 4) | (q3 << 6);
      uint16_t *q = (uint16_t *)(y + 32 + 32 * h + 16 * par);  // 110-byte blocks: 2-byte aligned
      q[2 * g] = (uint16_t)v; q[2 * g + 1] = (uint16_t)(v >> 16);
    }
  }
}

// Q4_0 / Q5_0 / Q4_1 / Q5_1: one thread per 32-weight block (quantize_row_q{4,5}_{0,1}_ref)
template <class T, int TYPE>
__global__ void __launch_bounds__(256) isq_legacy_kernel(const T *__restrict__ src, uint8_t *__restrict__ out, size_t nblocks) {
  const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (blk >= nblocks) return;
  constexpr bool SYM = TYPE == 2 || TYPE == 6, FIVEB = TYPE == 6 || TYPE == 7;
  constexpr int TS = TYPE == 2 ? 18 : TYPE == 3 ? 20 : TYPE == 6 ? 22 : 24;
  float x[32];
  const T *p = src + blk * 32;
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = to_f<T>(p[i]);
  uint8_t *y = out + blk * TS;
  uint8_t qs[16];
  uint32_t qh = 0;
  if constexpr (SYM) {
    constexpr int HR = FIVEB ? 16 : 8, TOP = 2 * HR - 1;
    float amax = 0, mx = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) if (fabsf(x[i]) > amax) { amax = fabsf(x[i]); mx = x[i]; }
    // an all-zero block gives d = 0 / -HR = -0 in GGML (sign bit set in the stored f16); spelled out because hipcc folds the division
    // by a constant into a multiplication sequence that loses the sign of zero
    const float d = mx == 0.f ? -0.0f : mx / -(float)HR, id = d ? 1.f / d : 0.f;
    *(uint16_t *)y = float_to_half_bits(d);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int x0 = min(TOP, (int)(int8_t)(x[i] * id + (HR + 0.5f))), x1 = min(TOP, (int)(int8_t)(x[i + 16] * id + (HR + 0.5f)));
      qs[i] = (uint8_t)((x0 & 0xF) | ((x1 & 0xF) << 4));
      qh |= (uint32_t)((x0 & 0x10) >> 4) << i;
      qh |= (uint32_t)((x1 & 0x10) >> 4) << (i + 16);
    }
  } else {
    constexpr int TOP = FIVEB ? 31 : 15;
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }  // comparisons, not v_min / v_max: the first of +0 / -0 stays, as in GGML
    const float d = (mx - mn) / TOP, id = d ? 1.f / d : 0.f;
    *(uint16_t *)y = float_to_half_bits(d);
    *(uint16_t *)(y + 2) = float_to_half_bits(mn);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int x0 = min(TOP, (int)(uint8_t)((x[i] - mn) * id + 0.5f)), x1 = min(TOP, (int)(uint8_t)((x[i + 16] - mn) * id + 0.5f));
      qs[i] = (uint8_t)((x0 & 0xF) | ((x1 & 0xF) << 4));
      qh |= (uint32_t)((x0 & 0x10) >> 4) << i;
      qh |= (uint32_t)((x1 & 0x10) >> 4) << (i + 16);
    }
  }
  constexpr int HDR = SYM ? 2 : 4;
  if constexpr (FIVEB) { *(uint16_t *)(y + HDR) = (uint16_t)qh; *(uint16_t *)(y + HDR + 2) = (uint16_t)(qh >> 16); }
  uint16_t *q16 = (uint16_t *)(y + HDR + (FIVEB ? 4 : 0));
#pragma unroll
  for (int i = 0; i < 8; ++i) q16[i] = (uint16_t)(qs[2 * i] | (qs[2 * i + 1] << 8));
}

template <class T> static int isq_dispatch(const T *src, uint8_t *dst, size_t n, int type, hipStream_t s) {
  const dim3 block(256);
  switch (type) {
  case 8: { const size_t nb = n / 32; hipLaunchKernelGGL(isq_q8_0_kernel<T>, dim3((unsigned)((nb + 7) / 8)), block, 0, s, src, dst, nb); return 0; }
  case 2: { const size_t nb = n / 32; hipLaunchKernelGGL((isq_legacy_kernel<T, 2>), dim3((unsigned)((nb + 255) / 256)), block, 0, s, src, dst, nb); return 0; }
  case 3: { const size_t nb = n / 32; hipLaunchKernelGGL((isq_legacy_kernel<T, 3>), dim3((unsigned)((nb + 255) / 256)), block, 0, s, src, dst, nb); return 0; }
  case 6: { const size_t nb = n / 32; hipLaunchKernelGGL((isq_legacy_kernel<T, 6>), dim3((unsigned)((nb + 255) / 256)), block, 0, s, src, dst, nb); return 0; }
  case 7: { const size_t nb = n / 32; hipLaunchKernelGGL((isq_legacy_kernel<T, 7>), dim3((unsigned)((nb + 255) / 256)), block, 0, s, src, dst, nb); return 0; }
  case 10: { const size_t nb = n / 256; hipLaunchKernelGGL((isq_q2_k_kernel<T, false>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, (const float *)nullptr, 0); return 0; }
  case 11: { const size_t nb = n / 256; hipLaunchKernelGGL((isq_q3_k_kernel<T, false>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, (const float *)nullptr, 0); return 0; }
  case 12: { const size_t nb = n / 256; hipLaunchKernelGGL((isq_q45_k_kernel<T, false, false>), dim3((unsigned)((nb + 31) / 32)), block, 0, s, src, dst, nb, (const float *)nullptr, 0); return 0; }
  case 13: { const size_t nb = n / 256; hipLaunchKernelGGL((isq_q45_k_kernel<T, true, false>), dim3((unsigned)((nb + 31) / 32)), block, 0, s, src, dst, nb, (const float *)nullptr, 0); return 0; }
  case 14: { const size_t nb = n / 256; hipLaunchKernelGGL((isq_q6_k_kernel<T, false>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, (const float *)nullptr, 0); return 0; }
  default: return -1;
  }
}

template <class T> static int isq_dispatch_imatrix(const T *src, uint8_t *dst, size_t nrows, int k, int type, const float *qw, hipStream_t s) {
  const dim3 block(256);
  const size_t nb = nrows * (size_t)(k / 256);
  const int spr = k / 256;
  switch (type) {
  case 10: hipLaunchKernelGGL((isq_q2_k_kernel<T, true>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, qw, spr); return 0;
  case 11: hipLaunchKernelGGL((isq_q3_k_kernel<T, true>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, qw, spr); return 0;
  case 12: hipLaunchKernelGGL((isq_q45_k_kernel<T, false, true>), dim3((unsigned)((nb + 31) / 32)), block, 0, s, src, dst, nb, qw, spr); return 0;
  case 13: hipLaunchKernelGGL((isq_q45_k_kernel<T, true, true>), dim3((unsigned)((nb + 31) / 32)), block, 0, s, src, dst, nb, qw, spr); return 0;
  case 14: hipLaunchKernelGGL((isq_q6_k_kernel<T, true>), dim3((unsigned)((nb + 15) / 16)), block, 0, s, src, dst, nb, qw, spr); return 0;
  default: return -1;
  }
}

// ------------------------------------------------------------------------------------------------ dequantize (QuantMethod::dequantize_w)
// One lane per 32-weight slice (gguf_blocks.cuh: two 16-weight runs, w = s*q - o in f32, the arithmetic of the in-tree format spec
// marlin_gguf_affine_repack.cu:218-278 get_affine_params); a wave reads consecutive 16-byte pieces of the packed row and writes two 64-byte
// spans per lane.  HBM-bound: type_size/blk bytes in, sizeof(OUT) bytes out per weight.
__device__ __forceinline__ int byte_of(int4 v, int i) { return (int)(int8_t)(((const uint32_t *)&v)[i >> 2] >> (8 * (i & 3))); }

template <int TYPE, class OUT>
__global__ void __launch_bounds__(256) dequantize_kernel(const uint8_t *__restrict__ w, OUT *__restrict__ out, int64_t nrows, int K) {
  const int nslices = K / 32;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nrows * nslices) return;
  const int64_t row = gid / nslices;
  const int s = (int)(gid % nslices);
  const Slice sl = load_slice<TYPE>(w + row * (int64_t)(K / Fmt<TYPE>::BLK) * Fmt<TYPE>::TS, s);
  int ra, rb;
  slice_runs<TYPE>(s, ra, rb);
  OUT *oa = out + row * K + (int64_t)ra * 16, *ob = out + row * K + (int64_t)rb * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    oa[i] = from_f<OUT>(sl.sa * (float)byte_of(sl.qa, i) - sl.oa);
    ob[i] = from_f<OUT>(sl.sb * (float)byte_of(sl.qb, i) - sl.ob);
  }
}

template <class OUT> static int dequantize_dispatch(const uint8_t *w, OUT *out, int64_t nrows, int K, int type, hipStream_t s) {
  const int64_t total = nrows * (K / 32);
  if (total <= 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define MRS_DQ(T) case T: hipLaunchKernelGGL((dequantize_kernel<T, OUT>), grid, block, 0, s, w, out, nrows, K); return 0;
  switch (type) {
    MRS_DQ(T_Q4_0) MRS_DQ(T_Q4_1) MRS_DQ(T_Q5_0) MRS_DQ(T_Q5_1) MRS_DQ(T_Q8_0) MRS_DQ(T_Q2_K) MRS_DQ(T_Q3_K) MRS_DQ(T_Q4_K) MRS_DQ(T_Q5_K) MRS_DQ(T_Q6_K)
  default: return -1;
  }
#undef MRS_DQ
}

// ------------------------------------------------------------------------------------------------ imatrix statistics
// ImatrixLayerStats::process / process_routed (mistralrs-quant/src/imatrix.rs:73-135): per input column, the sum of squares of the activations a
// layer has seen (`inp.sqr().sum(0)` added to the accumulator; routed layers scatter every (token, slot) row into its expert's accumulator with
// `index_add`).  One thread per column walks the rows in order, so the f32 sums do not depend on the launch geometry.
template <class T>
__global__ void __launch_bounds__(256) imatrix_dense_kernel(const T *__restrict__ x, float *__restrict__ accum, long long rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.0f;
  for (long long r = 0; r < rows; ++r) { const float v = to_f<T>(x[r * cols + c]); s += v * v; }
  accum[c] = accum[c] + s;
}
// x: [n][in] (a token's row goes to all k of its experts) or [n * k][in] (per_slot: row t * k + s goes to ids[t][s] only); ids: [n * k]
template <class T>
__global__ void __launch_bounds__(256) imatrix_routed_kernel(const T *__restrict__ x, const uint32_t *__restrict__ ids, float *__restrict__ accum,
                                                             float *__restrict__ counts, int n, int k, int cols, int per_slot, int num_experts) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    for (int s = 0; s < n * k; ++s) {
      const uint32_t e = ids[s];
      if (e >= (uint32_t)num_experts) continue;  // candle's index_add would refuse the id; nothing is written here
      const float v = to_f<T>(x[(size_t)(per_slot ? s : s / k) * cols + c]);
      accum[(size_t)e * cols + c] += v * v;
    }
  }
  if (c == 0)
    for (int s = 0; s < n * k; ++s)
      if (ids[s] < (uint32_t)num_experts) counts[ids[s]] += 1.0f;
}

}  // namespace mrs

// src: dense [N*K] elements, dtype 0 = f32, 1 = f16, 30 = bf16 (ggml ids); dst: N*K/32*34 bytes.  Returns 0 / -1.
extern "C" int mrs_isq_quantize_q8_0(const void *src, int src_dtype, void *dst, long long n_elements, void *stream) {
  if (n_elements <= 0) return 0;
  if (n_elements % 32) return -1;
  const size_t nb = (size_t)n_elements / 32;
  const dim3 grid((unsigned)((nb + 7) / 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
  case 0: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<float>, grid, block, 0, s, (const float *)src, (uint8_t *)dst, nb); return 0;
  case 1: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<mrs::f16_t>, grid, block, 0, s, (const mrs::f16_t *)src, (uint8_t *)dst, nb); return 0;
  case 30: hipLaunchKernelGGL(mrs::isq_q8_0_kernel<mrs::bf16_t>, grid, block, 0, s, (const mrs::bf16_t *)src, (uint8_t *)dst, nb); return 0;
  default: return -1;
  }
}

// ISQ to any GGML target of `generate_isq!` that the GGUF kernels read: ggml_type 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0, 10 Q2_K, 11 Q3_K,
// 12 Q4_K, 13 Q5_K, 14 Q6_K.  src dtype as above.  Returns 0, -1 for an unknown dtype / type or when n_elements is not a multiple of the block size (the
// reference then falls back to another dtype: utils/isq.rs:249-287).  Blocks are bit-identical to GGML's reference quantizers.
extern "C" int mrs_isq_quantize(const void *src, int src_dtype, void *dst, long long n_elements, int ggml_type, void *stream) {
  if (n_elements <= 0) return 0;
  const int blk = (ggml_type >= 10 && ggml_type <= 14) ? 256 : 32;
  if (n_elements % blk) return -1;
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
  case 0: return mrs::isq_dispatch<float>((const float *)src, (uint8_t *)dst, (size_t)n_elements, ggml_type, s);
  case 1: return mrs::isq_dispatch<mrs::f16_t>((const mrs::f16_t *)src, (uint8_t *)dst, (size_t)n_elements, ggml_type, s);
  case 30: return mrs::isq_dispatch<mrs::bf16_t>((const mrs::bf16_t *)src, (uint8_t *)dst, (size_t)n_elements, ggml_type, s);
  default: return -1;
  }
}

// Importance-weighted ISQ (QTensor::quantize_imatrix; call sites gguf/mod.rs:238-252, utils/isq.rs generate_isq_imatrix!): src dense [nrows][k],
// imatrix f32 [k] on the device (one value per input column, shared by the rows), ggml_type 10 Q2_K / 11 Q3_K / 12 Q4_K / 13 Q5_K / 14 Q6_K.  Returns 0, -1 for other
// types / dtypes or k % 256 != 0 (the caller then quantizes without the importance vector, as the reference does for non-K-quant targets).
extern "C" int mrs_isq_quantize_imatrix(const void *src, int src_dtype, void *dst, long long nrows, int k, int ggml_type, const float *imatrix, void *stream) {
  if (nrows <= 0) return 0;
  if (!src || !dst || !imatrix || k <= 0 || k % 256) return -1;
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
  case 0: return mrs::isq_dispatch_imatrix<float>((const float *)src, (uint8_t *)dst, (size_t)nrows, k, ggml_type, imatrix, s);
  case 1: return mrs::isq_dispatch_imatrix<mrs::f16_t>((const mrs::f16_t *)src, (uint8_t *)dst, (size_t)nrows, k, ggml_type, imatrix, s);
  case 30: return mrs::isq_dispatch_imatrix<mrs::bf16_t>((const mrs::bf16_t *)src, (uint8_t *)dst, (size_t)nrows, k, ggml_type, imatrix, s);
  default: return -1;
  }
}

// QuantMethod::dequantize_w for GGUF blocks (gguf/mod.rs:430-432 -> QTensor::dequantize): packed [nrows][K/blk] -> dense [nrows][K] of
// out_dtype 0 = f32, 1 = f16, 30 = bf16 (values computed in f32, one rounding).  Returns 0, -1 for an unknown type / dtype or K not a
// multiple of the block size.
extern "C" int mrs_dequantize(const void *w, int ggml_type, long long nrows, int K, void *out, int out_dtype, void *stream) {
  if (nrows <= 0 || K <= 0) return 0;
  const int blk = (ggml_type >= 10 && ggml_type <= 14) ? 256 : 32;
  if (K % blk) return -1;
  hipStream_t s = (hipStream_t)stream;
  switch (out_dtype) {
  case 0: return mrs::dequantize_dispatch<float>((const uint8_t *)w, (float *)out, nrows, K, ggml_type, s);
  case 1: return mrs::dequantize_dispatch<mrs::f16_t>((const uint8_t *)w, (mrs::f16_t *)out, nrows, K, ggml_type, s);
  case 30: return mrs::dequantize_dispatch<mrs::bf16_t>((const uint8_t *)w, (mrs::bf16_t *)out, nrows, K, ggml_type, s);
  default: return -1;
  }
}

// imatrix statistics (imatrix.rs:73-135).  x: [rows][cols] of dtype 0 = f32 / 1 = f16 / 30 = bf16; accum: f32 [cols], updated in place.
extern "C" int mrs_imatrix_accumulate(const void *x, int dtype, long long rows, int cols, float *accum, void *stream) {
  if (!x || !accum || rows < 0 || cols <= 0) return -1;
  if (rows == 0) return 0;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
  case 0: hipLaunchKernelGGL(mrs::imatrix_dense_kernel<float>, grid, block, 0, s, (const float *)x, accum, rows, cols); return 0;
  case 1: hipLaunchKernelGGL(mrs::imatrix_dense_kernel<mrs::f16_t>, grid, block, 0, s, (const mrs::f16_t *)x, accum, rows, cols); return 0;
  case 30: hipLaunchKernelGGL(mrs::imatrix_dense_kernel<mrs::bf16_t>, grid, block, 0, s, (const mrs::bf16_t *)x, accum, rows, cols); return 0;
  default: return -1;
  }
}
// routed layers: ids [n][k] u32 expert of every (token, slot); x [n][cols] (per_slot = 0) or [n][k][cols] (per_slot = 1); accum f32 [num_experts][cols],
// counts f32 [num_experts], both updated in place.
extern "C" int mrs_imatrix_accumulate_routed(const void *x, int dtype, const uint32_t *ids, int n, int k, int cols, int per_slot, int num_experts,
                                             float *accum, float *counts, void *stream) {
  if (!x || !ids || !accum || !counts || n < 0 || k <= 0 || cols <= 0 || num_experts <= 0) return -1;
  if (n == 0) return 0;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
  case 0: hipLaunchKernelGGL(mrs::imatrix_routed_kernel<float>, grid, block, 0, s, (const float *)x, ids, accum, counts, n, k, cols, per_slot, num_experts); return 0;
  case 1: hipLaunchKernelGGL(mrs::imatrix_routed_kernel<mrs::f16_t>, grid, block, 0, s, (const mrs::f16_t *)x, ids, accum, counts, n, k, cols, per_slot, num_experts); return 0;
  case 30: hipLaunchKernelGGL(mrs::imatrix_routed_kernel<mrs::bf16_t>, grid, block, 0, s, (const mrs::bf16_t *)x, ids, accum, counts, n, k, cols, per_slot, num_experts); return 0;
  default: return -1;
  }
}
