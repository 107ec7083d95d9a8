// gguf_blocks.cuh -- per-format "lane slice" decode for GGUF weight blocks on gfx950.
//
// Every supported format is consumed in slices of 32 weights per lane: two runs (A, B) of 16
// integer weights packed as 4 x int8x4 dwords, each run with one float scale and one float
// offset ( w = s*q - o ), and each run paired with one 16-element run of the int8 activation.
// This makes the dot product identical for all formats:
//     partial = sA*d8A*<qA,uA> + sB*d8B*<qB,uB> - oA*SA - oB*SB
// Block layouts: reference mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:134-226 and
// kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu:44-278 (the in-tree format spec);
// restated in oracle/ggml_oracle.c (block_ints / block_affine), which the tests check against.
//
// Weight rows are read straight from HBM into VGPRs with 16-byte loads (no LDS round trip: each
// weight byte is used exactly once per token); consecutive lanes read consecutive 16-byte pieces
// of the packed row so a wave's load instruction covers one contiguous 1-1.2 KiB span.
#pragma once
#include "common.cuh"

namespace mrs {

// ggml type ids (reference: mistralrs-quant/src/gguf/archive.rs:73-160)
enum : int { T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q8_1 = 9, T_Q2_K = 10, T_Q3_K = 11,
             T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14 };

struct Slice {
  int4 qa, qb;   // 16 + 16 weights, one per byte (non-negative, or signed for Q8_0)
  float sa, sb;  // scales
  float oa, ob;  // offsets (w = s*q - o)
};

template <int TYPE> struct Fmt;
// BLK: weights per block, TS: bytes per block, HAS_OFFSET: o != 0,
// SUM_MODE: 0 = offset term uses d8*sum(u) over the run (K-quants: vec_dot_q4_K_q8_1 etc. recompute it
//               with dp4a(0x01010101,u)), 1 = offset term uses the stored half(sum x) of the whole Q8_1
//               block, split evenly between the two runs (vec_dot_q4_0_q8_1_impl & friends)
template <> struct Fmt<T_Q4_0> { static constexpr int BLK = 32, TS = 18; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 1; };
template <> struct Fmt<T_Q4_1> { static constexpr int BLK = 32, TS = 20; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 1; };
template <> struct Fmt<T_Q5_0> { static constexpr int BLK = 32, TS = 22; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 1; };
template <> struct Fmt<T_Q5_1> { static constexpr int BLK = 32, TS = 24; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 1; };
template <> struct Fmt<T_Q8_0> { static constexpr int BLK = 32, TS = 34; static constexpr bool HAS_OFFSET = false; static constexpr int SUM_MODE = 0; };
// Q8_1 as a WEIGHT format (MoE launchers only: indexed_moe.cu:483-502, moe_grouped.cu:471-490): every vec_dot call returns
// d_w d_x sumi + s_w s_x and a block takes QI8_1 / VDR = 4 calls, i.e. per block d_w d_x <q,u> + 4 s_w s_x -> offset o = -4 s_w against the stored s_x
template <> struct Fmt<T_Q8_1> { static constexpr int BLK = 32, TS = 36; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 1; };
template <> struct Fmt<T_Q2_K> { static constexpr int BLK = 256, TS = 84; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 0; };
template <> struct Fmt<T_Q3_K> { static constexpr int BLK = 256, TS = 110; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 0; };
template <> struct Fmt<T_Q4_K> { static constexpr int BLK = 256, TS = 144; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 0; };
template <> struct Fmt<T_Q5_K> { static constexpr int BLK = 256, TS = 176; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 0; };
template <> struct Fmt<T_Q6_K> { static constexpr int BLK = 256, TS = 210; static constexpr bool HAS_OFFSET = true; static constexpr int SUM_MODE = 0; };

// Activation run indices (run = 16 consecutive activations) of slice `s` of a row.
template <int TYPE> __device__ __forceinline__ void slice_runs(int s, int &ra, int &rb) {
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const int sb = s >> 3, l = s & 7;
    ra = sb * 16 + (l >> 1) * 4 + (l & 1);  // element sb*256 + c*64 + 16*(l&1)
    rb = ra + 2;                            // + 32 elements
  } else if constexpr (TYPE == T_Q6_K) {
    const int sb = s >> 3, l = s & 7;
    ra = sb * 16 + (l >> 2) * 8 + (l & 3);  // element sb*256 + h*128 + qt*32 + 16*(m&1),  m = l&3
    rb = ra + 4;                            // + 64 elements
  } else {
    ra = 2 * s;
    rb = ra + 1;
  }
}

__device__ __forceinline__ int4 and4(int4 v, int m) { return make_int4(v.x & m, v.y & m, v.z & m, v.w & m); }
__device__ __forceinline__ int4 shr4(int4 v, int n) {
  return make_int4((int)((unsigned)v.x >> n), (int)((unsigned)v.y >> n), (int)((unsigned)v.z >> n), (int)((unsigned)v.w >> n));
}
__device__ __forceinline__ int4 shl4(int4 v, int n) { return make_int4(v.x << n, v.y << n, v.z << n, v.w << n); }
__device__ __forceinline__ int4 or4(int4 a, int4 b) { return make_int4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }

// spread the low 4 bits of x to bit `pos` of the 4 bytes of a dword
__device__ __forceinline__ int spread4(unsigned x, int pos) {
  return (int)((((x & 1u)) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21)) << pos);
}

template <int TYPE> __device__ __forceinline__ Slice load_slice(const uint8_t *__restrict__ row, int s) {
  Slice r;
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    // [half d, half dmin][12 B 6-bit scales/mins]([32 B qh])[128 B qs]
    const uint8_t *blk = row + (size_t)(s >> 3) * Fmt<TYPE>::TS;
    const int l = s & 7, c = l >> 1;
    const int4 hdr = ld16_a4(blk);
    const int4 qs = ld16_a4(blk + (TYPE == T_Q4_K ? 16 : 48) + l * 16);
    const float d = half_bits_to_float((uint16_t)(hdr.x & 0xffff));
    const float dmin = half_bits_to_float((uint16_t)((unsigned)hdr.x >> 16));
    const int sh = 16 * (c & 1);
    const unsigned y = (unsigned)hdr.y >> sh, z = (unsigned)hdr.z >> sh, w = (unsigned)hdr.w >> sh;
    unsigned sc0, sc1, m0, m1;
    if (c < 2) {
      sc0 = y & 63; sc1 = (y >> 8) & 63; m0 = z & 63; m1 = (z >> 8) & 63;
    } else {
      sc0 = (w & 15) | (((y >> 6) & 3) << 4);
      sc1 = ((w >> 8) & 15) | (((y >> 14) & 3) << 4);
      m0 = ((w >> 4) & 15) | (((z >> 6) & 3) << 4);
      m1 = ((w >> 12) & 15) | (((z >> 14) & 3) << 4);
    }
    r.qa = and4(qs, 0x0F0F0F0F);
    r.qb = and4(shr4(qs, 4), 0x0F0F0F0F);
    if constexpr (TYPE == T_Q5_K) {
      const int4 qh = ld16_a4(blk + 16 + 16 * (l & 1));
      r.qa = or4(r.qa, shl4(and4(shr4(qh, 2 * c), 0x01010101), 4));
      r.qb = or4(r.qb, shl4(and4(shr4(qh, 2 * c + 1), 0x01010101), 4));
    }
    r.sa = d * (float)sc0; r.sb = d * (float)sc1;
    r.oa = dmin * (float)m0; r.ob = dmin * (float)m1;
  } else if constexpr (TYPE == T_Q6_K) {
    // [128 B ql][64 B qh][16 x int8 scales][half d], 210 B => only 2-byte aligned
    const uint8_t *blk = row + (size_t)(s >> 3) * 210;
    const int l = s & 7, h = l >> 2, m = l & 3, qt = m >> 1;
    const int4 ql = ld16_a2(blk + l * 16);
    const int4 qh = ld16_a2(blk + 128 + h * 32 + 16 * (m & 1));
    const int g = h * 8 + qt * 2 + (m & 1);
    const float d = half_bits_to_float(ld2(blk + 208));
    const float s0 = d * (float)(int)(int8_t)blk[192 + g];
    const float s1 = d * (float)(int)(int8_t)blk[192 + g + 4];
    r.qa = or4(and4(ql, 0x0F0F0F0F), shl4(and4(shr4(qh, 2 * qt), 0x03030303), 4));
    r.qb = or4(and4(shr4(ql, 4), 0x0F0F0F0F), shl4(and4(shr4(qh, 2 * qt + 4), 0x03030303), 4));
    r.sa = s0; r.sb = s1; r.oa = 32.0f * s0; r.ob = 32.0f * s1;
  } else if constexpr (TYPE == T_Q8_0) {
    const uint8_t *blk = row + (size_t)s * 34;
    const float d = half_bits_to_float(ld2(blk));
    r.qa = ld16_a2(blk + 2); r.qb = ld16_a2(blk + 18);
    r.sa = r.sb = d; r.oa = r.ob = 0.0f;
  } else if constexpr (TYPE == T_Q8_1) {
    const uint8_t *blk = row + (size_t)s * 36;  // [half d][half s][32 x int8], 4-byte aligned
    const unsigned ds = *(const unsigned *)blk;
    r.qa = ld16_a4(blk + 4); r.qb = ld16_a4(blk + 20);
    r.sa = r.sb = half_bits_to_float((uint16_t)(ds & 0xffff));
    r.oa = r.ob = -4.0f * half_bits_to_float((uint16_t)(ds >> 16));
  } else if constexpr (TYPE == T_Q4_0 || TYPE == T_Q4_1) {
    const uint8_t *blk = row + (size_t)s * Fmt<TYPE>::TS;
    const float d = half_bits_to_float(ld2(blk));
    const int4 v = ld16_a2(blk + (TYPE == T_Q4_0 ? 2 : 4));
    r.qa = and4(v, 0x0F0F0F0F); r.qb = and4(shr4(v, 4), 0x0F0F0F0F);
    r.sa = r.sb = d;
    r.oa = r.ob = (TYPE == T_Q4_0) ? 8.0f * d : -half_bits_to_float(ld2(blk + 2));
  } else if constexpr (TYPE == T_Q5_0 || TYPE == T_Q5_1) {
    const uint8_t *blk = row + (size_t)s * Fmt<TYPE>::TS;
    const float d = half_bits_to_float(ld2(blk));
    const unsigned qh = (unsigned)ld4_a2(blk + (TYPE == T_Q5_0 ? 2 : 4));
    const int4 v = ld16_a2(blk + (TYPE == T_Q5_0 ? 6 : 8));
    r.qa = and4(v, 0x0F0F0F0F); r.qb = and4(shr4(v, 4), 0x0F0F0F0F);
    r.qa.x |= spread4(qh, 4);        r.qa.y |= spread4(qh >> 4, 4);
    r.qa.z |= spread4(qh >> 8, 4);   r.qa.w |= spread4(qh >> 12, 4);
    r.qb.x |= spread4(qh >> 16, 4);  r.qb.y |= spread4(qh >> 20, 4);
    r.qb.z |= spread4(qh >> 24, 4);  r.qb.w |= spread4(qh >> 28, 4);
    r.sa = r.sb = d;
    r.oa = r.ob = (TYPE == T_Q5_0) ? 16.0f * d : -half_bits_to_float(ld2(blk + 2));
  } else if constexpr (TYPE == T_Q2_K) {
    // [16 B scales (4-bit scale | 4-bit min)][64 B qs][half d][half dmin]
    const uint8_t *blk = row + (size_t)(s >> 3) * 84;
    const int l = s & 7, n = l >> 2, shift = 2 * (l & 3);
    const int4 v0 = ld16_a4(blk + 16 + n * 32), v1 = ld16_a4(blk + 16 + n * 32 + 16);
    const float d = half_bits_to_float(ld2(blk + 80)), dmin = half_bits_to_float(ld2(blk + 82));
    const unsigned sc0 = blk[2 * l], sc1 = blk[2 * l + 1];
    r.qa = and4(shr4(v0, shift), 0x03030303); r.qb = and4(shr4(v1, shift), 0x03030303);
    r.sa = d * (float)(sc0 & 15); r.sb = d * (float)(sc1 & 15);
    r.oa = dmin * (float)(sc0 >> 4); r.ob = dmin * (float)(sc1 >> 4);
  } else if constexpr (TYPE == T_Q3_K) {
    // [32 B hmask][64 B qs][12 B 6-bit scales][half d], 110 B => 2-byte aligned
    const uint8_t *blk = row + (size_t)(s >> 3) * 110;
    const int l = s & 7, n = l >> 2, shift = 2 * (l & 3);
    const int4 h0 = ld16_a2(blk), h1 = ld16_a2(blk + 16);
    const int4 v0 = ld16_a2(blk + 32 + n * 32), v1 = ld16_a2(blk + 32 + n * 32 + 16);
    const float d = half_bits_to_float(ld2(blk + 108));
    auto scale6 = [&](int j) -> int {
      const int lo = (j < 8) ? (blk[96 + j] & 0xF) : (blk[96 + j - 8] >> 4);
      const int hi = (blk[96 + 8 + (j & 3)] >> (2 * (j >> 2))) & 3;
      return (lo | (hi << 4)) - 32;
    };
    r.qa = or4(and4(shr4(v0, shift), 0x03030303), shl4(and4(shr4(h0, l), 0x01010101), 2));
    r.qb = or4(and4(shr4(v1, shift), 0x03030303), shl4(and4(shr4(h1, l), 0x01010101), 2));
    r.sa = d * (float)scale6(2 * l); r.sb = d * (float)scale6(2 * l + 1);
    r.oa = 4.0f * r.sa; r.ob = 4.0f * r.sb;
  }
  return r;
}


// ---------------------------------------------------------------------------------------------------
// Split form of load_slice for software pipelining: load_raw() only ISSUES the global loads of one slice
// (the registers it returns are not touched until decode_raw()), so a wave can keep several slices of
// the NEXT rows in flight while it decodes and accumulates the current ones.  decode_raw(load_raw(...))
// == load_slice(...) bit for bit.  The hot formats (Q4_K, Q5_K, Q6_K, Q8_0) have a true split; the other
// formats decode at load time (their Raw IS the Slice).
template <int TYPE> struct Raw { Slice s; };
template <> struct Raw<T_Q4_K> { int4 hdr, qs; };
template <> struct Raw<T_Q5_K> { int4 hdr, qs, qh; };
template <> struct Raw<T_Q6_K> { int4 ql, qh; int2 sc; unsigned d; };
template <> struct Raw<T_Q8_0> { int4 qa, qb; unsigned d; };

__device__ __forceinline__ int2 ld8_a2(const void *p) { int2_a2 v = *(const int2_a2 *)p; return make_int2(v.x, v.y); }

template <int TYPE> __device__ __forceinline__ Raw<TYPE> load_raw(const uint8_t *__restrict__ row, int s) {
  Raw<TYPE> r;
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *blk = row + (size_t)(s >> 3) * 144;
    r.hdr = ld16nt_a4(blk);
    r.qs = ld16nt_a4(blk + 16 + (s & 7) * 16);
  } else if constexpr (TYPE == T_Q5_K) {
    const uint8_t *blk = row + (size_t)(s >> 3) * 176;
    r.hdr = ld16nt_a4(blk);
    r.qh = ld16nt_a4(blk + 16 + 16 * (s & 1));
    r.qs = ld16nt_a4(blk + 48 + (s & 7) * 16);
  } else if constexpr (TYPE == T_Q6_K) {
    const uint8_t *blk = row + (size_t)(s >> 3) * 210;
    const int l = s & 7, h = l >> 2, m = l & 3;
    r.ql = ld16_a2(blk + l * 16);
    r.qh = ld16_a2(blk + 128 + h * 32 + 16 * (m & 1));
    r.sc = ld8_a2(blk + 192 + h * 8);
    r.d = ld2(blk + 208);
  } else if constexpr (TYPE == T_Q8_0) {
    const uint8_t *blk = row + (size_t)s * 34;
    r.d = ld2(blk);
    r.qa = ld16_a2(blk + 2);
    r.qb = ld16_a2(blk + 18);
  } else {
    r.s = load_slice<TYPE>(row, s);
  }
  return r;
}

template <int TYPE> __device__ __forceinline__ Slice decode_raw(const Raw<TYPE> &w, int s) {
  Slice r;
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const int l = s & 7, c = l >> 1;
    const int4 hdr = w.hdr;
    const float d = half_bits_to_float((uint16_t)(hdr.x & 0xffff));
    const float dmin = half_bits_to_float((uint16_t)((unsigned)hdr.x >> 16));
    // branch-free k4(g) for the sub-block pair (2c, 2c+1): 16-bit lanes A = scales16[c&1], B = scales16[(c&1)+2],
    // C = scales16[(c&1)+4]; c < 2: sc = A & 0x3f3f, m = B & 0x3f3f; else sc = (C & 0x0f0f) | ((A >> 2) & 0x3030),
    // m = ((C >> 4) & 0x0f0f) | ((B >> 2) & 0x3030)   (same bit surgery as vec_dot_q4_K_q8_1, mmvq_gguf.cu:600-607)
    const int sh = 16 * (c & 1);
    const unsigned A = (unsigned)hdr.y >> sh, B = (unsigned)hdr.z >> sh, C = (unsigned)hdr.w >> sh;
    const unsigned scH = (C & 0x0f0fu) | ((A >> 2) & 0x3030u), mH = ((C >> 4) & 0x0f0fu) | ((B >> 2) & 0x3030u);
    const unsigned sc = (c < 2) ? (A & 0x3f3fu) : scH, mm = (c < 2) ? (B & 0x3f3fu) : mH;
    const unsigned sc0 = sc & 0xff, sc1 = (sc >> 8) & 0xff, m0 = mm & 0xff, m1 = (mm >> 8) & 0xff;
    r.qa = and4(w.qs, 0x0F0F0F0F);
    r.qb = and4(shr4(w.qs, 4), 0x0F0F0F0F);
    if constexpr (TYPE == T_Q5_K) {
      r.qa = or4(r.qa, shl4(and4(shr4(w.qh, 2 * c), 0x01010101), 4));
      r.qb = or4(r.qb, shl4(and4(shr4(w.qh, 2 * c + 1), 0x01010101), 4));
    }
    r.sa = d * (float)sc0; r.sb = d * (float)sc1;
    r.oa = dmin * (float)m0; r.ob = dmin * (float)m1;
  } else if constexpr (TYPE == T_Q6_K) {
    const int l = s & 7, m = l & 3, qt = m >> 1;
    const int gi = qt * 2 + (m & 1);  // scale byte index inside this half's 8 scales; partner at +4
    const float d = half_bits_to_float((uint16_t)w.d);
    const int sc_lo = (int)(int8_t)(((unsigned)w.sc.x >> (8 * gi)) & 0xff);
    const int sc_hi = (int)(int8_t)(((unsigned)w.sc.y >> (8 * gi)) & 0xff);
    const float s0 = d * (float)sc_lo, s1 = d * (float)sc_hi;
    r.qa = or4(and4(w.ql, 0x0F0F0F0F), shl4(and4(shr4(w.qh, 2 * qt), 0x03030303), 4));
    r.qb = or4(and4(shr4(w.ql, 4), 0x0F0F0F0F), shl4(and4(shr4(w.qh, 2 * qt + 4), 0x03030303), 4));
    r.sa = s0; r.sb = s1; r.oa = 32.0f * s0; r.ob = 32.0f * s1;
  } else if constexpr (TYPE == T_Q8_0) {
    const float d = half_bits_to_float((uint16_t)w.d);
    r.qa = w.qa; r.qb = w.qb; r.sa = r.sb = d; r.oa = r.ob = 0.0f;
  } else {
    r = w.s;
  }
  return r;
}

__device__ __forceinline__ int dot16(int4 q, int4 u) {
  return dot4(q.w, u.w, dot4(q.z, u.z, dot4(q.y, u.y, dot4(q.x, u.x, 0))));
}

}  // namespace mrs
