// ext_gemm2.hip -- prompt GEMM, round 3: block-dequant fused into a bf16 MFMA GEMM with the WEIGHT operand taken straight from global memory in
// MFMA operand layout (no LDS round trip for B), LDS for the activations only.
//
// Role: fast_mmq::{plain, fused_qkv, fused_glu, fused_ffn} (mistralrs-quant/src/gguf/fast_mmq.rs:528-635,762-821) -- as ext_gemm.hip, whose
// kernels stay for M <= 128, Q5_K / Q8_0 and the MoE forms.  Same arithmetic as ext_gemm.hip: w = fma(d * sc, q, -(dmin * m)) in f32, rounded
// once to bf16; activations pre-rounded to bf16 (k-slab-major [K/64][M][64]); v_mfma_f32_32x32x16_bf16, k ascending -> identical bits.
//
// Why (profiles/round2_prefill_gemm.md): in the 256 x 128 x 64 tile of ext_gemm.hip a k-step needs about one MFMA-time of EACH of three resources -- the
// matrix pipe, ~270 VALU issue slots per SIMD for the block decode, ~1000 LDS-array cycles (48 KiB written, 96-128 KiB read) -- and they
// overlap poorly (2.1-2.4 x the MFMA time).  Here:
//   * a wave owns 64 weight rows (two MFMA B tiles) and 128 token rows: one decoded B fragment (8 weights per lane, ~22 VALU) feeds 4 MFMAs, so the
//     decode is ~6 VALU per MFMA instead of ~9, and it never touches LDS;
//   * the weights are read in a load-time MFMA LAYOUT (mrs_gemm2_repack): per 32-row tile and 64-k chunk one KiB holding, lane by lane,
//     the 32 quants the lane's four B fragments of that chunk need -> one coalesced 16-byte load per lane per 64 k (Q6_K: + 8 bytes of high
//     bits), scales in a second plane (4 bytes per row per chunk);
//   * LDS carries the activation tile only: 32 KiB written and 128 KiB read (8 waves x 16 fragments, each feeding two MFMAs) per 64 k against
//     2048 matrix-pipe cycles -> 64 B / clk, a quarter of the ds_read_b128 peak.
// Workgroup tile 256 (tokens) x 256 (weight rows) x 64 (k), 512 threads, 8 waves x (128 x 64), 128 accumulator registers per lane, LDS 64 KiB
// (double-buffered A).  Grid (n tiles, m tiles, k splits); shapes with fewer tiles than CUs split K into f32 partials + the fixed-order
// reduce kernel of ext_gemm.hip (deterministic).
#include "common.cuh"
#include "gguf_blocks.cuh"
#include <stdlib.h>
#include <algorithm>

namespace mrs {
namespace g2 {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

constexpr int TM = 256, TN = 256, TK = 64, NT2 = 512;

// ------------------------------------------------------------------------------------------------ MFMA layout of a weight tensor
// n tiles of 32 rows (the last one padded with zero rows), chunks of 64 k.  Lane l of a wave = (row n = l % 32, k half kh = l / 32); its B
// fragment of MFMA step s (k = chunk * 64 + s * 16 + kh * 8 + j, j < 8) needs 8 weights of row n.
//   q plane   [ntile][chunk][64 lanes][QB bytes]   Q4_K: 16 B = bytes qs[t * 16 + kh * 8 + j] of the chunk's 32, t = 0, 1: low nibbles = steps 0, 1,
//                                                               high nibbles = steps 2, 3
//                                                  Q6_K: 16 B of 4-bit parts, dword (r, t) = steps s = 2 r + t: value j in bits 8 (j % 4) + 4 (j / 4);
//                                                        + 8 B of 2-bit parts, dword r: value (t, j) in bits 8 (j % 4) + 2 (j / 4) + 4 t
//   s plane   [ntile][chunk][32 rows][4 B]         Q4_K: sc(2c) m(2c) sc(2c+1) m(2c+1) as bytes;  Q6_K: the 4 int8 scales of the chunk
//   d plane   [ntile][superblock][32 rows][4 B]    Q4_K: f16 d, f16 dmin;  Q6_K: f16 d, 0
struct Layout { size_t q, s, d, total; int qb; };
__host__ __device__ inline Layout layout_of(int type, long long n, long long k) {
  const size_t nt = (size_t)((n + 31) / 32), ch = (size_t)(k / 64), sb = (size_t)(k / 256);
  Layout L{};
  L.qb = type == T_Q6_K ? 24 : 16;
  L.q = 0;
  L.s = (nt * ch * 64 * (size_t)L.qb + 255) & ~(size_t)255;
  L.d = (L.s + nt * ch * 32 * 4 + 255) & ~(size_t)255;
  L.total = (L.d + nt * sb * 32 * 4 + 255) & ~(size_t)255;
  return L;
}
__host__ __device__ inline bool type_ok(int t) { return t == T_Q4_K || t == T_Q6_K; }

// one thread per (ntile, chunk, lane)
template <int TYPE>
__global__ void __launch_bounds__(256) repack2_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const Layout L, long long n, long long k) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long ch = k / 64, nt = (n + 31) / 32;
  if (gid >= nt * ch * 64) return;
  const int lane = (int)(gid & 63), row32 = lane & 31, kh = lane >> 5;
  const long long c = (gid >> 6) % ch, tile = (gid >> 6) / ch;
  const long long row = tile * 32 + row32;
  const int sb = (int)(c >> 2), cc = (int)(c & 3);
  const bool live = row < n;
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *b = src + ((size_t)(live ? row : 0) * (size_t)(k / 256) + (size_t)sb) * 144;  // padding rows of the last tile never touch memory past the tensor
    uint8_t *q = dst + L.q + (size_t)gid * 16;
    for (int t = 0; t < 2; ++t)
      for (int j = 0; j < 8; ++j) q[t * 8 + j] = live ? b[16 + cc * 32 + t * 16 + kh * 8 + j] : 0;
    if (kh == 0) {
      uint8_t *s = dst + L.s + ((size_t)(tile * ch + c) * 32 + row32) * 4;
      uint8_t sc[2], mn[2];
      for (int i = 0; i < 2; ++i) {  // get_scale_min_k4 (marlin_gguf_affine_repack.cu:200-210)
        const int g = 2 * cc + i;
        const uint8_t *p = b + 4;
        if (g < 4) { sc[i] = p[g] & 63; mn[i] = p[g + 4] & 63; }
        else { sc[i] = (p[g + 4] & 15) | ((p[g - 4] >> 6) << 4); mn[i] = (p[g + 4] >> 4) | ((p[g] >> 6) << 4); }
      }
      s[0] = live ? sc[0] : 0; s[1] = live ? mn[0] : 0; s[2] = live ? sc[1] : 0; s[3] = live ? mn[1] : 0;
      if (cc == 0) {
        uint8_t *d = dst + L.d + ((size_t)(tile * (ch / 4) + sb) * 32 + row32) * 4;
        for (int i = 0; i < 4; ++i) d[i] = live ? b[i] : 0;
      }
    }
  } else {  // Q6_K
    const uint8_t *b = src + ((size_t)(live ? row : 0) * (size_t)(k / 256) + (size_t)sb) * 210;
    const uint8_t *ql = b, *qh = b + 128;
    const int8_t *scs = (const int8_t *)(b + 192);
    uint32_t lo[4] = {0, 0, 0, 0}, hi[2] = {0, 0};
    for (int r = 0; r < 2; ++r)
      for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 8; ++j) {
          const int e = cc * 64 + r * 32 + t * 16 + kh * 8 + j;  // element of the superblock
          const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32, ii = hh * 64 + pos + (qt % 2) * 32;
          const uint32_t l4 = live ? (qt < 2 ? (ql[ii] & 15u) : (uint32_t)(ql[ii] >> 4)) : 0u, h2 = live ? ((qh[hh * 32 + pos] >> (qt * 2)) & 3u) : 0u;
          lo[r * 2 + t] |= l4 << (8 * (j & 3) + 4 * (j >> 2));
          hi[r] |= h2 << (8 * (j & 3) + 2 * (j >> 2) + 4 * t);
        }
    uint32_t *q = (uint32_t *)(dst + L.q + (size_t)gid * 24);
    for (int i = 0; i < 4; ++i) q[i] = lo[i];
    q[4] = hi[0]; q[5] = hi[1];
    if (kh == 0) {
      uint8_t *s = dst + L.s + ((size_t)(tile * ch + c) * 32 + row32) * 4;
      for (int i = 0; i < 4; ++i) s[i] = live ? (uint8_t)scs[cc * 4 + i] : 0;
      if (cc == 0) {
        uint8_t *d = dst + L.d + ((size_t)(tile * (ch / 4) + sb) * 32 + row32) * 4;
        d[0] = live ? b[208] : 0; d[1] = live ? b[209] : 0; d[2] = 0; d[3] = 0;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct Args {
  const uint8_t *w[3];   // MFMA-layout tensors (mrs_gemm2_repack), same type
  float *out[3];
  int N[3], ldo[3], tile0[3];  // tile0: first n tile (of 256) of segment i in blockIdx.x
  int nseg;
  const uint16_t *x;     // bf16 k-slab-major [K/64][M][64]
  int M, K, accumulate, splits, ldp;
  float *partial;        // [splits][M][ldp] when splits > 1 (column = global n tile * 256 + n)
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }  // as ext_gemm.hip: conflict-free ds_read_b128

template <int TYPE> struct BRaw;
template <> struct BRaw<T_Q4_K> { v4u q; unsigned s; };
template <> struct BRaw<T_Q6_K> { v4u q; v2u h; unsigned s; };

// 8 weights of MFMA step s as 4 packed bf16 pairs
template <int TYPE> __device__ __forceinline__ bf16x8 decode_frag(const BRaw<TYPE> &w, unsigned dd, int s) {
  unsigned o[4];
  if constexpr (TYPE == T_Q4_K) {
    const float d = half_bits_to_float((uint16_t)(dd & 0xffff)), dmin = half_bits_to_float((uint16_t)(dd >> 16));
    const int hi = s >> 1, t = s & 1;
    const float sc = d * (float)((w.s >> (16 * hi)) & 0xff), m = dmin * (float)((w.s >> (16 * hi + 8)) & 0xff);
    const unsigned q0 = ((t ? w.q.z : w.q.x) >> (4 * hi)) & 0x0f0f0f0fu, q1 = ((t ? w.q.w : w.q.y) >> (4 * hi)) & 0x0f0f0f0fu;
    o[0] = pack_bf16(fmaf(sc, (float)(q0 & 0xff), -m), fmaf(sc, (float)((q0 >> 8) & 0xff), -m));
    o[1] = pack_bf16(fmaf(sc, (float)((q0 >> 16) & 0xff), -m), fmaf(sc, (float)(q0 >> 24), -m));
    o[2] = pack_bf16(fmaf(sc, (float)(q1 & 0xff), -m), fmaf(sc, (float)((q1 >> 8) & 0xff), -m));
    o[3] = pack_bf16(fmaf(sc, (float)((q1 >> 16) & 0xff), -m), fmaf(sc, (float)(q1 >> 24), -m));
  } else {
    const int r = s >> 1, t = s & 1;
    const float sc = half_bits_to_float((uint16_t)(dd & 0xffff)) * (float)(int)(int8_t)((w.s >> (8 * s)) & 0xff), m = 32.0f * sc;
    const unsigned l = s == 0 ? w.q.x : (s == 1 ? w.q.y : (s == 2 ? w.q.z : w.q.w));
    const unsigned h = r ? w.h.y : w.h.x;
    const unsigned q0 = (l & 0x0f0f0f0fu) | (((h >> (4 * t)) & 0x03030303u) << 4), q1 = ((l >> 4) & 0x0f0f0f0fu) | (((h >> (4 * t + 2)) & 0x03030303u) << 4);
    o[0] = pack_bf16(fmaf(sc, (float)(q0 & 0xff), -m), fmaf(sc, (float)((q0 >> 8) & 0xff), -m));
    o[1] = pack_bf16(fmaf(sc, (float)((q0 >> 16) & 0xff), -m), fmaf(sc, (float)(q0 >> 24), -m));
    o[2] = pack_bf16(fmaf(sc, (float)(q1 & 0xff), -m), fmaf(sc, (float)((q1 >> 8) & 0xff), -m));
    o[3] = pack_bf16(fmaf(sc, (float)((q1 >> 16) & 0xff), -m), fmaf(sc, (float)(q1 >> 24), -m));
  }
  const v4u v = {o[0], o[1], o[2], o[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// Wave tile 128 (tokens) x 64 (weight rows): 4 x 2 MFMA tiles, 128 accumulator registers.  An A fragment read from LDS feeds 2 MFMAs and a decoded B
// fragment 4 (the first version gave every wave 256 x 32: one LDS read per MFMA = 256 KiB per 64 k per CU, as much LDS time as matrix time --
// profiles/round3_prefill.md).  Waves: wm = wave & 1 (token half), wn = wave >> 1 (64-row group of the 256 weight rows).
template <int TYPE>
__global__ void __launch_bounds__(NT2) gemm_qd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][256 x 64 bf16] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int seg = 0;
  if (a.nseg > 2 && (int)blockIdx.x >= a.tile0[2]) seg = 2;
  else if (a.nseg > 1 && (int)blockIdx.x >= a.tile0[1]) seg = 1;
  const int segN = a.N[seg];
  const int wm = (wave & 1) * 128, wn = (wave >> 1) * 64;
  const int m0 = blockIdx.y * TM, n0 = ((int)blockIdx.x - a.tile0[seg]) * TN + wn;  // the wave's first weight row
  const int nk_all = a.K / TK, kz = (int)blockIdx.z;
  const int k_lo = (int)((long)nk_all * kz / a.splits), k_hi = (int)((long)nk_all * (kz + 1) / a.splits), nk = k_hi - k_lo;
  const Layout L = layout_of(TYPE, segN, a.K);
  const uint8_t *wb = a.w[seg];
  // descriptors: the planes of the wave's two 32-row tiles.  A tile past the tensor's last rows is clamped to tile 0 (it multiplies real data and stores nothing)
  __amdgpu_buffer_rsrc_t rq[2], rs[2], rd[2];
  bool live[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    live[j] = n0 + 32 * j < segN;
    const size_t tsel = live[j] ? (size_t)((n0 >> 5) + j) : 0;
    rq[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + L.q + tsel * (size_t)nk_all * 64 * L.qb), (short)0, nk_all * 64 * L.qb, 0x00020000);
    rs[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + L.s + tsel * (size_t)nk_all * 128), (short)0, nk_all * 128, 0x00020000);
    rd[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + L.d + tsel * (size_t)(nk_all / 4) * 128), (short)0, (nk_all / 4) * 128, 0x00020000);
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, (short)0, (int)((size_t)a.M * a.K * 2), 0x00020000);
  const int xc = tid & 7, xr0 = tid >> 3;  // A: 16-byte chunk, rows xr0 + 64 i
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xoff[i] = (unsigned)((min(m0 + xr0 + 64 * i, a.M - 1) * TK + xc * 8) * 2);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4u xa[4];
  BRaw<TYPE> bw[2];
  unsigned bd[2] = {0, 0};
  auto issue = [&](int kb_raw) {  // unconditional, k index clamped (a load under a branch makes hipcc drain vmcnt)
    const int kb = k_lo + min(kb_raw, nk - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff[i] + (unsigned)kb * (unsigned)(a.M * TK * 2), 0, 16);  // sc1: keep the use-once A tile out of L1
    const unsigned qo = ((unsigned)kb * 64u + (unsigned)lane) * (unsigned)L.qb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bw[j].q = __builtin_amdgcn_raw_buffer_load_b128(rq[j], qo, 0, 0);
      if constexpr (TYPE == T_Q6_K) {
        bw[j].h.x = __builtin_amdgcn_raw_buffer_load_b32(rq[j], qo + 16u, 0, 0);
        bw[j].h.y = __builtin_amdgcn_raw_buffer_load_b32(rq[j], qo + 20u, 0, 0);
      }
      bw[j].s = __builtin_amdgcn_raw_buffer_load_b32(rs[j], ((unsigned)kb * 32u + (unsigned)(lane & 31)) * 4u, 0, 0);
      bd[j] = __builtin_amdgcn_raw_buffer_load_b32(rd[j], ((unsigned)(kb >> 2) * 32u + (unsigned)(lane & 31)) * 4u, 0, 0);
    }
  };
  auto commit = [&](char *buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(v4u *)(buf + tile_off(xr0 + 64 * i, xc)) = xa[i];
  };
  const int frow = lane & 31, fk = lane >> 5;
  char *buf0 = smem, *buf1 = smem + TM * TK * 2;
  issue(0);
  commit(buf0);
  BRaw<TYPE> cw[2] = {bw[0], bw[1]};
  unsigned cd[2] = {bd[0], bd[1]};
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const char *A = (kb & 1) ? buf1 : buf0;
    issue(kb + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 bf0 = decode_frag<TYPE>(cw[0], cd[0], s), bf1 = decode_frag<TYPE>(cw[1], cd[1], s);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 af = *(const bf16x8 *)(A + tile_off(wm + i * 32 + frow, s * 2 + fk));
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf1, acc[i][1], 0, 0, 0);
      }
    }
    commit((kb & 1) ? buf0 : buf1);  // stage kb + 1 goes to the other buffer: nobody reads it during this stage
    cw[0] = bw[0]; cw[1] = bw[1];
    cd[0] = bd[0]; cd[1] = bd[1];
    __syncthreads();
  }
  float *obase;
  int ldo, ncol0;
  if (a.splits > 1) { obase = a.partial + (size_t)blockIdx.z * a.M * a.ldp; ldo = a.ldp; ncol0 = a.tile0[seg] * TN; }
  else { obase = a.out[seg]; ldo = a.ldo[seg]; ncol0 = 0; }
  const bool accum = a.accumulate && a.splits == 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 32 * j + (lane & 31);
    if (!live[j]) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < segN) {
          float *p = obase + (size_t)m * ldo + ncol0 + n;
          *p = accum ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
  }
}

// the fixed-order reduce of the split-K partials (same order as ext_gemm.hip: split 0 first)
__global__ void __launch_bounds__(256) gemm2_splitk_reduce_kernel(const Args a) {
  const long long total = (long long)a.M * a.ldp, i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int m = (int)(i / a.ldp), c = (int)(i % a.ldp), tile = c / TN, nn = c % TN;
  int seg = 0;
  if (a.nseg > 2 && tile >= a.tile0[2]) seg = 2;
  else if (a.nseg > 1 && tile >= a.tile0[1]) seg = 1;
  const int n = (tile - a.tile0[seg]) * TN + nn;
  if (n >= a.N[seg]) return;
  float s = 0.f;
  for (int z = 0; z < a.splits; ++z) s += a.partial[((size_t)z * a.M + m) * a.ldp + c];
  float *p = a.out[seg] + (size_t)m * a.ldo[seg] + n;
  *p = a.accumulate ? *p + s : s;
}

}  // namespace g2
}  // namespace mrs

using namespace mrs;
using namespace mrs::g2;

extern "C" int mrs_gemm2_supported(int ggml_type) { return type_ok(ggml_type) ? 1 : 0; }
extern "C" size_t mrs_gemm2_repack_bytes(int type, long long n, long long k) { return type_ok(type) && k % 256 == 0 && n > 0 ? layout_of(type, n, k).total : 0; }
// GGUF blocks [n][k / 256] -> the MFMA layout (load time; pure bit permutation + the 6-bit scale expansion)
extern "C" int mrs_gemm2_repack(const void *gguf_blocks, int type, long long n, long long k, void *dst, void *stream) {
  if (!mrs_gemm2_repack_bytes(type, n, k) || !gguf_blocks || !dst) return -1;
  const Layout L = layout_of(type, n, k);
  const long long threads = ((n + 31) / 32) * (k / 64) * 64;
  const dim3 grid((unsigned)((threads + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dst, 0, L.total, s) != hipSuccess) return -1;
  if (type == T_Q4_K) hipLaunchKernelGGL(repack2_kernel<T_Q4_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)dst, L, n, k);
  else hipLaunchKernelGGL(repack2_kernel<T_Q6_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)dst, L, n, k);
  return 0;
}
// f32 partials of the split-K launches: splits * M * ldp floats, ldp = total n tiles * 256
extern "C" size_t mrs_gemm2_workspace_bytes(int M, long long n_total_padded, int max_splits) { return (size_t)max_splits * (size_t)M * (size_t)n_total_padded * 4; }

// out[i] [M][ldo[i]] (+)= x [M][K] . W_i^T for up to three MFMA-layout tensors of one type sharing the activations (fused q / k / v, gate / up).
// x_slabs: bf16 k-slab-major.  workspace (may be NULL): f32 partials for split-K; without it the launch does not split.
// Returns 0, -1 on bad arguments, -3 when the shape belongs to ext_gemm.hip (type, M <= 128, K % 256).
extern "C" int mrs_gemm2_q_bf16_multi(int nseg, const void *const *w, const int *N, float *const *out, const int *ldo, int ggml_type, int K,
                                      const void *x_slabs, int M, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  if (nseg < 1 || nseg > 3 || !w || !N || !out || !ldo || !x_slabs || M <= 0 || K <= 0) return -1;
  if (!type_ok(ggml_type) || K % 256 || M <= 128) return -3;
  Args a{};
  a.nseg = nseg; a.x = (const uint16_t *)x_slabs; a.M = M; a.K = K; a.accumulate = accumulate; a.splits = 1;
  int tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!w[i] || !out[i] || N[i] <= 0) return -1;
    a.w[i] = (const uint8_t *)w[i]; a.out[i] = out[i]; a.N[i] = N[i]; a.ldo[i] = ldo[i]; a.tile0[i] = tiles;
    tiles += (N[i] + TN - 1) / TN;
  }
  const int mt = (M + TM - 1) / TM, nk = K / TK;
  static const int cus = [] { int dev = 0, c = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256; return c; }();
  int splits = 1;
  if (workspace && tiles * mt < cus) {  // fewer workgroups than CUs: split K (each split keeps >= 8 k-steps)
    splits = std::min(std::min(std::max(1, cus / (tiles * mt)), std::max(1, nk / 16)), 4);  // every split keeps >= 16 k-steps; partial traffic grows with the split count
    a.ldp = tiles * TN;
    while (splits > 1 && (size_t)splits * M * a.ldp * 4 > workspace_bytes) --splits;
  }
  { static const int force = [] { const char *e = getenv("MRS_GEMM2_SPLITS"); return e ? atoi(e) : 0; }(); if (force > 0 && workspace) { splits = std::min(force, std::max(1, nk)); a.ldp = tiles * TN; if ((size_t)splits * M * a.ldp * 4 > workspace_bytes) return -1; } }
  a.splits = splits;
  a.partial = splits > 1 ? (float *)workspace : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(tiles, mt, splits);
  const size_t lds = (size_t)2 * TM * TK * 2;
  if (ggml_type == T_Q4_K) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)gemm_qd_kernel<T_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(gemm_qd_kernel<T_Q4_K>, grid, dim3(NT2), lds, s, a);
  } else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)gemm_qd_kernel<T_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(gemm_qd_kernel<T_Q6_K>, grid, dim3(NT2), lds, s, a);
  }
  if (splits > 1) {
    const long long total = (long long)M * a.ldp;
    hipLaunchKernelGGL(gemm2_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
  return 0;
}
