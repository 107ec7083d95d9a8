// ext_dec2.hip -- prototype entry points of the round-4 GEMV core (dec_core2.cuh): decode-layout repack and a plain GEMV launch, used by the
// microbenchmark (scripts/bench_dec.py --v2) and tests/test_dec2_core.py while ext_dec.hip is ported to the new core.
#include "dec_core2.cuh"
#include <stdio.h>
#include <stdlib.h>

namespace mrs {
namespace dec2 {

#ifdef MRS_DEC2_ONLY_Q4K  // experiment builds
#define MRS_DEC2_TYPE_SWITCH(t, ...) { constexpr int TT = T_Q4_K; __VA_ARGS__ }
#else
#define MRS_DEC2_TYPE_SWITCH(t, ...)                             \
  switch (t) {                                                   \
  case T_Q4_K: { constexpr int TT = T_Q4_K; __VA_ARGS__ } break; \
  case T_Q5_K: { constexpr int TT = T_Q5_K; __VA_ARGS__ } break; \
  case T_Q6_K: { constexpr int TT = T_Q6_K; __VA_ARGS__ } break; \
  case T_Q8_0: { constexpr int TT = T_Q8_0; __VA_ARGS__ } break; \
  default: break;                                                \
  }
#endif

// ------------------------------------------------------------------------------------------------ repack GGUF blocks -> decode layout (dec_core2.cuh)
// one thread per stored slot (record, a): source = superblock sb of row `row` of the row-major GGUF tensor, or zeros for a slot without one
__device__ __forceinline__ void st16(uint8_t *p, const uint8_t *b) { *(v4u *)p = *(const v4u *)b; }
__device__ __forceinline__ void k4_scale_min(const uint8_t *sc12, uint8_t *sc, uint8_t *mn) {  // get_scale_min_k4 (marlin_gguf_affine_repack.cu:200-210)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < 4) { sc[g] = sc12[g] & 63; mn[g] = sc12[g + 4] & 63; }
    else { sc[g] = (sc12[g + 4] & 15) | ((sc12[g - 4] >> 6) << 4); mn[g] = (sc12[g + 4] >> 4) | ((sc12[g] >> 6) << 4); }
  }
}
template <int TYPE>
__global__ void __launch_bounds__(256) repack2_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long long n, int K, long long nslots) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nslots) return;
  const Geo g = geo_for(K);
  const long long rec = i / g.A;
  const int a = (int)(i % g.A), A = g.A;
  const long long rgi = rec / g.TPC;
  const int ts = (int)(rec % g.TPC);
  const int j = a % g.W, p = (a / g.W) & 3, r = a / (4 * g.W);
  const long long row = rgi * g.R + r;
  const int sbi = ts * g.W + j, sb = p * g.Cs + sbi;
  const bool have = row < n && sbi < g.Cs && sb < g.S;
  uint8_t *rb = dst + rec * rec_bytes(TYPE, g);
  alignas(16) uint8_t buf[16];
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    constexpr int TS = TYPE == T_Q4_K ? 144 : 176, NP = TYPE == T_Q4_K ? 8 : 10;
    const uint8_t *b = src + ((size_t)row * g.S + sb) * TS;
    const uint8_t *qs = b + (TYPE == T_Q5_K ? 48 : 16);
    for (int pi = 0; pi < 8; ++pi) {
      for (int k = 0; k < 16; ++k) buf[k] = have ? (uint8_t)(qs[pi * 16 + k] ^ (TYPE == T_Q4_K ? 0x80 : 0x00)) : 0;
      st16(rb + ((size_t)pi * A + a) * 16, buf);
    }
    if constexpr (TYPE == T_Q5_K) {
      const uint8_t *qh = b + 16;
      for (int half = 0; half < 2; ++half) {
        for (int pi = 0; pi < 4; ++pi) {  // piece i = 4 half + pi: quarter c = i / 2, hp = i & 1: low nibbles = weights c*64 + hp*16 + e, high = + 32; fifth bits = qh[hp*16 + e] bits 2c, 2c+1
          const int ii = 4 * half + pi, c = ii >> 1, hp = ii & 1;
          uint32_t wv = 0;
          if (have)
            for (int k = 0; k < 4; ++k)
              for (int jj = 0; jj < 4; ++jj) {
                const uint32_t v = qh[hp * 16 + 4 * k + jj];
                wv |= ((v >> (2 * c)) & 1u) << (8 * jj + k);
                wv |= ((v >> (2 * c + 1)) & 1u) << (8 * jj + 4 + k);
              }
          *(uint32_t *)(buf + 4 * pi) = wv;
        }
        st16(rb + ((size_t)(8 + half) * A + a) * 16, buf);
      }
    }
    uint8_t sc[8], mn[8];
    if (have) k4_scale_min(b + 4, sc, mn);
    for (int k = 0; k < 8; ++k) { buf[k] = have ? sc[k] : 0; buf[8 + k] = have ? mn[k] : 0; }
    st16(rb + ((size_t)NP * A + a) * 16, buf);
    *(uint32_t *)(rb + (size_t)(NP + 1) * 16 * A + 4 * a) = have ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) : 0u;
  } else if constexpr (TYPE == T_Q6_K) {
    const uint8_t *b = src + ((size_t)row * g.S + sb) * 210;
    const uint8_t *ql = b, *qh = b + 128;
    auto lo4 = [&](int e) { const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32, ii = hh * 64 + pos + (qt % 2) * 32; return qt < 2 ? (ql[ii] & 15) : (ql[ii] >> 4); };
    auto hi2 = [&](int e) { const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32; return (qh[hh * 32 + pos] >> (qt * 2)) & 3; };
    for (int pi = 0; pi < 8; ++pi) {
      for (int k = 0; k < 16; ++k) buf[k] = have ? (uint8_t)(lo4((2 * pi) * 16 + k) | (lo4((2 * pi + 1) * 16 + k) << 4)) : 0;
      st16(rb + ((size_t)pi * A + a) * 16, buf);
    }
    for (int gq = 0; gq < 4; ++gq) {
      for (int k = 0; k < 16; ++k)
        buf[k] = have ? (uint8_t)(hi2((4 * gq) * 16 + k) | (hi2((4 * gq + 1) * 16 + k) << 2) | (hi2((4 * gq + 2) * 16 + k) << 4) | (hi2((4 * gq + 3) * 16 + k) << 6)) : 0;
      st16(rb + ((size_t)(8 + gq) * A + a) * 16, buf);
    }
    for (int k = 0; k < 16; ++k) buf[k] = have ? b[192 + k] : 0;
    st16(rb + ((size_t)12 * A + a) * 16, buf);
    *(uint16_t *)(rb + (size_t)208 * A + 2 * a) = have ? (uint16_t)(b[208] | (b[209] << 8)) : (uint16_t)0;
  } else {  // Q8_0: superblock = blocks 8 sb .. 8 sb + 7 of the row
    const uint8_t *b = src + ((size_t)row * (K / 32) + (size_t)sb * 8) * 34;
    for (int pi = 0; pi < 16; ++pi) {
      for (int k = 0; k < 16; ++k) buf[k] = have ? b[(pi >> 1) * 34 + 2 + (pi & 1) * 16 + k] : 0;
      st16(rb + ((size_t)pi * A + a) * 16, buf);
    }
    for (int k = 0; k < 8; ++k) { buf[2 * k] = have ? b[k * 34] : 0; buf[2 * k + 1] = have ? b[k * 34 + 1] : 0; }
    st16(rb + ((size_t)16 * A + a) * 16, buf);
  }
}

bool make_mat(Mat &m, const void *planes, int type, long long n, long long k) {
  if (!planes || !dec_type(type) || n <= 0 || k <= 0 || k % 256) return false;
  const size_t tb = tensor_bytes(type, n, k);
  if (tb >= 0xffffff00ull) return false;  // one 32-bit buffer descriptor per tensor
  m.base = (const uint8_t *)planes; m.bytes = (unsigned)tb; m.type = type; m.n = (int)n; m.k = (int)k;
  return true;
}

// ------------------------------------------------------------------------------------------------ plain GEMV: out[c][row] = W[row] . act(x[c])
struct PlainArgs {
  Mat m;
  const float *x; int ldx; const float *norm_w; float eps;
  float *out; int out_stride;
  int units, spec, ring;
  unsigned long long *tl;
};
template <int NCOLS, bool SPEC>
__global__ void __launch_bounds__(NT) dec2_plain_kernel(const PlainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8 * 8];
  __shared__ int ctr;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.m.k, mode = act_mode_for(a.m.type);
  Job jb{};
  jb.nseg = 1; jb.rgpu = 1; jb.ring = a.ring; jb.mat[0] = a.m; jb.nrows = a.m.n; jb.tl = a.tl; jb.sel = nullptr; jb.sel_mode = 1; jb.upe = a.units > 0 ? a.units : 1; jb.ergs = 0;
  jb.u0 = (int)((long long)blockIdx.x * a.units / gridDim.x); jb.u1 = (int)((long long)(blockIdx.x + 1) * a.units / gridDim.x);
  auto noaux = [](int, int, int) { return NoAux{}; };
  const Geo g = geo_for(K);
  const int lpr = 4 * g.LPC, rr = lane / lpr;
  auto epi = [&](int, int row0, int nvalid, int, const float(&sum)[NCOLS], const NoAux &) {
    if ((lane & (lpr - 1)) == owner_off(g) && rr < nvalid) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) a.out[(size_t)c * a.out_stride + row0 + rr] = sum[c];
    }
  };
  if constexpr (SPEC) {
    SpecRegs spre;
    auto stage = [&](int st) { if (st == 0) spre = act_issue_spec(a.x, a.norm_w, K, wave); else act_finish_spec<NCOLS>(smem, spre, a.x, a.ldx, a.norm_w, a.eps, K, mode, wave); };
    MRS_DEC2_TYPE_SWITCH(a.m.type, { stream<TT, NCOLS, true>(jb, K, NCOLS, mode, smem, &ctr, stage, noaux, epi); })
  } else {
    ActRegs<2> pre;
    auto stage = [&](int st) { if (st == 0) pre = act_issue_all<2>(a.x, a.norm_w, K); else act_finish_all<NCOLS, 2>(smem, red, pre, a.x, a.ldx, a.norm_w, a.eps, K, mode); };
    MRS_DEC2_TYPE_SWITCH(a.m.type, { stream<TT, NCOLS, false>(jb, K, NCOLS, mode, smem, &ctr, stage, noaux, epi); })
  }
}

}  // namespace dec2
}  // namespace mrs

using namespace mrs;

struct mrs_dec_mat_c2 { const void *planes; int type; long long n, k; };

extern "C" size_t mrs_dec2_repack_bytes(int type, long long n, long long k) {
  if (!dec2::dec_type(type) || k % 256 || n <= 0) return 0;
  return dec2::tensor_bytes(type, n, k);
}
extern "C" int mrs_dec2_repack(const void *gguf_blocks, int type, long long n, long long k, void *planes, void *stream) {
  if (!mrs_dec2_repack_bytes(type, n, k) || !gguf_blocks || !planes) return -1;
  const dec2::Geo g = dec2::geo_for((int)k);
  const long long nslots = ((n + g.R - 1) / g.R) * g.TPC * g.A;
  const dim3 grid((unsigned)((nslots + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
  case T_Q4_K: hipLaunchKernelGGL(dec2::repack2_kernel<T_Q4_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  case T_Q5_K: hipLaunchKernelGGL(dec2::repack2_kernel<T_Q5_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  case T_Q6_K: hipLaunchKernelGGL(dec2::repack2_kernel<T_Q6_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  default: hipLaunchKernelGGL(dec2::repack2_kernel<T_Q8_0>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  }
  return 0;
}
static unsigned long long *g_tl2 = nullptr;
extern "C" void mrs_dec2_timeline(void *buf) { g_tl2 = (unsigned long long *)buf; }
extern "C" int mrs_dec2_gemv(const mrs_dec_mat_c2 *w, const float *x, int ldx, const float *norm_w, float eps, float *out, int ld_out, int b, void *stream) {
  dec2::PlainArgs a{};
  if (!w || !dec2::make_mat(a.m, w->planes, w->type, w->n, w->k)) return -1;
  a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.out = out; a.out_stride = ld_out; a.tl = g_tl2;
  const dec2::Geo g = dec2::geo_for((int)w->k);
  a.units = (int)((w->n + g.R - 1) / g.R);
  static const int gmax = [] { const char *e = getenv("MRS_DEC2_GRID"); return e ? atoi(e) : 256; }();
  const int grid = a.units < gmax ? a.units : gmax;
  const size_t wg_bytes = (size_t)((a.units + grid - 1) / grid) * g.TPC * dec2::rec_bytes(w->type, g);
  { static int force = -2; if (force == -2) { const char *e = getenv("MRS_DEC2_SPEC"); force = e ? atoi(e) : -1; } a.spec = force >= 0 ? force : (wg_bytes > 96 * 1024 ? 1 : 0); }
  { static const int r = [] { const char *e = getenv("MRS_DEC2_RING"); return e ? atoi(e) : 0; }(); a.ring = r > 0 ? r : (a.spec ? 8 : 1); }
  const size_t lds = (dec2::act_bytes((int)w->k, b) + 15) & ~(size_t)15;
  if (lds > 158 * 1024) return -2;
  hipStream_t s = (hipStream_t)stream;
#define MRS_GO(NC) { if (a.spec) { auto kern = dec2::dec2_plain_kernel<NC, true>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(grid), dim3(dec2::NT), lds, s, a); } \
                     else { auto kern = dec2::dec2_plain_kernel<NC, false>; lds_attr_once((const void *)kern, 158 * 1024); hipLaunchKernelGGL(kern, dim3(grid), dim3(dec2::NT), lds, s, a); } }
  switch (b) {
  case 1: MRS_GO(1) break;
  case 2: MRS_GO(2) break;
  case 4: MRS_GO(4) break;
  default: return -1;
  }
#undef MRS_GO
  return 0;
}
