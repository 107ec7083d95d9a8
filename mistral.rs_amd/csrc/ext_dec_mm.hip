// ext_dec_mm.hip -- MI355X decode engine, batched steps (2 .. 8 sequences per step) on the matrix cores.
//
// Reference role: the MMVQ launchers take 1 .. 8 activation columns from ONE pass over the weights (mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:724-792,
// gguf/fast_mmvq.rs:52 MMVQ_MAX_BATCH; GgufMatMul::try_fast_forward, gguf/mod.rs:298-323).  The engine's batched GEMV (dec_core2.cuh, NCOLS = 2 .. 8) does that on the
// vector ALU: 535 VALU + 48 ds_read per tile of 16 superblocks x 8 columns -- at batch 8 the launches are bound by that arithmetic, not by the weight stream
// (profiles/round5_decode.md: 0.13 of the step roofline).  Here the integer dots run on v_mfma_i32_32x32x32_i8 / 32x32x16_i8:
//   * weights: the MFMA-order copy the exact prompt GEMM already keeps in HBM (ext_gemm_qi.hip qi_repack_kernel: 32-row panels, one record per panel and superblock,
//     every lane's 16 bytes of a piece contiguous): a record is 4.6 KB (Q4_K) / 8.6 KB (Q5_K, Q6_K, Q8_0) of consecutive bytes, `nt` loads, 2-3 records in flight per wave;
//   * activations: the image mrs_dec_act_image builds once per phase (int8 quants, block scales, run sums: dec_core2.cuh), copied into LDS by every workgroup; the MFMA's
//     A operand = 8 or 16 int8 of a token read straight from it (tokens are the M rows: lane l supplies token l % 32, only rows < b are real), B = the weight bytes;
//     the accumulator registers 0 .. 3 of a lane are tokens 4 (l / 32) + i of weight row l % 32: four live outputs per lane for b <= 8;
//   * per sub-block the exact int32 dot times the 6 / 8-bit scale (v_mad per live output), mins / offsets as ONE f16 MFMA over the 16 run sums, then the engine's
//     f32 term per superblock -- ORD-U, the order of dec_core2.cuh / ext_gemm_qi.hip / oracle orc_gemv_engine: a token's result is bit for bit what the batch-1 GEMV,
//     the batched VALU GEMV and the prompt GEMM give (tests/test_dec_mm.py);
//   * workgroup = 4 waves = the 4 ORD-U runs of superblocks of ONE 32-row panel at a time; the four run sums meet in LDS (double buffered: one barrier per panel),
//     256 threads = 8 tokens x 32 rows finish ((c0 + c1) + c2) + c3 and run the phase's epilogue (store / residual / GLU / RoPE + paged-cache write).
// The weight stream is what a launch costs: ~4.9 GB per step for the 8B Q4_K_M model whatever b is.
#include "dec_gemv.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace mrs {
namespace mm {
using namespace mrs::dec2;
using mrs::dec::EPI_STORE; using mrs::dec::EPI_RESID; using mrs::dec::EPI_GLU; using mrs::dec::EPI_QKV;
using mrs::dec::TM_Q4K; using mrs::dec::TM_Q5K; using mrs::dec::TM_Q6K; using mrs::dec::TM_Q80; using mrs::dec::tmask_of;

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int MT = 256;                                  // 4 waves
constexpr int REC_Q4K = 4096 + 512 + 128, REC_BIG = 8192 + 512 + 128;  // == ext_gemm_qi.hip rec_bytes_qi
__host__ __device__ constexpr int rec_bytes(int type) { return type == T_Q4_K ? REC_Q4K : REC_BIG; }
__host__ __device__ inline bool mm_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0; }

struct MmTensor { const uint8_t *w; unsigned bytes; int type, npanels, nrows; };
struct MmArgs {
  MmTensor m[3];
  int epi, nseg, units;        // GLU: a unit = panel u of m[0] (gate) then panel u of m[1] (up); QKV: units enumerate the panels of m[0], m[1], m[2]; else panels of m[0]
  int K, nc, mode;             // activation columns (<= 8), ACT_Q8K / ACT_Q80
  const char *img; unsigned gc0p, gnp;  // the image(s) mrs_dec_act_image wrote: column c lives in the group image at act_bytes(K, gc0), gn columns wide (gc0 / gn - 1: 4 bits per column; an
                                       // array indexed by a run-time column would make hipcc copy the whole argument block to scratch memory)
  float *out; int out_stride; float resid_scale; int activation;
  float *q_out; void *k_cache, *v_cache; const int64_t *slot_mapping; const int32_t *positions; const float *cos_t, *sin_t;
  int head_dim, rot_pairs, num_kv_heads, block_size, cache_x, kv_f16, hd_shift, bs_shift, x_shift;
  unsigned long long *tl;  // experiments: 8 s_memrealtime stamps (100 MHz) per wave, or nullptr (mrs_dec_mm_timeline)
};
#ifdef MRS_MM_TIMELINE  // (a conditional store inside the streaming loop makes every wait of the ring stricter: off in the product build)
#define MRS_MM_TL(i) do { if (a.tl && (tid & 63) == 0) a.tl[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MRS_MM_TL(i) do { } while (0)
#endif

// LDS: quants [nc][QCS] (a column = the image's 4 runs of CQ bytes, + 32 bytes of skew: the 8 tokens of a ds_read_b64 lane group land in 8 different bank pairs);
// Q8K mode: yd [Sp][8] f32, bsf [Sp][8][16] f16;  Q80 mode: yd8 [Sp][8 blocks][8] f32;  ex [2][4][8][32] f32
__host__ __device__ inline int qcs_of(int K) { return 4 * act_cq(K) + 32; }
__host__ __device__ inline size_t lds_bytes(int K, int nc, int mode) {
  const size_t sp = (size_t)act_sp(K);
  return (size_t)nc * qcs_of(K) + (mode == ACT_Q80 ? sp * 256 : sp * (32 + 256)) + 2 * 4 * 8 * 32 * 4;
}

template <int NPC> struct Rec { v4u q[NPC]; v4u hs; unsigned hd; };

__device__ __forceinline__ i4v mk_i4(v2u a, v2u b) { return i4v{(int)a.x, (int)a.y, (int)b.x, (int)b.y}; }
__device__ __forceinline__ long mk_l(unsigned lo, unsigned hi) { return (long)(((unsigned long long)hi << 32) | lo); }

// the engine's f32 term T of superblock sb for the lane's four live tokens 4 hf + i (weight row nn of the panel), from the record's registers
template <int TYPE, int NPC>
__device__ __forceinline__ void term(const Rec<NPC> &w, int sb, const char *qcol /* the lane's token column + the superblock's offset */, const char *ydp, const char *bsp,
                                     int hf, float (&T)[4]) {
  const i16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (TYPE == T_Q8_0) {
    // block g: 32 int8 of the weight row x 32 int8 of the token: one MFMA; p_g = ((float)isum dw_g) dx_g, T = p_0 + p_1 + ... left to right (Tile<T_Q8_0>)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      i16v d[4];
      i4v wa[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) wa[h] = *(const i4v *)(qcol + 32 * (4 * g2 + h) + 16 * hf);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 4; ++h) d[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[h], __builtin_bit_cast(i4v, w.q[4 * g2 + h]), zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int g = 4 * g2 + h;
        const unsigned hw = g < 2 ? w.hs.x : g < 4 ? w.hs.y : g < 6 ? w.hs.z : w.hs.w;
        const float dwg = half_bits_to_float((uint16_t)((g & 1) ? (hw >> 16) : (hw & 0xffffu)));
        const float4 y4 = *(const float4 *)(ydp + (g * 8 + 4 * hf) * 4);
        const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = ((float)d[h][i] * dwg) * yy[i];
          T[i] = g == 0 ? p : T[i] + p;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    int isum[4] = {0, 0, 0, 0};
    i16v dq[TYPE == T_Q4_K ? 4 : 1];  // Q4_K: the results of the second group of MFMAs wait here for the mins' MFMA to be issued
    if constexpr (TYPE == T_Q4_K) {
      // piece c: bytes 0 .. 7 = qs[32 c + 8 hf ..], 8 .. 15 = qs[32 c + 16 + 8 hf ..]: low nibbles = elements 64 c + {8 hf .., 16 + 8 hf ..} (sub-block 2 c), high = + 32 (2 c + 1)
      // FOUR independent MFMAs per group, then their scale products (first GPU run: hipcc funnelled every MFMA of a record through one accumulator tuple -- read the
      // four live results back, overwrite -- and a record cost a wave 0.66 us: nine exposed MFMA latencies; the scheduling barriers keep a group's MFMAs together)
      // Schedule of a record (scheduling barriers pin it; one wave per SIMD has nobody to hide a latency behind): operands of the first four sub-blocks | their MFMAs |
      // LDS reads + unpack of the other four and the mins' operand WHILE those run | scale products of the first four | MFMAs of the other four + the mins' MFMA (issued
      // by the caller's code right behind) | scale products.
      auto operands = [&](int g2, i4v (&wa)[4], i4v (&wb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = 2 * g2 + h;
          const v4u lo = w.q[c] & 0x0F0F0F0Fu, hi = (w.q[c] >> 4) & 0x0F0F0F0Fu;
          const char *qa = qcol + 64 * c + 8 * hf;
          wa[2 * h] = mk_i4(*(const v2u *)(qa), *(const v2u *)(qa + 16)); wa[2 * h + 1] = mk_i4(*(const v2u *)(qa + 32), *(const v2u *)(qa + 48));
          wb[2 * h] = __builtin_bit_cast(i4v, lo); wb[2 * h + 1] = __builtin_bit_cast(i4v, hi);
        }
      };
      auto products = [&](unsigned scw, const i16v (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          isum[i] += (__mul24(byte_of(scw, 0), d[0][i]) + __mul24(byte_of(scw, 1), d[1][i])) + (__mul24(byte_of(scw, 2), d[2][i]) + __mul24(byte_of(scw, 3), d[3][i]));
      };
      i4v wa0[4], wb0[4], wa1[4], wb1[4];
      operands(0, wa0, wb0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 4; ++h) dq[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa0[h], wb0[h], zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      operands(1, wa1, wb1);  // in the shadow of the four MFMAs above
      __builtin_amdgcn_sched_barrier(0);
      products(w.hs.x, dq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 4; ++h) dq[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa1[h], wb1[h], zero, 0, 0, 0);
      // (their scale products follow the mins' MFMA below: its operand build runs in the shadow of these four)
    } else if constexpr (TYPE == T_Q5_K) {
      // piece g = sub-block g, the 5-bit values as bytes: bytes 0 .. 7 = elements 32 g + 8 hf .., 8 .. 15 = 32 g + 16 + 8 hf ..
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        i16v d[4];
        i4v wa[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const char *qa = qcol + 32 * (4 * g2 + h) + 8 * hf;
          wa[h] = mk_i4(*(const v2u *)(qa), *(const v2u *)(qa + 16));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) d[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[h], __builtin_bit_cast(i4v, w.q[4 * g2 + h]), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned scw = g2 == 0 ? w.hs.x : w.hs.y;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          isum[i] += (__mul24(byte_of(scw, 0), d[0][i]) + __mul24(byte_of(scw, 1), d[1][i])) + (__mul24(byte_of(scw, 2), d[2][i]) + __mul24(byte_of(scw, 3), d[3][i]));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {  // Q6_K: piece g = runs 2 g, 2 g + 1, the 6-bit values as bytes 0 .. 63 (the - 32 goes through the run sums below): one 32x32x16 MFMA per run of 16
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {  // runs 4 g4 .. 4 g4 + 3 = pieces 2 g4, 2 g4 + 1
        i16v d[4];
        long wa[4], wb[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int r = 4 * g4 + h, g = r >> 1;
          const v2u a8 = *(const v2u *)(qcol + 16 * r + 8 * hf);
          wa[h] = mk_l(a8.x, a8.y);
          wb[h] = (h & 1) == 0 ? mk_l(w.q[g].x, w.q[g].y) : mk_l(w.q[g].z, w.q[g].w);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) d[h] = __builtin_amdgcn_mfma_i32_32x32x16_i8(wa[h], wb[h], zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned scw = g4 == 0 ? w.hs.x : g4 == 1 ? w.hs.y : g4 == 2 ? w.hs.z : w.hs.w;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          isum[i] += (__mul24(sbyte_of(scw, 0), d[0][i]) + __mul24(sbyte_of(scw, 1), d[1][i])) + (__mul24(sbyte_of(scw, 2), d[2][i]) + __mul24(sbyte_of(scw, 3), d[3][i]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the terms over the 16 run sums of the token's superblock: K-quants with mins: M = sum_j m_j bsum_j;  Q6_K: S = sum_r sc_r bsum_r (isum -= 32 S).  One f16 MFMA:
    // A = the run sums (|.| <= 2032: exact in f16), B = the row's 6-bit mins, each twice (two runs per sub-block) / the int8 run scales; products and sums < 2^24: exact
    const h8 bsf = *(const h8 *)(bsp + hf * 16);
    h8 wm;
    if constexpr (TYPE == T_Q6_K) {
      const unsigned s0 = hf ? w.hs.z : w.hs.x, s1 = hf ? w.hs.w : w.hs.y;  // runs 8 hf .. 8 hf + 7
#pragma unroll
      for (int k = 0; k < 4; ++k) { wm[k] = (_Float16)(float)sbyte_of(s0, k); wm[4 + k] = (_Float16)(float)sbyte_of(s1, k); }
    } else {
      const unsigned mw = hf ? w.hs.w : w.hs.z;  // operand slot (hf, jj) <-> run 8 hf + jj -> sub-block 4 hf + jj / 2
#pragma unroll
      for (int k = 0; k < 4; ++k) { const _Float16 mk = (_Float16)(float)byte_of(mw, k); wm[2 * k] = mk; wm[2 * k + 1] = mk; }
    }
    const f16v zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (TYPE == T_Q4_K) __builtin_amdgcn_sched_barrier(0);
    const f16v M = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, wm, zf, 0, 0, 0);
    if constexpr (TYPE == T_Q4_K) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        isum[i] += (__mul24(byte_of(w.hs.y, 0), dq[0][i]) + __mul24(byte_of(w.hs.y, 1), dq[1][i])) + (__mul24(byte_of(w.hs.y, 2), dq[2][i]) + __mul24(byte_of(w.hs.y, 3), dq[3][i]));
    }
    const float4 y4 = *(const float4 *)(ydp + 4 * hf * 4);
    const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
    const float d = half_bits_to_float((uint16_t)(w.hd & 0xffff));
    if constexpr (TYPE == T_Q6_K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) T[i] = (d * yy[i]) * (float)(isum[i] - ((int)M[i] << 5));
    } else {
      const float dmin = half_bits_to_float((uint16_t)(w.hd >> 16));
#pragma unroll
      for (int i = 0; i < 4; ++i) T[i] = fmaf(d * yy[i], (float)isum[i], -((dmin * yy[i]) * M[i]));
    }
  }
}

// tensor index and panel of (unit, segment)
__device__ __forceinline__ void where_of(int epi, int np0, int np1, int u, int seg, int &ti, int &pn) {
  if (epi == EPI_QKV) { ti = u >= np0 + np1 ? 2 : (u >= np0 ? 1 : 0); pn = u - (ti == 2 ? np0 + np1 : (ti == 1 ? np0 : 0)); }
  else { ti = seg; pn = u; }
}
// request the next record of the wave's stream (unit cu, segment cseg, superblock ct of run p) into a ring slot and advance the cursor.  Every request is unconditional and
// asks for the same number of loads (a cursor past its last record asks past the tensor: zeros, no traffic), so hipcc's waits come out exact (dec_core2.cuh stream()).
template <int NPC, bool ONLY4>
__device__ __forceinline__ void issue_rec(Rec<NPC> &slot, int &m, int &cu, int &cseg, int &ct, __amdgpu_buffer_rsrc_t rs0, __amdgpu_buffer_rsrc_t rs1, __amdgpu_buffer_rsrc_t rs2,
                                          int ty0, int ty1, int ty2, int np0, int np1, int epi, int nseg, int units, int ntp, int p, int Cs, int S, int lane, int nn, int step) {
  constexpr unsigned DEAD = 0xF0000000u;
  const bool live = cu < units && ntp > 0;
  int ti = 0, pn = 0;
  where_of(epi, np0, np1, live ? cu : 0, cseg, ti, pn);
  const int type = ti == 0 ? ty0 : (ti == 1 ? ty1 : ty2);
  const unsigned rec = live ? ((unsigned)pn * (unsigned)S + (unsigned)(p * Cs + ct)) * (unsigned)rec_bytes(type) : DEAD;
  const __amdgpu_buffer_rsrc_t rs = ti == 0 ? rs0 : (ti == 1 ? rs1 : rs2);
  const bool small = ONLY4 || type == T_Q4_K;
  const unsigned tail = small ? 4096u : 8192u;
#pragma unroll
  for (int c = 0; c < NPC; ++c) slot.q[c] = ldb128(rs, (c < 4 || !small) ? rec + (unsigned)(c * 64 + lane) * 16u : DEAD);  // a Q4_K record in a mixed launch: pieces 4 .. 7 ask past the tensor
  slot.hs = ldb128(rs, rec + tail + (unsigned)nn * 16u);
  slot.hd = ldb32(rs, rec + tail + 512u + (unsigned)nn * 4u);
  m = live ? (type | (ct << 8) | ((ct == ntp - 1 ? 1 : 0) << 24)) : -1;
  if (live) {  // scalar bookkeeping only
    if (++ct == ntp) { ct = 0; if (++cseg == nseg) { cseg = 0; cu += step; } }
  }
}

// the epilogue of a finished (unit, segment): all 256 threads, thread = (token tid / 32, row tid % 32 of the panel); called behind the barrier that publishes ex[par].
// (A plain function, not a lambda of the kernel: as a by-reference closure its stores kept every local of the kernel -- and a copy of the argument block -- in scratch memory.)
__device__ __forceinline__ void finish_unit(const MmArgs &a, const float *ex, int par, int eu, int eseg, int tid, float &gsave) {
  const int epi = a.epi, nc = a.nc, np0 = a.m[0].npanels, np1 = a.m[1].npanels, nr0 = a.m[0].nrows, nr1 = a.m[1].nrows, nr2 = a.m[2].nrows;

  const int tok = tid >> 5, row = tid & 31;
  const float *e = ex + (size_t)par * 1024 + tok * 32 + row;
  const float sum = ((e[0] + e[256]) + e[512]) + e[768];
  int ti = 0, pn = 0;
  if (epi == EPI_QKV) { ti = eu >= np0 + np1 ? 2 : (eu >= np0 ? 1 : 0); pn = eu - (ti == 2 ? np0 + np1 : (ti == 1 ? np0 : 0)); } else { ti = eseg; pn = eu; }
  const int n = pn * 32 + row;
  const int nrows = ti == 0 ? nr0 : (ti == 1 ? nr1 : nr2);
  const bool ok = tok < nc && n < nrows;
  if (epi == EPI_STORE) { if (ok) a.out[(size_t)tok * a.out_stride + n] = sum; }
  else if (epi == EPI_RESID) { if (ok) { float *o = a.out + (size_t)tok * a.out_stride + n; *o = *o * a.resid_scale + sum * 1.0f; } }
  else if (epi == EPI_GLU) {
    if (eseg == 0) gsave = sum;
    else if (ok) a.out[(size_t)tok * a.out_stride + n] = (a.activation == 0 ? silu_engine(gsave) : glu_act(gsave, a.activation)) * sum;
  } else {  // EPI_QKV, interleaved RoPE: rows 2 i, 2 i + 1 are a pair = lanes l, l ^ 1 of this wave (dec_gemv.cuh EPI_QKV, R >= 2)
    const float other = __shfl_xor(sum, 1, 64);
    if (ok) {
      const int c = tok;
      const bool odd = (n & 1) != 0;
      const int lr = n & ~1, head = lr >> a.hd_shift, dd = lr & (a.head_dim - 1);
      const int pair_i = dd >> 1;
      const bool rot = ti < 2 && pair_i < a.rot_pairs;
      const int pi = min(pair_i, a.rot_pairs - 1);
      const size_t tix = (size_t)a.positions[c] * a.rot_pairs + pi;
      const float cs = a.cos_t[tix], sn = a.sin_t[tix];
      const float ca = rot ? cs : 1.0f, sa = rot ? sn : 0.0f;
      const float xs = odd ? other : sum, ys = odd ? sum : other;
      float x, y;
      rope_pair<float>(xs, ys, ca, sa, x, y);
      const int d0 = dd, d1 = dd + 1;
      if (ti == 0) {
        if (!odd) a.q_out[(size_t)c * nr0 + head * a.head_dim + d0] = x;
        else a.q_out[(size_t)c * nr0 + head * a.head_dim + d1] = y;
      } else {
        const int slot = (int)a.slot_mapping[c];
        if (slot >= 0) {
          const unsigned blk = (unsigned)slot >> a.bs_shift, off = (unsigned)slot & (unsigned)(a.block_size - 1);
          uint16_t *kc = (uint16_t *)a.k_cache, *vc = (uint16_t *)a.v_cache;
          const uint16_t xb = a.kv_f16 ? float_to_half_bits(x) : float_to_bf16_bits(x), yb = a.kv_f16 ? float_to_half_bits(y) : float_to_bf16_bits(y);
          if (ti == 1) {
            const int X = a.cache_x;
            const size_t hb = ((size_t)blk * a.num_kv_heads + head) * (size_t)(a.head_dim >> a.x_shift);
            if (!odd) kc[(hb + ((unsigned)d0 >> a.x_shift)) * a.block_size * X + off * X + ((unsigned)d0 & (unsigned)(X - 1))] = xb;
            else kc[(hb + ((unsigned)d1 >> a.x_shift)) * a.block_size * X + off * X + ((unsigned)d1 & (unsigned)(X - 1))] = yb;
          } else {
            const size_t o = (((size_t)blk * a.num_kv_heads + head) * a.head_dim + dd) * a.block_size + off;
            if (!odd) vc[o] = xb; else vc[o + a.block_size] = yb;
          }
        }
      }
    }
  }
}

template <int TMASK>
__device__ __forceinline__ void dec_mm_body(const MmArgs &a, char *smem) {
  constexpr bool ONLY4 = TMASK == TM_Q4K;
  constexpr int NPC = ONLY4 ? 4 : 8;
  // Records in flight per wave.  Measured and NOT adopted on the launches of <= 256 panels (o_proj, down_proj, q / k / v: one workgroup per CU at best): rings of 8 / 4 records
  // at one wave per SIMD (o_proj 8.1 -> 10.3 us, down_proj 16.2 -> 20.3) and eight waves per workgroup, two per run of superblocks, their terms exchanged through LDS at a
  // barrier per step (o_proj 7.95 -> 8.63, down_proj 15.7 -> 16.2): neither more bytes in flight nor more waves per CU moves these launches -- a CU draws ~25 GB/s through
  // this path and 128 panels occupy 128 CUs (profiles/round6_decode.md section 5).
  constexpr int R = ONLY4 ? 3 : 2;
  const int tid = tid_opaque(), lane = tid & 63, p = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave p = run p of the row's superblocks
  const int nn = lane & 31, hf = lane >> 5;
  const int K = a.K, S = K / 256, Cs = (S + 3) / 4, Sp = 4 * Cs, CQ = act_cq(K), QCS = qcs_of(K), nc = a.nc;
  const int ntp = max(0, min(Cs, S - p * Cs));  // live superblocks of this wave's run
  char *qs = smem;
  char *ydl = smem + (size_t)nc * QCS;                                  // Q8K: [Sp][8] f32;  Q80: [Sp][8][8] f32
  char *bsl = ydl + (a.mode == ACT_Q80 ? 0 : (size_t)Sp * 32);         // Q8K: [Sp][8][16] f16
  float *ex = (float *)(ydl + (a.mode == ACT_Q80 ? (size_t)Sp * 256 : (size_t)Sp * (32 + 256)));
  // ---- the record stream of this wave: unit cu, segment cseg, superblock ct of the run
  // (every field is read into a local first: a select between kernel-argument ADDRESSES makes hipcc copy the whole argument block to scratch memory; the launcher
  // points the unused tensors of a launch at m[0] with 0 bytes)
  const uint8_t *wp0 = a.m[0].w, *wp1 = a.m[1].w, *wp2 = a.m[2].w;
  const unsigned wb0 = a.m[0].bytes, wb1 = a.m[1].bytes, wb2 = a.m[2].bytes;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)wp0, (short)0, (int)wb0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)wp1, (short)0, (int)wb1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)wp2, (short)0, (int)wb2, 0x00020000);
  const int np0 = a.m[0].npanels, np1 = a.m[1].npanels;
  const int ty0 = a.m[0].type, ty1 = a.m[1].type, ty2 = a.m[2].type;
  const int nr0 = a.m[0].nrows, nr1 = a.m[1].nrows, nr2 = a.m[2].nrows;
  const int epi = a.epi, nseg = a.nseg, units = a.units;
  int cu = (int)blockIdx.x, cseg = 0, ct = 0;
  Rec<NPC> ring[R];
  int meta[R];  // type | ct << 8 | last << 24, or -1
  // (issue_rec / where_of are plain functions: as by-reference closures of the kernel their selects between captured variables -- which tensor's resource, type, panel
  // count -- became run-time offsets INTO the closure object, which then could not be dissolved: every local of the kernel lived in scratch memory)
  auto issue = [&](Rec<NPC> &slot, int &m) __attribute__((always_inline)) {
    issue_rec<NPC, ONLY4>(slot, m, cu, cseg, ct, rs0, rs1, rs2, ty0, ty1, ty2, np0, np1, epi, nseg, units, ntp, p, Cs, S, lane, nn, (int)gridDim.x);
  };
  MRS_MM_TL(0);
  // ---- the activation image -> LDS (every workgroup).  The first batch of image requests leaves BEFORE the weight ring: loads return in order, the image comes out of
  // the L2 in ~1 us and is stored while the weights are still on their way from HBM (behind the ring it arrived 2.5 us later: timeline of the second GPU run)
  auto gc0_of = [&](int c) __attribute__((always_inline)) { return (int)((a.gc0p >> (4 * c)) & 15u); };
  auto gn_of = [&](int c) __attribute__((always_inline)) { return (int)((a.gnp >> (4 * c)) & 15u) + 1; };
  // Every request of a stage leaves before the first of its values is stored (batches of up to 16 loads per lane): a load -> store loop is one memory round trip per
  // iteration (hipcc keeps them in order: both pointers are generic), 19 of them in a row at 8 columns of K = 4096 -- ~10 us in front of the first MFMA of every launch
  // (first GPU run of this file: o_proj 9.8 us at every batch size).  Requests go through ONE buffer resource over the image(s): a piece past its column asks past the
  // end (zeros, no branch).
  {
    const unsigned img_bytes = (unsigned)act_bytes(K, nc);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)a.img, (short)0, (int)img_bytes, 0x00020000);
    constexpr unsigned PAST = 0xF0000000u;
    const int nv = 4 * CQ / 16;  // 16-byte pieces of a column
    // quants: wave p copies columns p and p + 4; 8 pieces of each per batch
    const int c0 = p, c1 = p + 4;
    const unsigned s0 = (unsigned)(act_bytes(K, gc0_of(c0)) + (size_t)(c0 - gc0_of(c0)) * 4 * CQ), s1 = (unsigned)(act_bytes(K, gc0_of(c1 & 7)) + (size_t)((c1 & 7) - gc0_of(c1 & 7)) * 4 * CQ);
    const bool h0 = c0 < nc, h1 = c1 < nc;
    v4u r0[8], r1[8];
    auto ld_batch = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = j0 + j * 64 + lane;
        r0[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, h0 && i < nv ? s0 + (unsigned)i * 16u : PAST, 0, 0);
        r1[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, h1 && i < nv ? s1 + (unsigned)i * 16u : PAST, 0, 0);
      }
    };
    auto st_batch = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = j0 + j * 64 + lane;
        if (h0 && i < nv) *(v4u *)(qs + (size_t)c0 * QCS + i * 16) = r0[j];
        if (h1 && i < nv) *(v4u *)(qs + (size_t)c1 * QCS + i * 16) = r1[j];
      }
    };
    // scales and run sums: item -> (superblock, column[, ...]); up to 8 items per thread and batch
    auto col_base = [&](int c) __attribute__((always_inline)) { return (unsigned)(act_bytes(K, gc0_of(c)) + (size_t)gn_of(c) * 4 * CQ); };  // the group's d region
    const bool q80 = a.mode == ACT_Q80;
    const int nsc = q80 ? Sp * 64 : Sp * 32;  // Q80: block scales, image d [ncols][Sp][12] f32 (8 + pad) -> yd8 [sb][block][token];  Q8K: four run sums per item, the
                                              // superblock scale rides along (items with r4 == 0)
    v4u bq[8]; unsigned dv[8];
    auto ld_sc = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * MT + tid;
        if (q80) {
          const int sb = i >> 6, g = (i >> 3) & 7, c = i & 7;
          bq[j] = v4u{0u, 0u, 0u, 0u};
          dv[j] = __builtin_amdgcn_raw_buffer_load_b32(ri, c < nc && i < nsc ? col_base(c) + (unsigned)(((c - gc0_of(c)) * Sp + sb) * 12 + g) * 4u : PAST, 0, 0);
        } else {
          const int sb = i >> 5, c = (i >> 2) & 7, r4 = i & 3;
          const bool ok = c < nc && i < nsc;
          const unsigned db = col_base(c), cb = (unsigned)((c - gc0_of(c)) * Sp + sb);
          bq[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ok ? db + (unsigned)gn_of(c) * (unsigned)Sp * 4u + (cb * ACT_BS + 4u * r4) * 4u : PAST, 0, 0);
          dv[j] = __builtin_amdgcn_raw_buffer_load_b32(ri, ok && r4 == 0 ? db + cb * 4u : PAST, 0, 0);
        }
      }
    };
    auto st_sc = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * MT + tid;
        if (i < nsc) {
          if (q80) ((unsigned *)ydl)[i] = dv[j];
          else {
            const int sb = i >> 5, c = (i >> 2) & 7, r4 = i & 3;
            _Float16 *o = (_Float16 *)(bsl + ((size_t)sb * 8 + c) * 32 + r4 * 8);
            o[0] = (_Float16)(float)(int)bq[j].x; o[1] = (_Float16)(float)(int)bq[j].y; o[2] = (_Float16)(float)(int)bq[j].z; o[3] = (_Float16)(float)(int)bq[j].w;
            if (r4 == 0) ((unsigned *)ydl)[sb * 8 + c] = dv[j];
          }
        }
      }
    };
    // Order: every image batch but the last of each kind is requested and stored; the LAST quant batch and the LAST scale batch are requested, then the weight ring, then
    // they are stored (their values come out of the L2 while the weights are on their way).  The ring must be the last thing requested in front of the streaming loop:
    // with image requests behind it hipcc's waits inside the loop came out as vmcnt(0) / (6) / (12) for the three ring slots instead of (12) each (the third GPU run:
    // 0.65 us per record and wave, one record in flight).
    const int qlast = ((nv - 1) / (64 * 8)) * (64 * 8), slast = ((nsc - 1) / (MT * 8)) * (MT * 8);
    for (int j0 = 0; j0 < qlast; j0 += 64 * 8) { ld_batch(j0); st_batch(j0); }
    for (int i0 = 0; i0 < slast; i0 += MT * 8) { ld_sc(i0); st_sc(i0); }
    ld_batch(qlast);
    ld_sc(slast);
#pragma unroll
    for (int i = 0; i < R; ++i) issue(ring[i], meta[i]);  // the weight ring
    MRS_MM_TL(1);
    st_batch(qlast);
    st_sc(slast);
  }
  MRS_MM_TL(2);
  __syncthreads();
  MRS_MM_TL(3);
  // ---- per-lane constants: the token column this lane supplies as the MFMA's A row (rows >= nc repeat column 0: their results are never read)
  const int tcol = nn < nc ? nn : 0;
  const char *qlane = qs + (size_t)tcol * QCS + (size_t)p * CQ;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float gsave = 0.f;
  int eu = (int)blockIdx.x, eseg = 0, par = 0;  // the (unit, segment) whose run sums are being collected
  auto finish = [&]() __attribute__((always_inline)) {
    finish_unit(a, ex, par, eu, eseg, tid, gsave);
    if (++eseg == nseg) { eseg = 0; eu += (int)gridDim.x; }
    par ^= 1;
  };
  auto publish = [&](const float (&v)[4]) __attribute__((always_inline)) {  // this wave's run sums of the finished (unit, segment) -> ex[par][p][token][row]
#pragma unroll
    for (int i = 0; i < 4; ++i) ex[(size_t)par * 1024 + p * 256 + (4 * hf + i) * 32 + nn] = v[i];
  };
  if (ntp == 0) {  // a run without superblocks contributes + 0 (rows of fewer than four superblocks): no records, the barriers and its share of the epilogues
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    while (eu < units) { publish(z); __syncthreads(); finish(); }
    return;
  }
  auto compute = [&](const Rec<NPC> &w, int m) __attribute__((always_inline)) -> bool {
    if (m < 0) return false;  // wave-uniform
    const int type = m & 0xff, t = (m >> 8) & 0xffff, sb = p * Cs + t;
    float T[4];
    const char *qcol = qlane + t * 256;
    const char *ydp = ydl + (size_t)sb * (a.mode == ACT_Q80 ? 256 : 32);
    const char *bsp = bsl + ((size_t)sb * 8 + (tcol & 7)) * 32;
    switch (type) {
    case T_Q4_K: if constexpr ((TMASK & TM_Q4K) != 0) term<T_Q4_K, NPC>(w, sb, qcol, ydp, bsp, hf, T); break;
    case T_Q5_K: if constexpr ((TMASK & TM_Q5K) != 0) term<T_Q5_K, NPC>(w, sb, qcol, ydp, bsp, hf, T); break;
    case T_Q6_K: if constexpr ((TMASK & TM_Q6K) != 0) term<T_Q6_K, NPC>(w, sb, qcol, ydp, bsp, hf, T); break;
    default: if constexpr ((TMASK & TM_Q80) != 0) term<T_Q8_0, NPC>(w, sb, qcol, ydp, bsp, hf, T); break;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = t == 0 ? T[i] : acc[i] + T[i];
    return (m >> 24) != 0;  // the run of this (unit, segment) is complete
  };
  while (eu < units) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (compute(ring[i], meta[i])) { MRS_MM_TL(4); publish(acc); __syncthreads(); MRS_MM_TL(5); finish(); MRS_MM_TL(6); }
      issue(ring[i], meta[i]);
    }
  }
}

#if defined(__HIPCC__) || defined(__HIP__)
#define MRS_MM_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#else
#define MRS_MM_WAVES_PER_EU(lo, hi)  // (the kernel sources also compile for the host: test infrastructure)
#endif
// The kernels: Q4_K alone (4.6 KB records, three in flight) fits two workgroups per CU in 256 registers; the 8.6 KB records of the other formats (two in flight + four
// accumulator tuples) take one workgroup's worth -- asking for two spilled 26 registers into the streaming loop.  Without a cap on waves per SIMD hipcc schedules for
// three and funnels every MFMA of a record through one accumulator tuple (exposed MFMA latency x 9 per record).
__global__ void __launch_bounds__(MT) MRS_MM_WAVES_PER_EU(2, 2) dec_mm_kernel_q4k(const MmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dec_mm_body<TM_Q4K>(a, smem);
}
template <int TMASK>
__global__ void __launch_bounds__(MT) MRS_MM_WAVES_PER_EU(1, 2) dec_mm_kernel(const MmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dec_mm_body<TMASK>(a, smem);
}
static unsigned long long *g_mm_tl = nullptr;
static unsigned long long g_mm_launches = 0;  // launches of this route since the library was loaded (tests: did a batched step really take it?)
static int launch(const MmArgs &a0, hipStream_t s) {
  MmArgs a = a0;
  a.tl = g_mm_tl;
  int tmask = 0;
  const int nt = a.epi == EPI_QKV ? 3 : (a.epi == EPI_GLU ? 2 : 1);
  for (int i = 0; i < nt; ++i) {
    if (!mm_type(a.m[i].type) || !a.m[i].w) return -1;
    if (act_mode_for(a.m[i].type) != a.mode) return -1;
    tmask |= tmask_of(a.m[i].type);
  }
  for (int i = nt; i < 3; ++i) { a.m[i] = a.m[0]; a.m[i].bytes = 0; a.m[i].npanels = 0; }
  if (a.K <= 0 || a.K % 256 || a.nc < 1 || a.nc > 8 || !a.img || a.units < 1) return -1;
  const size_t lds = (lds_bytes(a.K, a.nc, a.mode) + 15) & ~(size_t)15;
  if (lds > (size_t)158 * 1024) return -2;
  static const int wg_per_cu = [] { const char *e = getenv("MRS_DEC_MM_WG_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();
  const int fit = (tmask == TM_Q4K || tmask == TM_Q5K || tmask == TM_Q80) ? 2 : 1;  // workgroups per CU the kernel's registers admit (see the kernels above)
  const int cap = 256 * std::min<int>(std::min(wg_per_cu, fit), std::max<size_t>(1, ((size_t)160 * 1024) / lds));
  const int grid = std::min(a.units, cap);
  auto go = [&](auto kern) {
    ++g_mm_launches;
    lds_attr_once((const void *)kern, 158 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MT), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -4;
  };
  switch (tmask) {
  case TM_Q4K: return go(dec_mm_kernel_q4k);
  case TM_Q6K: return go(dec_mm_kernel<TM_Q6K>);
  case TM_Q5K: return go(dec_mm_kernel<TM_Q5K>);
  case TM_Q80: return go(dec_mm_kernel<TM_Q80>);
  case TM_Q4K | TM_Q6K: return go(dec_mm_kernel<TM_Q4K | TM_Q6K>);
  case TM_Q5K | TM_Q6K: return go(dec_mm_kernel<TM_Q5K | TM_Q6K>);
  default: return go(dec_mm_kernel<TM_Q4K | TM_Q5K | TM_Q6K>);
  }
}

// the column groups mrs_dec_act_image wrote (ext_dec.hip col_groups: halved until a group's image fits the GEMV kernels' LDS)
static void col_groups(int K, int c0, int b, int *gc0, int *gn) {
  if (b > 1 && act_bytes(K, b) > (size_t)158 * 1024) { col_groups(K, c0, b / 2, gc0, gn); col_groups(K, c0 + b / 2, b - b / 2, gc0, gn); return; }
  for (int c = c0; c < c0 + b; ++c) { gc0[c] = c0; gn[c] = b; }
}
static bool set_tensor(MmTensor &t, const void *qi, int type, long long n, long long k) {
  if (!qi || !mm_type(type) || n <= 0 || k <= 0 || k % 256) return false;
  const size_t bytes = (size_t)((n + 31) / 32) * (size_t)(k / 256) * rec_bytes(type);
  if (bytes >= 0xF0000000ull) return false;
  t.w = (const uint8_t *)qi; t.bytes = (unsigned)bytes; t.type = type; t.npanels = (int)((n + 31) / 32); t.nrows = (int)n;
  return true;
}
static bool base(MmArgs &a, const void *x_img, int k, int b, int type0) {
  if (!x_img || b < 1 || b > 8 || ((uintptr_t)x_img & 15)) return false;
  a.K = k; a.nc = b; a.img = (const char *)x_img; a.mode = act_mode_for(type0);
  int gc0[8] = {0}, gn[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  col_groups(k, 0, b, gc0, gn);
  for (int c = 0; c < b; ++c) { a.gc0p |= (unsigned)gc0[c] << (4 * c); a.gnp |= (unsigned)(gn[c] - 1) << (4 * c); }
  return true;
}

}  // namespace mm
}  // namespace mrs

using namespace mrs;
using namespace mrs::mm;

extern "C" unsigned long long mrs_dec_mm_launch_count(void) { return g_mm_launches; }
extern "C" void mrs_dec_mm_timeline(void *buf) { g_mm_tl = (unsigned long long *)buf; }  // experiments: [grid * 4 waves][8] u64 stamps of the next launches, or NULL
// Can the batched matrix-core route take a launch of this weight type / reduction length / column count?  (K-quants and Q8_0; the LDS budget bounds k x b)
extern "C" int mrs_dec_mm_supported(int type, int k, int b) {
  if (!mm_type(type) || k <= 0 || k % 256 || b < 1 || b > 8) return 0;
  return lds_bytes(k, b, dec2::act_mode_for(type)) + 15 <= (size_t)158 * 1024 ? 1 : 0;
}
// out [b][ld_out] = W . x (mode 0) or out * resid_scale + W . x (mode 1); qi = the MFMA-order copy of W (mrs_gemm_qi_repack), x_img = mrs_dec_act_image(.., type, b, ..)
extern "C" int mrs_dec_mm_proj(const void *qi, int type, int n, int k, const void *x_img, float *out, int ld_out, int mode, float resid_scale, int b, void *stream) {
  MmArgs a{};
  if (!out || !set_tensor(a.m[0], qi, type, n, k) || !base(a, x_img, k, b, type)) return -1;
  a.epi = mode ? EPI_RESID : EPI_STORE; a.nseg = 1; a.units = a.m[0].npanels; a.out = out; a.out_stride = ld_out; a.resid_scale = resid_scale;
  return launch(a, (hipStream_t)stream);
}
// act_out [b][ld_out] = act(W_g . x) * (W_u . x)
extern "C" int mrs_dec_mm_gate_up(const void *qi_gate, const void *qi_up, int type, int n, int k, const void *x_img, int activation, float *act_out, int ld_out, int b,
                                  void *stream) {
  MmArgs a{};
  if (!act_out || !set_tensor(a.m[0], qi_gate, type, n, k) || !set_tensor(a.m[1], qi_up, type, n, k) || !base(a, x_img, k, b, type)) return -1;
  a.epi = EPI_GLU; a.nseg = 2; a.units = a.m[0].npanels; a.out = act_out; a.out_stride = ld_out; a.activation = activation;
  return launch(a, (hipStream_t)stream);
}
// q / k / v projections + interleaved RoPE + paged-cache write (mrs_dec_qkv_img's contract, neox = 0 only: a rotate-half pair spans two panels)
extern "C" int mrs_dec_mm_qkv(const void *qi_q, int type_q, int nq, const void *qi_k, int type_k, int nk, const void *qi_v, int type_v, int nv, int k, const void *x_img,
                              float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t,
                              int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream) {
  MmArgs a{};
  if (!q_out || !k_cache || !v_cache || !slot_mapping || !positions || !cos_t || !sin_t) return -1;
  if (!set_tensor(a.m[0], qi_q, type_q, nq, k) || !set_tensor(a.m[1], qi_k, type_k, nk, k) || !set_tensor(a.m[2], qi_v, type_v, nv, k) || !base(a, x_img, k, b, type_q)) return -1;
  if (((nq | nk | nv | head_dim) & 1) || (kv_dtype != 0 && kv_dtype != 1)) return -1;
  auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  a.epi = EPI_QKV; a.nseg = 1; a.units = a.m[0].npanels + a.m[1].npanels + a.m[2].npanels;
  a.q_out = q_out; a.k_cache = k_cache; a.v_cache = v_cache; a.slot_mapping = slot_mapping; a.positions = positions; a.cos_t = cos_t; a.sin_t = sin_t;
  a.head_dim = head_dim; a.rot_pairs = rot_pairs; a.num_kv_heads = num_kv_heads; a.block_size = block_size; a.cache_x = 8; a.kv_f16 = kv_dtype == 0;
  a.hd_shift = lg2(head_dim); a.bs_shift = lg2(block_size); a.x_shift = lg2(a.cache_x);
  if (a.hd_shift < 0 || a.bs_shift < 0 || head_dim < a.cache_x || rot_pairs < 1) return -1;
  return launch(a, (hipStream_t)stream);
}
