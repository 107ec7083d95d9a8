// ext_dec.hip -- MI355X decode engine: the GEMV phases of a decode step on the core of dec_core2.cuh, the decode-layout repack, the decode attention
// launch, and the C entry points (include/mrs_hip_ext.h, mrs_dec_*).
//
// A decode step of one Llama layer is five launches (reference call sequence: mistralrs-core/src/models/llama.rs:68-157, 243-260):
//   qkv      RMSNorm(h) -> Q8_K/Q8_0 activations in LDS -> q, k, v GEMV rows -> RoPE -> q (f32), k / v into the paged cache
//   attn     decode attention over the cache (dec_attn.cuh): splits + last-arriver merge + Q8_K image of the result
//   o_proj   image -> GEMV -> h = h * s + W_o . attn
//   gate/up  RMSNorm(h) -> quantize -> gate and up rows -> act(gate) * up (f32)
//   down     quantize(act) -> GEMV -> h = h * s + W_d . act
// Every GEMV phase is the same kernel body: request the activation vector, request the first weight records, run the activation prologue (norm + quantize),
// stream the workgroup's units, apply the phase's epilogue per finished row.  Arithmetic and f32 order: header of dec_core2.cuh.
#include "dec_attn.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include "../../include/mistralrs_paged_attn.h"

#include "dec_gemv.cuh"

#ifndef MRS_WAIT_VMCNT0
#define MRS_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

namespace mrs {
namespace dec {

// ------------------------------------------------------------------------------------------------ repack GGUF blocks -> decode layout (dec_core2.cuh)
// one thread per (tile, lane): lane (r, p, c) of tile t of row group rg <- quarter c of superblock p * Cs + t of row 4 rg + r of the row-major GGUF tensor,
// or zeros where there is none (rows past n, the short tail of the last chunks)
__device__ __forceinline__ void st16(uint8_t *p, const uint8_t *b) { *(v4u *)p = *(const v4u *)b; }
__device__ __forceinline__ void k4_scale_min(const uint8_t *sc12, uint8_t *sc, uint8_t *mn) {  // get_scale_min_k4 (marlin_gguf_affine_repack.cu:200-210)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < 4) { sc[g] = sc12[g] & 63; mn[g] = sc12[g + 4] & 63; }
    else { sc[g] = (sc12[g + 4] & 15) | ((sc12[g - 4] >> 6) << 4); mn[g] = (sc12[g + 4] >> 4) | ((sc12[g] >> 6) << 4); }
  }
}
template <int TYPE>
__global__ void __launch_bounds__(256) repack_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long long n, int K, long long nlanes) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nlanes) return;
  const Geo g = geo_for(K);
  const long long tile = i >> 6;
  const int L = (int)(i & 63), r = L >> 4, p = (L >> 2) & 3, c = L & 3, slot = L >> 2;
  const long long rg = tile / g.Cs;
  const int t = (int)(tile - rg * g.Cs);
  const long long row = rg * g.R + r;
  const int sb = p * g.Cs + t;
  const bool have = row < n && sb < g.S;
  uint8_t *tb = dst + (size_t)tile * tile_bytes(TYPE);
  alignas(16) uint8_t buf[16];
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    constexpr int TS = TYPE == T_Q4_K ? 144 : 176;
    constexpr unsigned HS = TYPE == T_Q4_K ? 2048u : 2560u, HD = TYPE == T_Q4_K ? 2304u : 2816u;
    const uint8_t *b = src + ((size_t)row * g.S + sb) * TS;
    const uint8_t *qs = b + (TYPE == T_Q5_K ? 48 : 16);
    for (int h = 0; h < 2; ++h) {  // pieces 2c + h of the superblock's qs
      for (int k = 0; k < 16; ++k) buf[k] = have ? (uint8_t)(qs[32 * c + 16 * h + k] ^ (TYPE == T_Q4_K ? 0x80 : 0x00)) : 0;
      st16(tb + (size_t)h * 1024 + (size_t)L * 16, buf);
    }
    if constexpr (TYPE == T_Q5_K) {
      const uint8_t *qh = b + 16;
      for (int h = 0; h < 2; ++h) {  // piece 2c + h: low nibbles = weights 64 c + 16 h + e, high = + 32; fifth bits = qh[16 h + e] bits 2c, 2c + 1
        uint32_t wv = 0;
        if (have)
          for (int k = 0; k < 4; ++k)
            for (int jj = 0; jj < 4; ++jj) {
              const uint32_t v = qh[h * 16 + 4 * k + jj];
              wv |= ((v >> (2 * c)) & 1u) << (8 * jj + k);
              wv |= ((v >> (2 * c + 1)) & 1u) << (8 * jj + 4 + k);
            }
        *(uint32_t *)(tb + 2048 + (size_t)L * 8 + 4 * h) = wv;
      }
    }
    uint8_t sc[8], mn[8];
    if (have) k4_scale_min(b + 4, sc, mn);
    *(uint32_t *)(tb + HS + (size_t)L * 4) = have ? ((uint32_t)sc[2 * c] | ((uint32_t)sc[2 * c + 1] << 8) | ((uint32_t)mn[2 * c] << 16) | ((uint32_t)mn[2 * c + 1] << 24)) : 0u;
    if (c == 0) *(uint32_t *)(tb + HD + (size_t)slot * 4) = have ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) : 0u;
  } else if constexpr (TYPE == T_Q6_K) {
    const uint8_t *b = src + ((size_t)row * g.S + sb) * 210;
    const uint8_t *ql = b, *qh = b + 128;
    auto lo4 = [&](int e) { const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32, ii = hh * 64 + pos + (qt % 2) * 32; return qt < 2 ? (ql[ii] & 15) : (ql[ii] >> 4); };
    auto hi2 = [&](int e) { const int hh = e / 128, pos = e % 32, qt = (e % 128) / 32; return (qh[hh * 32 + pos] >> (qt * 2)) & 3; };
    for (int h = 0; h < 2; ++h) {  // runs 4c + 2h (low nibble), 4c + 2h + 1 (high nibble)
      for (int k = 0; k < 16; ++k) buf[k] = have ? (uint8_t)(lo4((4 * c + 2 * h) * 16 + k) | (lo4((4 * c + 2 * h + 1) * 16 + k) << 4)) : 0;
      st16(tb + (size_t)h * 1024 + (size_t)L * 16, buf);
    }
    for (int k = 0; k < 16; ++k)
      buf[k] = have ? (uint8_t)(hi2((4 * c) * 16 + k) | (hi2((4 * c + 1) * 16 + k) << 2) | (hi2((4 * c + 2) * 16 + k) << 4) | (hi2((4 * c + 3) * 16 + k) << 6)) : 0;
    st16(tb + 2048 + (size_t)L * 16, buf);
    *(uint32_t *)(tb + 3072 + (size_t)L * 4) = have ? ((uint32_t)b[192 + 4 * c] | ((uint32_t)b[193 + 4 * c] << 8) | ((uint32_t)b[194 + 4 * c] << 16) | ((uint32_t)b[195 + 4 * c] << 24)) : 0u;
    if (c == 0) *(uint16_t *)(tb + 3328 + (size_t)slot * 2) = have ? (uint16_t)(b[208] | (b[209] << 8)) : (uint16_t)0;
  } else {  // Q8_0: superblock = blocks 8 sb .. 8 sb + 7 of the row; quarter c = blocks 2c, 2c + 1
    const uint8_t *b = src + ((size_t)row * (K / 32) + (size_t)sb * 8 + 2 * c) * 34;
    for (int pi = 0; pi < 4; ++pi) {
      for (int k = 0; k < 16; ++k) buf[k] = have ? b[(pi >> 1) * 34 + 2 + (pi & 1) * 16 + k] : 0;
      st16(tb + (size_t)pi * 1024 + (size_t)L * 16, buf);
    }
    *(uint32_t *)(tb + 4096 + (size_t)L * 4) = have ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[34] << 16) | ((uint32_t)b[35] << 24)) : 0u;
  }
}

static bool make_mat(Mat &m, const void *planes, int type, long long n, long long k) {
  if (!planes || !dec_type(type) || n <= 0 || k <= 0 || k % 256) return false;
  const size_t tb = tensor_bytes(type, n, k);
  if (tb >= 0xF0000000ull) return false;  // one 32-bit buffer descriptor per tensor; offsets from 0xF0000000 on are the out-of-range requests of dec_core2.cuh stream()
  m.base = (const uint8_t *)planes; m.bytes = (unsigned)tb; m.type = type; m.n = (int)n; m.k = (int)k;
  return true;
}


// ------------------------------------------------------------------------------------------------ decode attention, split + last-arriver merge
// Grid (kv heads, sequences, splits), G waves per workgroup = the G query heads of ONE split (round 5; rounds 3-4: 4 splits per workgroup, all G heads per
// wave), no barrier in the split phase.  TICKET variant: instead of a second launch for the merge (4.9 us + a kernel boundary per layer), every workgroup publishes its
// partials write-through at agent scope, drains its stores and takes a ticket on the (sequence, kv head) counter; the workgroup that draws the
// last ticket merges all G heads of the kv head: wave w = query heads 2w, 2w + 1 (256 output values = ONE Q8_K superblock of the attention
// vector), lane = 4 consecutive dims, sequential over the splits (attn_merge_core's order).  It writes the f32 result and -- what o_proj's
// prologue would otherwise recompute in all 256 workgroups -- the Q8_K activation image that dec_gemv_kernel copies into LDS (quantize4: same
// lane <-> element mapping as the prologue, so the same bits).  The counter resets itself; hand-off recipe: MI355X guide, "handoff-flag"
// (sc1 payload -> vmcnt(0) -> agent atomic; consumer: returned atomic -> sc1 loads).
struct Attn2Args {
  AttnArgs t;
  unsigned *ticket;  // [seqs][kv heads], zero at rest
  uint8_t *img;      // Q8_K image of [seqs] columns of num_heads * 128 values, or nullptr (odd GQA groups: the caller's o_proj quantizes)
};
// merge of one (sequence, kv head): wave w <-> query heads (2w, 2w + 1) of the group (G == 1: wave 0, one head in lanes 0..31), lane = 4 consecutive
// dims, sequential over the splits.  AGENT: the partials were published by other workgroups of the SAME launch (sc1 loads); else plain loads.
template <int G, bool AGENT>
__device__ __forceinline__ void attn2_merge(const Attn2Args &a, int kvh, int seq, int ns, int wave, int lane, int ncols) {
  constexpr int HD = 128, NP = (G + 1) / 2;
  const AttnArgs &t = a.t;
  if (wave >= NP) return;
  const int head0 = kvh * G, hsel = lane >> 5, head = head0 + 2 * wave + hsel;
  const bool live = 2 * wave + hsel < G;
  const size_t pA = ((size_t)seq * t.num_heads + head0 + 2 * wave) * t.max_splits;                 // partials of head A (lanes 0..31)
  const size_t pB = ((size_t)seq * t.num_heads + head0 + min(2 * wave + 1, G - 1)) * t.max_splits;  // head B (lanes 32..63)
  auto ldw = [&](const float *p) { if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p; };
  const float mA = lane < ns ? ldw(t.part_m + pA + lane) : -FLT_MAX, lA = lane < ns ? ldw(t.part_l + pA + lane) : 0.f;
  const float mB = lane < ns ? ldw(t.part_m + pB + lane) : -FLT_MAX, lB = lane < ns ? ldw(t.part_l + pB + lane) : 0.f;
  // one descriptor for both heads' partials (wave-uniform); the lanes of head B add its distance.  The partial outputs of up to 16 splits are
  // requested before anything is consumed
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(t.part_o + pA * HD), (short)0, (int)((pB - pA + ns) * HD * 4), 0x00020000);
  const unsigned off0 = (unsigned)(lane & 31) * 16u + (hsel ? (unsigned)((pB - pA) * HD * 4) : 0u);
  constexpr int MB = 16;
  constexpr int AUX = AGENT ? 16 : 0;
  v4u r[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, off0 + (unsigned)i * 512u, 0, AUX);  // past the last split of head B: out of range, zeros
  const float wA = fast_exp_ref(mA - dec2::wave_max_all(mA)), wB = fast_exp_ref(mB - dec2::wave_max_all(mB));  // DPP + readlane (a maximum is exact in any order)
  float s_all = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto step = [&](int j, v4u raw) {
    // split j's weight and sum live in lane j: v_readlane (a few cycles, j is wave-uniform) instead of four ds_bpermute round trips per split -- the merge is a
    // serial chain behind the last ticket, and 18 splits x 4 LDS exchanges were ~1.5 us of it (round 5)
    const float wAj = dec2::rlf(wA, j), wBj = dec2::rlf(wB, j), lAj = dec2::rlf(lA, j), lBj = dec2::rlf(lB, j);
    const float wj = hsel ? wBj : wAj, lj = hsel ? lBj : lAj;
    const float lw = lj * wj;
    s_all = s_all + lw;
    const float4 o = as_f4(raw);
    const float t0 = o.x * wj, t1 = o.y * wj, t2 = o.z * wj, t3 = o.w * wj;
    acc.x = acc.x + t0; acc.y = acc.y + t1; acc.z = acc.z + t2; acc.w = acc.w + t3;
  };
  for (int j0 = 0; j0 < ns; j0 += MB) {
    if (j0 > 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, off0 + (unsigned)(j0 + i) * 512u, 0, AUX);
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
      if (j0 + i < ns) step(j0 + i, r[i]);  // wave-uniform
  }
  const float inv = 1.0f / s_all;
  const float4 v = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (t.out && live) *(float4 *)(t.out + ((size_t)seq * t.num_heads + head) * HD + (lane & 31) * 4) = v;
  if (a.img) {  // G even: the wave's 256 values = superblock (head0 + 2 wave) / 2 of column seq
    const int K = t.num_heads * HD, sb = (head0 + 2 * wave) >> 1;
    quantize_sb(v, sb, seq, ACT_Q8K, (char *)a.img, K, ncols);
  }
}

// TICKET: one launch (the last workgroup of a (sequence, kv head) merges); else the split phase only and dec_attn2_merge_kernel follows.
// Round 5 geometry: grid (kv heads, sequences, splits), G waves per workgroup: wave g = query head head0 + g of ONE split (round 4: 4 splits per workgroup, every wave
// all G heads: 56 workgroups of 329 VGPRs at a 512-token context).  The per-head arithmetic is the same function (attn_split_core<1>): same bits; the G waves read the
// same 16 KB of K / V (L1 / L2 hits), every wave's serial chain is 1 / G as long, and a 512-token context fills 8 x 24 = 192 CUs.
template <int G, class CT, bool TICKET>
__global__ void __launch_bounds__(64 * G) dec_attn2_kernel(const Attn2Args a) {
  constexpr int HD = 128;
  __shared__ __attribute__((aligned(16))) float q_s[G][HD + 32];
  __shared__ int last_s;
  const AttnArgs &t = a.t;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kvh = blockIdx.x, seq = blockIdx.y;
  const int split = blockIdx.z, head0 = kvh * G;
  // the first page index of the split does not depend on the context length: both scalar loads leave together (they used to be a chain)
  const unsigned blk_first = t.block_tables[(size_t)seq * t.max_blocks_per_seq + min(split * t.bpw, t.max_blocks_per_seq - 1)];
  const int nblk = ((int)t.context_lens[seq] + 31) / 32;
  const int ns = (nblk + t.bpw - 1) / t.bpw;
  if (split >= ns) return;  // workgroup-uniform: no split of this sequence lands here
  {
    const int b0 = split * t.bpw, b1 = min(b0 + t.bpw, nblk);
    const int ctx_w = (int)t.context_lens[seq], lo_w = t.window > 0 && ctx_w > t.window ? ctx_w - t.window : 0;
    auto publish = [&](int, float o0, float o1, float m, float l) {
      const size_t pi = ((size_t)seq * t.num_heads + head0 + wave) * t.max_splits + split;
      if constexpr (TICKET) {
        __hip_atomic_store(t.part_o + pi * HD + lane, o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.part_o + pi * HD + lane + 64, o1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
          __hip_atomic_store(t.part_m + pi, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(t.part_l + pi, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        t.part_o[pi * HD + lane] = o0;
        t.part_o[pi * HD + lane + 64] = o1;
        if (lane == 0) { t.part_m[pi] = m; t.part_l[pi] = l; }
      }
    };
    if (b1 * 32 <= lo_w) {  // every block of the split lies before the sliding window: the partial a fully masked pass would give, without reading K / V
      publish(0, 0.f, 0.f, -FLT_MAX, 0.f);
    } else {
      attn_split_core<1, CT>(t, kvh, head0 + wave, seq, b0, b1, q_s[wave], q_s[wave] + HD, publish, (int)blk_first);
    }
  }
  if constexpr (TICKET) {
    MRS_WAIT_VMCNT0();  // this wave's partials have left the CU
    __syncthreads();
    unsigned *tk = a.ticket + (size_t)seq * t.num_kv_heads + kvh;
    // The ticket's ordering is a build-time choice (MRS_DEC_ATTN_TICKET_ORDER, default __ATOMIC_RELAXED since round 5).  The hand-off is the guide's R1 form
    // (MI355X_MICROARCH.md, "Valid forms": sc1 write-through payload stores -> drained vmcnt(0) -> agent-scope flag; consumer: returned atomic -> sc1 loads, which never
    // read the CU's L1): the payload is already in memory when the ticket is drawn, and the merge reads it past L1.  An ACQ_REL ticket (rounds 3-4) adds
    // `buffer_wbl2 sc1` + `buffer_inv sc1` around the atomic in EVERY workgroup -- ~1.7 us each by the guide's price list, on the serial chain of the last arriver --
    // and orders nothing this protocol relies on (no dirty lines to write back, no L1-served load to invalidate for).
#ifndef MRS_DEC_ATTN_TICKET_ORDER
#define MRS_DEC_ATTN_TICKET_ORDER __ATOMIC_RELAXED
#endif
    if (tid == 0) last_s = __hip_atomic_fetch_add(tk, 1u, MRS_DEC_ATTN_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ns - 1);
    __syncthreads();
    if (!last_s) return;
    if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every other workgroup of this (seq, kv head) has already drawn
    attn2_merge<G, true>(a, kvh, seq, ns, wave, lane, gridDim.y);
  }
}
// the merge as its own launch: grid (kv heads, sequences), (G + 1) / 2 waves
template <int G>
__global__ void __launch_bounds__(64 * ((G + 1) / 2)) dec_attn2_merge_kernel(const Attn2Args a) {
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = ((int)a.t.context_lens[blockIdx.y] + 31) / 32;
  attn2_merge<G, false>(a, blockIdx.x, blockIdx.y, (nblk + a.t.bpw - 1) / a.t.bpw, wave, lane, gridDim.y);
}


// ------------------------------------------------------------------------------------------------ the activation image as its own launch (batched decode)
// A GEMV workgroup builds the image of ALL activation columns before it streams: at 8 columns that is 16 (K = 4096) to 28 (K = 14336, 4 columns per launch)
// superblocks per wave, 256 times over -- 17-25 us of a 40-60 us launch (rocprofv3, round 5).  dec_act_image_kernel builds it ONCE, one workgroup per column
// with the GEMV prologue's own functions (same virtual waves, same bytes), in column groups that each fit LDS (Launch::run's halving); the GEMV then copies its
// group's image (GemvArgs::x_img, like o_proj after mrs_dec_attention).
struct ActImgArgs {
  const float *x; int ldx; const float *norm_w; float eps; int K, mode;
  char *img;
  int gc0[8], gn[8];  // per column: first column and size of its group (the group's image starts at act_bytes(K, gc0))
};
__global__ void __launch_bounds__(NT) dec_act_image_kernel(const ActImgArgs a) {
  __shared__ float red[8];
  const int tid = tid_opaque(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = (int)blockIdx.x;
  const float *x = a.x + (size_t)c * a.ldx;
  const ActRegs<1> pre = act_issue_all<1>(x, (unsigned)a.K * 4u, a.norm_w, a.K, wave);
  act_sumsq_all<1, 1>(red, pre, x, a.ldx, a.norm_w, a.K, wave);
  if (a.norm_w) __syncthreads();
  const int c0 = a.gc0[c], n = a.gn[c];
  act_quantize_all<1, 1>(a.img + act_bytes(a.K, c0), red, pre, x, a.ldx, a.norm_w, a.eps, a.K, a.mode, wave, c - c0, n);
}
// the column groups of a batch of b columns of K values: halved until a group's image fits LDS (the same recursion as Launch::run)
constexpr size_t LDS_IMG_MAX = (size_t)158 * 1024;
static void col_groups(int K, int c0, int b, int *gc0, int *gn) {
  if (b > 1 && act_bytes(K, b) > LDS_IMG_MAX) { col_groups(K, c0, b / 2, gc0, gn); col_groups(K, c0 + b / 2, b - b / 2, gc0, gn); return; }
  for (int c = c0; c < c0 + b; ++c) { gc0[c] = c0; gn[c] = b; }
}

// ------------------------------------------------------------------------------------------------ launch
static unsigned long long *g_tl_buf = nullptr;
template <int EPI> struct Launch {
  template <int NCOLS> static int go(GemvArgs a, hipStream_t s) {
    const Geo g = geo_for(a.K);
    constexpr int NCI = EPI == EPI_RESID2 ? 2 : NCOLS;
    a.tl = g_tl_buf;
    a.rgpu = EPI == EPI_QKV && g.R < 2 ? 2 : 1;
    const int upr = g.R * a.rgpu;  // rows per unit
    int grid = 0;
    if (EPI == EPI_QKV) {
      // workgroups per tensor in proportion to its bytes (a workgroup streams one tensor), at least one each
      double bytes[3]; double tot = 0; int want = 0;
      for (int i = 0; i < 3; ++i) {
        if (a.nrows[i] % upr) return -3;
        a.units[i] = a.nrows[i] / upr;
        bytes[i] = (double)a.units[i] * a.rgpu * g.TPC * (double)rec_bytes(a.m[i].type, g); tot += bytes[i]; want += a.units[i];
      }
      const int G = want < 256 ? want : 256;
      int wgs[3], used = 0;
      for (int i = 0; i < 3; ++i) { wgs[i] = std::max(1, std::min(a.units[i], (int)(G * bytes[i] / tot + 0.5))); used += wgs[i]; }
      while (used > G) { int big = 0; for (int i = 1; i < 3; ++i) if (wgs[i] > wgs[big]) big = i; if (wgs[big] <= 1) break; --wgs[big]; --used; }
      a.wg0[0] = 0; for (int i = 0; i < 3; ++i) a.wg0[i + 1] = a.wg0[i] + wgs[i];
      grid = a.wg0[3];
    } else {
      const int slots = a.slots > 1 ? a.slots : 1;
      if ((slots > 1 || a.expert_sel) && a.nrows[0] % upr) return -3;  // a unit must not straddle two experts
      const int upe = (a.nrows[0] + upr - 1) / upr;
      a.units[0] = upe * slots;
      grid = a.units[0] < 256 ? a.units[0] : 256;
    }
    if (grid < 1) return -1;
    // the units of a tensor over its workgroups, and the units per expert slot: divisions done here, not at the head of every workgroup
    for (int i = 0; i < (EPI == EPI_QKV ? 3 : 1); ++i) {
      const int nwg = EPI == EPI_QKV ? a.wg0[i + 1] - a.wg0[i] : grid;
      a.ubase[i] = a.units[i] / nwg; a.urem[i] = a.units[i] % nwg;
    }
    a.upe = EPI == EPI_RESID2 ? a.units[0] : a.units[0] / (a.slots > 1 ? a.slots : 1);
    if (a.upe < 1) a.upe = 1;
    size_t lds = (act_bytes(a.K, NCI) + 15) & ~(size_t)15;
    if (lds > LDS_IMG_MAX) return -2;
    int tmask = 0;
    for (int i = 0; i < (EPI == EPI_QKV ? 3 : 1); ++i) tmask |= tmask_of(a.m[i].type);
    // ring depth: 2 tiles per wave when every wave of a workgroup streams (>= 8 units per workgroup), the format's 4 otherwise (dec_core2.cuh stream() RING2)
    int total_units = 0;
    for (int i = 0; i < (EPI == EPI_QKV ? 3 : 1); ++i) total_units += a.units[i];
    static const int force_ring2 = [] { const char *e = getenv("MRS_DEC_RING2"); return e ? atoi(e) : -1; }();
    const bool ring2 = force_ring2 >= 0 ? force_ring2 != 0 : total_units >= 8 * grid;
    return gemv_launch<NCOLS>(EPI, tmask, ring2, grid, lds, a, s);
  }
  // activation columns [c0, ...) of a batched launch: every per-column pointer moves
  static GemvArgs shift_cols(GemvArgs a, int c0) {
    if (a.x) a.x += (size_t)c0 * a.ldx;
    if (a.out) a.out += (size_t)c0 * a.out_stride;
    if (a.q_out) a.q_out += (size_t)c0 * a.nrows[0];
    if (a.positions) a.positions += c0;
    if (a.slot_mapping) a.slot_mapping += c0;
    if (a.x_img) a.x_img = (const char *)a.x_img + act_bytes(a.K, c0);  // (mrs_dec_act_image writes one image per column group: col_groups)
    return a;
  }
  static int run(const GemvArgs &a, int b, hipStream_t s) {
    // the activation image of all columns must fit LDS (1.39 K bytes per column): wider batches run as column groups, each a launch of its own
    // (Llama-3-70B down_proj, K = 28672: 4 columns per launch)
    if (b > 1 && act_bytes(a.K, b) > LDS_IMG_MAX) {
      const int half = b / 2;
      const int rc = run(a, half, s);
      return rc ? rc : run(shift_cols(a, half), b - half, s);
    }
    switch (b) {
    case 1: return go<1>(a, s); case 2: return go<2>(a, s); case 3: return go<3>(a, s); case 4: return go<4>(a, s);
    case 5: return go<5>(a, s); case 6: return go<6>(a, s); case 7: return go<7>(a, s); case 8: return go<8>(a, s);
    default: return -1;
    }
  }
};

}  // namespace dec
}  // namespace mrs

using namespace mrs;
using namespace mrs::dec;

struct mrs_dec_mat_c { const void *planes; int type; long long n, k; };  // == mrs_dec_mat (include/mrs_hip_ext.h)

extern "C" void mrs_dec_timeline(void *buf, int) { g_tl_buf = (unsigned long long *)buf; }
extern "C" int mrs_dec_supported(int ggml_type) { return dec_type(ggml_type) ? 1 : 0; }
extern "C" size_t mrs_dec_repack_bytes(int type, long long n, long long k) {
  if (!dec_type(type) || k <= 0 || k % 256 || n <= 0) return 0;
  return tensor_bytes(type, n, k);
}
extern "C" int mrs_dec_repack(const void *gguf_blocks, int type, long long n, long long k, void *planes, void *stream) {
  if (!mrs_dec_repack_bytes(type, n, k) || !gguf_blocks || !planes) return -1;
  const Geo g = geo_for((int)k);
  const long long nslots = ((n + g.R - 1) / g.R) * g.TPC * 64;  // one thread per lane of every tile
  const dim3 grid((unsigned)((nslots + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
  case T_Q4_K: hipLaunchKernelGGL(repack_kernel<T_Q4_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  case T_Q5_K: hipLaunchKernelGGL(repack_kernel<T_Q5_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  case T_Q6_K: hipLaunchKernelGGL(repack_kernel<T_Q6_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  default: hipLaunchKernelGGL(repack_kernel<T_Q8_0>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, n, (int)k, nslots); break;
  }
  return 0;
}
// q, k, v projections of the decode step: h [b][ldh] f32 -> RMSNorm -> quantize -> GEMV -> RoPE (interleaved pairs) -> q_out f32 [b][nq],
// k / v into the paged cache (kv_dtype 1 = bf16, 0 = f16).  All three tensors must share the activation format (K-quants or Q8_0).
static int dec_qkv_impl(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                        float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                        const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, int neox, void *stream,
                        const void *x_img = nullptr) {
  GemvArgs a{};
  a.neox = neox;
  a.x_img = x_img;
  if (neox && rot_pairs * 2 != head_dim) return -1;  // pair order (i, i + head_dim / 2) is the rotate-half pairing only when every dim rotates
  if (!wq || !wk || !wv || !make_mat(a.m[0], wq->planes, wq->type, wq->n, wq->k) || !make_mat(a.m[1], wk->planes, wk->type, wk->n, wk->k) ||
      !make_mat(a.m[2], wv->planes, wv->type, wv->n, wv->k)) return -1;
  if (wq->k != wk->k || wq->k != wv->k || ((wq->n | wk->n | wv->n | head_dim) & 1)) return -1;
  if (act_mode_for(wq->type) != act_mode_for(wk->type) || act_mode_for(wq->type) != act_mode_for(wv->type)) return -1;
  if (kv_dtype != 0 && kv_dtype != 1) return -1;
  a.nrows[0] = (int)wq->n; a.nrows[1] = (int)wk->n; a.nrows[2] = (int)wv->n; a.K = (int)wq->k;
  a.x = h; a.ldx = ldh; a.norm_w = norm_w; a.eps = eps; a.q_out = q_out; a.k_cache = k_cache; a.v_cache = v_cache; a.slot_mapping = slot_mapping;
  a.positions = positions; a.cos_t = cos_t; a.sin_t = sin_t; a.head_dim = head_dim; a.rot_pairs = rot_pairs; a.num_kv_heads = num_kv_heads;
  a.block_size = block_size; a.cache_x = 8; a.kv_f16 = kv_dtype == 0;
  auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  a.hd_shift = lg2(head_dim); a.bs_shift = lg2(block_size); a.x_shift = lg2(a.cache_x);
  if (a.hd_shift < 0 || a.bs_shift < 0 || head_dim < a.cache_x) return -1;  // powers of two (every head size the engine's attention takes, every block size of the reference's cache)
  return Launch<EPI_QKV>::run(a, b, (hipStream_t)stream);
}
extern "C" int mrs_dec_qkv(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                           float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                           const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream) {
  return dec_qkv_impl(wq, wk, wv, h, ldh, norm_w, eps, q_out, k_cache, v_cache, slot_mapping, positions, cos_t, sin_t, head_dim, rot_pairs, num_kv_heads, block_size,
                      kv_dtype, b, 0, stream);
}
// Rotate-half ("neox") RoPE, as safetensors Llama / Mistral checkpoints use it (RotaryEmbedding::forward with is_gpt_neox, layers.rs:2978): the planes of
// wq and wk must have been repacked from rows in PAIR order -- inside every head the original rows 0, hd/2, 1, hd/2 + 1, ... -- so that a wave still holds
// both members of a pair; the results are written back to dims i and i + hd/2 (q_out and the K pages keep the model's dim order).  wv: original order.
extern "C" int mrs_dec_qkv_neox(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                                float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                                const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream) {
  return dec_qkv_impl(wq, wk, wv, h, ldh, norm_w, eps, q_out, k_cache, v_cache, slot_mapping, positions, cos_t, sin_t, head_dim, rot_pairs, num_kv_heads, block_size,
                      kv_dtype, b, 1, stream);
}

// gate / up: h -> RMSNorm -> quantize -> act(W_g . x) * (W_u . x) -> act_out f32 [b][ld_out].  expert_sel != nullptr: stacked experts
// [E * n][K], the expert index is read on the device (graph-capturable routing).
extern "C" int mrs_dec_gate_up(const mrs_dec_mat_c *wg, const mrs_dec_mat_c *wu, int n, const int32_t *expert_sel, const float *h, int ldh, const float *norm_w,
                               float eps, int activation, float *act_out, int ld_out, int b, void *stream) {
  GemvArgs a{};
  if (!wg || !wu || !make_mat(a.m[0], wg->planes, wg->type, wg->n, wg->k) || !make_mat(a.m[1], wu->planes, wu->type, wu->n, wu->k)) return -1;
  if (wg->type != wu->type || wg->n != wu->n || wg->k != wu->k || n <= 0 || wg->n % n || (!expert_sel && wg->n != n) || wg->n / n > 256) return -1;
  a.nrows[0] = a.nrows[1] = n; a.K = (int)wg->k; a.x = h; a.ldx = ldh; a.norm_w = norm_w; a.eps = eps; a.activation = activation;
  a.out = act_out; a.out_stride = ld_out; a.expert_sel = expert_sel;
  return Launch<EPI_GLU>::run(a, b, (hipStream_t)stream);
}

// MoE decode: the gate / up phase of ALL top-k experts of one token in one launch (one activation prologue, one dispatch, 2 x top-k x n rows streamed):
// expert_sel [topk] on the device, act_out [topk][ld_out].  -3: the row count does not split into whole waves per expert (caller loops over mrs_dec_gate_up).
extern "C" int mrs_dec_gate_up_topk(const mrs_dec_mat_c *wg, const mrs_dec_mat_c *wu, int n, const int32_t *expert_sel, int topk, const float *h, const float *norm_w,
                                    float eps, int activation, float *act_out, int ld_out, void *stream) {
  GemvArgs a{};
  if (!wg || !wu || !expert_sel || topk < 1 || topk > 8 || !make_mat(a.m[0], wg->planes, wg->type, wg->n, wg->k) || !make_mat(a.m[1], wu->planes, wu->type, wu->n, wu->k)) return -1;
  if (wg->type != wu->type || wg->n != wu->n || wg->k != wu->k || n <= 0 || wg->n % n || wg->n / n > 256) return -1;
  a.nrows[0] = a.nrows[1] = n; a.K = (int)wg->k; a.x = h; a.ldx = (int)wg->k; a.norm_w = norm_w; a.eps = eps; a.activation = activation;
  a.out = act_out; a.out_stride = ld_out; a.expert_sel = expert_sel; a.slots = topk; a.slot_out_stride = ld_out;
  return Launch<EPI_GLU>::run(a, 1, (hipStream_t)stream);
}

// MoE decode, top-2: out = (out * resid_scale + w[0] W_{sel[0]} . x[0]) + w[1] W_{sel[1]} . x[1] in one launch; x [2][ldx] = the two experts' activations,
// expert_sel / acc_scale [2] on the device.  Same bits as two mrs_dec_proj launches (mode 1: resid_scale, then 1).
extern "C" int mrs_dec_proj_top2(const mrs_dec_mat_c *w, int n, const int32_t *expert_sel, const float *x, int ldx, float *out, float resid_scale,
                                 const float *acc_scale, void *stream) {
  GemvArgs a{};
  if (!w || !expert_sel || !acc_scale || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n % n || w->n / n > 256) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x = x; a.ldx = ldx; a.out = out; a.out_stride = n; a.resid_scale = resid_scale; a.acc_scale = acc_scale;
  a.expert_sel = expert_sel;
  return Launch<EPI_RESID2>::go<1>(a, (hipStream_t)stream);
}

// plain projection: x [b][ldx] f32 (-> RMSNorm when norm_w) -> quantize -> GEMV.  mode 0: out = W.x;  mode 1: out = out * resid_scale + s * W.x
// (s = *acc_scale or 1; resid_scale = 1 / world_size under tensor parallelism, distributed/layers.rs:965-975)
extern "C" int mrs_dec_proj(const mrs_dec_mat_c *w, int n, const int32_t *expert_sel, const float *x, int ldx, const float *norm_w, float eps, float *out,
                            int ld_out, int mode, float resid_scale, const float *acc_scale, int b, void *stream) {
  GemvArgs a{};
  if (!w || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n % n || (!expert_sel && w->n != n) || w->n / n > 256) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.out = out; a.out_stride = ld_out;
  a.resid_scale = resid_scale; a.acc_scale = acc_scale; a.expert_sel = expert_sel;
  return mode ? Launch<EPI_RESID>::run(a, b, (hipStream_t)stream) : Launch<EPI_STORE>::run(a, b, (hipStream_t)stream);
}

// lm_head of a greedy decode step (round 6): out = W . RmsNorm(x) for ONE column, and the arg-max of the launch folded into its epilogue: every workgroup's largest
// output as a packed (value, index) key -> atomicMax(amax) (amax: one u64, zero before the launch; mrs_sample_advance_embed consumes and re-zeroes it).  The role of
// sample_cuda_top1_row (mistralrs-core/src/ops.rs:2206): one pass over the logits, here while they are still in registers.
extern "C" int mrs_dec_proj_argmax(const mrs_dec_mat_c *w, int n, const float *x, int ldx, const float *norm_w, float eps, float *out, int ld_out, void *amax, void *stream) {
  GemvArgs a{};
  if (!w || !amax || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n != n) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.out = out; a.out_stride = ld_out; a.resid_scale = 1.0f;
  a.amax = (unsigned long long *)amax;
  return Launch<EPI_STORE>::run(a, 1, (hipStream_t)stream);
}

// o_proj & friends on activations the producer already quantized: x_img = the LDS image of b columns of k values, in the quantization w's type takes (Q8_K for the
// K-quants: what mrs_dec_attention writes; Q8_0 for Q8_0 weights: mrs_dec_act_image with that weight type -- the caller pairs them, the image carries no tag)
extern "C" int mrs_dec_proj_img(const mrs_dec_mat_c *w, int n, const void *x_img, float *out, int ld_out, int mode, float resid_scale, int b, void *stream) {
  GemvArgs a{};
  if (!w || !x_img || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n != n) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x_img = x_img; a.out = out; a.out_stride = ld_out; a.resid_scale = resid_scale;
  return mode ? Launch<EPI_RESID>::run(a, b, (hipStream_t)stream) : Launch<EPI_STORE>::run(a, b, (hipStream_t)stream);
}
extern "C" size_t mrs_dec_act_image_bytes(int k, int b) { return act_bytes(k, b); }
extern "C" size_t mrs_dec_proj_img_max_bytes(void) { return LDS_IMG_MAX; }

// The activation image of a batch, built once: x [b][ldx] f32 (-> RmsNorm when norm_w) -> the images of the column groups, act_bytes(k, b) bytes in all, for the
// *_img entry points of a weight of type `weight_type` (K-quants: Q8_K quantization, Q8_0: Q8_0).  One workgroup per column; b <= 8.
extern "C" int mrs_dec_act_image(const float *x, int ldx, const float *norm_w, float eps, int k, int weight_type, int b, void *img_out, void *stream) {
  if (!x || !img_out || b < 1 || b > 8 || k <= 0 || k % 256 || !dec_type(weight_type) || ((uintptr_t)img_out & 15)) return -1;
  ActImgArgs a{};
  a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.K = k; a.mode = act_mode_for(weight_type); a.img = (char *)img_out;
  col_groups(k, 0, b, a.gc0, a.gn);
  hipLaunchKernelGGL(dec_act_image_kernel, dim3(b), dim3(NT), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
// mrs_dec_qkv / mrs_dec_qkv_neox (neox != 0) on an image of RmsNorm(h) made by mrs_dec_act_image for these weights
extern "C" int mrs_dec_qkv_img(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const void *x_img, float *q_out, void *k_cache, void *v_cache,
                               const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t, int head_dim, int rot_pairs,
                               int num_kv_heads, int block_size, int kv_dtype, int b, int neox, void *stream) {
  if (!x_img) return -1;
  return dec_qkv_impl(wq, wk, wv, nullptr, 0, nullptr, 0.f, q_out, k_cache, v_cache, slot_mapping, positions, cos_t, sin_t, head_dim, rot_pairs, num_kv_heads, block_size,
                      kv_dtype, b, neox, stream, x_img);
}
// mrs_dec_gate_up (dense) on an image of RmsNorm(h)
extern "C" int mrs_dec_gate_up_img(const mrs_dec_mat_c *wg, const mrs_dec_mat_c *wu, int n, const void *x_img, int activation, float *act_out, int ld_out, int b,
                                   void *stream) {
  GemvArgs a{};
  if (!x_img || !wg || !wu || !make_mat(a.m[0], wg->planes, wg->type, wg->n, wg->k) || !make_mat(a.m[1], wu->planes, wu->type, wu->n, wu->k)) return -1;
  if (wg->type != wu->type || wg->n != wu->n || wg->k != wu->k || n <= 0 || wg->n != n) return -1;
  a.nrows[0] = a.nrows[1] = n; a.K = (int)wg->k; a.x_img = x_img; a.activation = activation; a.out = act_out; a.out_stride = ld_out;
  return Launch<EPI_GLU>::run(a, b, (hipStream_t)stream);
}

// Decode attention of the engine, one launch (dec_attn2_kernel).  out_f32 [b][num_heads * 128] (may be NULL when an image is written);
// img_out: Q8_K activation image for mrs_dec_proj_img (written when the GQA group is even; may be NULL); ticket: [b * num_kv_heads] u32, zero
// before the first call (the kernel leaves it zero); part_*: the partials workspace of mrs_decode_attention_f32_*.  Returns 1 when the image
// was written, 0 when only out_f32 was, < 0 on a shape outside the kernel (head size 128, 32-token pages, GQA group 1 / 2 / 4 / 8).
extern "C" int mrs_dec_attention(float *out_f32, void *img_out, unsigned *ticket, float *part_o, float *part_m, float *part_l, const float *q, const void *k_cache,
                                 const void *v_cache, int num_kv_heads, float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size,
                                 int max_context_len, int num_seqs, int num_heads, int head_size, int max_blocks_per_seq, int q_stride, int kv_block_stride,
                                 int kv_head_stride, int kv_dtype, int sliding_window, void *stream) {
  if (!ticket || !part_o || !part_m || !part_l || block_size != 32 || head_size != 128 || num_seqs <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads ||
      (kv_dtype != 0 && kv_dtype != 1) || max_context_len <= 0) return -1;
  const int G = num_heads / num_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8) return -1;
  const bool with_img = img_out && (G % 2 == 0) && num_seqs <= 8;
  if (!with_img && !out_f32) return -1;
  Attn2Args a{};
  AttnArgs &t = a.t;
  t.q = q; t.k_cache = (const uint16_t *)k_cache; t.v_cache = (const uint16_t *)v_cache; t.block_tables = block_tables; t.context_lens = context_lens;
  t.part_o = part_o; t.part_m = part_m; t.part_l = part_l; t.out = out_f32;
  t.num_heads = num_heads; t.num_kv_heads = num_kv_heads; t.max_blocks_per_seq = max_blocks_per_seq; t.q_stride = q_stride;
  t.kv_block_stride = kv_block_stride; t.kv_head_stride = kv_head_stride; t.num_seqs = num_seqs; t.scale = scale;
  t.window = sliding_window > 0 ? sliding_window : 0;
  const int nblk = (max_context_len + 31) / 32;
  t.bpw = nblk <= 64 ? 1 : (nblk + 63) / 64;                        // == dec_bpw() of paged_attention.hip: at most 64 splits
  { static const int force = [] { const char *e = getenv("MRS_DEC_ATTN_BPW"); return e ? atoi(e) : 0; }(); if (force > 0) t.bpw = force; }  // measurements only: the oracle's order follows the rule above
  t.max_splits = mrs_decode_attention_max_splits(max_context_len);  // stride of the partials, as in the two-launch route
  a.ticket = ticket; a.img = with_img ? (uint8_t *)img_out : nullptr;
  const int nsplit = (nblk + t.bpw - 1) / t.bpw;
  const dim3 grid(num_kv_heads, num_seqs, nsplit);
  hipStream_t s = (hipStream_t)stream;
  // MRS_DEC_ATTN_TICKET (default 1): the merge inside the split launch (last arriver); 0 = a second launch for the merge.  Measured on the MI355X
  // (profiles/round3_decode.md): the hand-off (write-through partials, drain, device-scope ticket, sc1 loads) costs about what the second launch and its
  // boundary cost -- 472.8 vs 465.4 tok/s for the whole step
  static const int one_launch = [] { const char *e = getenv("MRS_DEC_ATTN_TICKET"); return e ? atoi(e) : 1; }();
  const dim3 mgrid(num_kv_heads, num_seqs);
#define MRS_A2(GG, CT)                                                                                                   \
  do {                                                                                                                   \
    if (one_launch) hipLaunchKernelGGL((dec_attn2_kernel<GG, CT, true>), grid, dim3(64 * GG), 0, s, a);                      \
    else {                                                                                                               \
      hipLaunchKernelGGL((dec_attn2_kernel<GG, CT, false>), grid, dim3(64 * GG), 0, s, a);                                   \
      hipLaunchKernelGGL((dec_attn2_merge_kernel<GG>), mgrid, dim3(64 * ((GG + 1) / 2)), 0, s, a);                       \
    }                                                                                                                    \
  } while (0)
#define MRS_A2G(CT) switch (G) { case 1: MRS_A2(1, CT); break; case 2: MRS_A2(2, CT); break; case 4: MRS_A2(4, CT); break; default: MRS_A2(8, CT); break; }
  if (kv_dtype == 1) { MRS_A2G(bf16_t) } else { MRS_A2G(f16_t) }
#undef MRS_A2G
#undef MRS_A2
  return with_img ? 1 : 0;
}


// ---- the plain launcher of the GEMV core (tests/test_dec2_core.py, scripts/bench_dec.py): out [b][ld_out] = W . (RmsNorm(x) or x), nothing else
extern "C" size_t mrs_dec2_repack_bytes(int type, long long n, long long k) { return mrs_dec_repack_bytes(type, n, k); }
extern "C" int mrs_dec2_repack(const void *gguf_blocks, int type, long long n, long long k, void *planes, void *stream) { return mrs_dec_repack(gguf_blocks, type, n, k, planes, stream); }
extern "C" void mrs_dec2_timeline(void *buf) { g_tl_buf = (unsigned long long *)buf; }
extern "C" int mrs_dec2_gemv(const mrs_dec_mat_c *w, const float *x, int ldx, const float *norm_w, float eps, float *out, int ld_out, int b, void *stream) {
  if (!w) return -1;
  return mrs_dec_proj(w, (int)w->n, nullptr, x, ldx, norm_w, eps, out, ld_out, 0, 1.0f, nullptr, b, stream);
}
