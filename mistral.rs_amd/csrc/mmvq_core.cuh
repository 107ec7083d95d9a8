// mmvq_core.cuh -- decode GEMV core: GGUF weight rows x int8-quantized activations, wave64.
//
// What bounds this kernel on MI355X (measured, profiles/round1): with one v_dot4 path per 32 weights the
// kernel was VALU-ISSUE bound (each wave64 VALU instruction holds the SIMD for 4 cycles; ~95 instructions per
// 32 weights/lane => 5 us of pure issue for a 33 MB matrix), not HBM bound.  The design therefore minimises
// instructions per weight byte:
//   * hot formats (Q4_K, Q5_K, Q6_K) are consumed in WIDE units of 64 weights per lane (4 lanes per 256-weight
//     superblock): one scale/min decode and one set of activation reads per 64 weights;
//   * one wave computes one output row per step: lane l takes unit j*64 + l of the row, so one load instruction
//     of the wave covers a contiguous >= 2 KiB span of the packed row (weights go HBM -> VGPR, used once);
//   * the int8 activations are staged ONCE per workgroup in LDS (int8 values in a bank-conflict-free swizzle,
//     f32 block scales, f32 / int16 offset sums) and read with ds_read_b128;
//   * loads of the NEXT steps are issued before the current step is decoded (software pipeline, exact vmcnt),
//     and the first loads are issued before the activation prologue;
//   * integer dot products with v_dot4_i32_i8, f32 scale/accumulate, DPP wave reduction (no LDS traffic).
// The remaining formats (Q4_0/1, Q5_0/1, Q8_0, Q2_K, Q3_K) keep the 32-weight "slice" path of gguf_blocks.cuh.
//
// Reference semantics: mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:724-996 (mmvq_core_impl,
// fused_glu, fused_qkv).  Same integer arithmetic; the f32 summation order differs (documented
// tolerance in tests/test_mmvq.py).
#pragma once
#include "gguf_blocks.cuh"

namespace mrs {

// ------------------------------------------------------------------------------------------ activations in LDS
// q:   [col][K/16] 16-byte pieces of int8, piece p stored at swz(p) (see below)
// d8:  [col][K/32] f32 Q8_1 block scales
// S:   [col][K/16] f32 offset sums per 16-run: d8 * sum(u) (K-quants, Q6_K's -32) or 0.5 * half(sum x) (Fmt::SUM_MODE 1)
struct ActLds {
  const int4 *q;
  const float *d8;
  const float *S;
  int runs;  // K/16
};

// Piece swizzle: a lane of the wide path reads 4 consecutive pieces (64 B); with a linear layout lanes l and l+4
// hit the same banks (4-way conflict inside every 16-lane ds_read_b128 group).  XOR-ing the low two bits of the
// piece index with bits [5:4] makes the 16 lanes of a group touch 16 distinct 16-byte slots.
__device__ __forceinline__ int swz(int p) { return p ^ ((p >> 4) & 3); }

template <int TYPE> struct Hot { static constexpr bool value = (TYPE == T_Q4_K || TYPE == T_Q5_K || TYPE == T_Q6_K || TYPE == T_Q8_0); };
// measured (profiles/round1): the 64-weight unit wins for Q6_K only; Q4_K / Q5_K are faster on the 32-weight slice path
template <int TYPE> struct Wide { static constexpr bool value = (TYPE == T_Q6_K); };

// K + K/32*4 + K/16*4 = 1.375 K bytes per column
__host__ __device__ inline size_t act_lds_bytes(int K, int ncols, bool /*has_offset*/ = true) {
  return (size_t)ncols * ((size_t)K + (size_t)(K / 32) * 4 + (size_t)(K / 16) * 4);
}

template <int NCOLS> __device__ __forceinline__ ActLds act_view(char *smem, int K) {
  const int nblk = K / 32;
  float *d8 = (float *)(smem + (size_t)NCOLS * K);
  float *S = d8 + (size_t)NCOLS * nblk;
  return ActLds{(const int4 *)smem, d8, S, K / 16};
}

// Stage Q8_1 blocks (36 B: half d, half sum(x), 32 x int8) from global memory into LDS.
// y: [col][stride_col_y] blocks.  Reference layout: mmvq_gguf.cu:141-146 (block_q8_1).
template <int TYPE, int NCOLS>
__device__ __forceinline__ ActLds stage_q8_1(char *smem, const uint8_t *__restrict__ y, int K, int stride_col_y, const int *col_rows = nullptr) {
  const int runs = K / 16, nblk = K / 32;
  const ActLds v = act_view<NCOLS>(smem, K);
  int4 *q = (int4 *)v.q;
  float *d8 = (float *)v.d8, *S = (float *)v.S;
  for (int i = threadIdx.x; i < NCOLS * runs; i += blockDim.x) {
    const int col = i / runs, run = i - col * runs;
    const uint8_t *blk = y + ((size_t)(col_rows ? col_rows[col] : col) * stride_col_y + (run >> 1)) * 36;  // col_rows: gathered rows (grouped MoE)
    const int4 u = ld16_a4(blk + 4 + (run & 1) * 16);
    q[col * runs + swz(run)] = u;
    const unsigned ds = *(const unsigned *)blk;
    const float d = half_bits_to_float((uint16_t)(ds & 0xffff));
    if ((run & 1) == 0) d8[col * nblk + (run >> 1)] = d;
    if constexpr (Fmt<TYPE>::HAS_OFFSET) {
      if constexpr (Fmt<TYPE>::SUM_MODE == 0) S[i] = d * (float)dot16(make_int4(0x01010101, 0x01010101, 0x01010101, 0x01010101), u);
      else S[i] = 0.5f * half_bits_to_float((uint16_t)(ds >> 16));
    }
  }
  return v;
}

// ------------------------------------------------------------------------------------------ narrow path (32 weights / lane)
template <int TYPE, int NCOLS>
__device__ __forceinline__ void accumulate_slice(const Slice &sl, int ra, int rb, const ActLds &act, float (&acc)[NCOLS], bool live = true) {
  const int nblk = act.runs >> 1;
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    const int4 ua = act.q[c * act.runs + swz(ra)];
    const int4 ub = act.q[c * act.runs + swz(rb)];
    const float da = act.d8[c * nblk + (ra >> 1)];
    const float db = act.d8[c * nblk + (rb >> 1)];
    float p = (sl.sa * da) * (float)dot16(sl.qa, ua) + (sl.sb * db) * (float)dot16(sl.qb, ub);
    if constexpr (Fmt<TYPE>::HAS_OFFSET) p -= sl.oa * act.S[c * act.runs + ra] + sl.ob * act.S[c * act.runs + rb];
    acc[c] += live ? p : 0.0f;  // `live` folds away when the caller passes a constant
  }
}

// ------------------------------------------------------------------------------------------ wide path (64 weights / lane)
// Unit u of a row: superblock u >> 2, quarter c = u & 3.
//   Q4_K / Q5_K: quarter c = qs[c*32 .. c*32+32): low nibbles = sub-block 2c, high nibbles = sub-block 2c+1
//                (kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu:146-198 get_quant; mmvq_gguf.cu:586-617)
//   Q6_K:        quarter c = (h = c >> 1, j = c & 1): ql[h*64 + j*32 .. +32): low nibbles = weights h*128 + j*32 + i,
//                high nibbles = weights h*128 + 64 + j*32 + i; 2 more bits from qh[h*32 + i] >> (2j) and >> (2j + 4)
template <int TYPE> struct RawW;
template <> struct RawW<T_Q4_K> { int4 hdr, q0, q1; };
template <> struct RawW<T_Q5_K> { int4 hdr, q0, q1, h0, h1; };
template <> struct RawW<T_Q6_K> { int4 l0, l1, h0, h1; int2 sc; unsigned d; };

template <int TYPE> __device__ __forceinline__ RawW<TYPE> load_wide(const uint8_t *__restrict__ row, int u) {
  RawW<TYPE> r;
  if constexpr (TYPE == T_Q4_K) {
    const uint8_t *blk = row + (size_t)(u >> 2) * 144;
    r.hdr = ld16nt_a4(blk);
    r.q0 = ld16nt_a4(blk + 16 + (u & 3) * 32);
    r.q1 = ld16nt_a4(blk + 32 + (u & 3) * 32);
  } else if constexpr (TYPE == T_Q5_K) {
    const uint8_t *blk = row + (size_t)(u >> 2) * 176;
    r.hdr = ld16nt_a4(blk);
    r.h0 = ld16nt_a4(blk + 16);
    r.h1 = ld16nt_a4(blk + 32);
    r.q0 = ld16nt_a4(blk + 48 + (u & 3) * 32);
    r.q1 = ld16nt_a4(blk + 64 + (u & 3) * 32);
  } else {
    const uint8_t *blk = row + (size_t)(u >> 2) * 210;
    const int c = u & 3, h = c >> 1;
    r.l0 = ld16_a2(blk + c * 32);
    r.l1 = ld16_a2(blk + c * 32 + 16);
    r.h0 = ld16_a2(blk + 128 + h * 32);
    r.h1 = ld16_a2(blk + 144 + h * 32);
    r.sc = ld8_a2(blk + 192 + h * 8);
    r.d = ld2(blk + 208);
  }
  return r;
}

__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

template <int TYPE, int NCOLS>
__device__ __forceinline__ void accumulate_wide(const RawW<TYPE> &w, int u, const ActLds &act, float (&acc)[NCOLS], bool live = true) {
  const int nblk = act.runs >> 1;
  const int blk = u >> 2, c = u & 3;
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const float d = half_bits_to_float((uint16_t)(w.hdr.x & 0xffff));
    const float dmin = half_bits_to_float((uint16_t)((unsigned)w.hdr.x >> 16));
    // branch-free k4(g) for the sub-block pair (2c, 2c+1), see decode_raw()
    const int sh = 16 * (c & 1);
    const unsigned A = (unsigned)w.hdr.y >> sh, B = (unsigned)w.hdr.z >> sh, C = (unsigned)w.hdr.w >> sh;
    const unsigned scH = (C & 0x0f0fu) | ((A >> 2) & 0x3030u), mH = ((C >> 4) & 0x0f0fu) | ((B >> 2) & 0x3030u);
    const unsigned sc = (c < 2) ? (A & 0x3f3fu) : scH, mm = (c < 2) ? (B & 0x3f3fu) : mH;
    const float sc0 = (float)(sc & 0xff), sc1 = (float)((sc >> 8) & 0xff), m0 = (float)(mm & 0xff), m1 = (float)((mm >> 8) & 0xff);
    int4 a0 = and4(w.q0, 0x0F0F0F0F), a1 = and4(w.q1, 0x0F0F0F0F);
    int4 b0 = and4(shr4(w.q0, 4), 0x0F0F0F0F), b1 = and4(shr4(w.q1, 4), 0x0F0F0F0F);
    if constexpr (TYPE == T_Q5_K) {
      a0 = or4(a0, shl4(and4(shr4(w.h0, 2 * c), 0x01010101), 4));
      a1 = or4(a1, shl4(and4(shr4(w.h1, 2 * c), 0x01010101), 4));
      b0 = or4(b0, shl4(and4(shr4(w.h0, 2 * c + 1), 0x01010101), 4));
      b1 = or4(b1, shl4(and4(shr4(w.h1, 2 * c + 1), 0x01010101), 4));
    }
    const int ab = blk * 8 + 2 * c;  // Q8_1 block of sub-block 2c; 2c+1 is the next one
    const int x = (2 * ab >> 4) & 3;  // 4 consecutive pieces: the swizzle only permutes inside the aligned group of 4
#pragma unroll
    for (int col = 0; col < NCOLS; ++col) {
      const int4 *qq = act.q + col * act.runs + (2 * ab);
      const int4 ua0 = qq[0 ^ x], ua1 = qq[1 ^ x], ub0 = qq[2 ^ x], ub1 = qq[3 ^ x];
      const float2 dd = *(const float2 *)(act.d8 + col * nblk + ab);
      const float4 S4 = *(const float4 *)(act.S + col * act.runs + 2 * ab);
      const float2 SS = make_float2(S4.x + S4.y, S4.z + S4.w);
      const int dotA = dot4(a1.w, ua1.w, dot4(a1.z, ua1.z, dot4(a1.y, ua1.y, dot4(a1.x, ua1.x, dot16(a0, ua0)))));
      const int dotB = dot4(b1.w, ub1.w, dot4(b1.z, ub1.z, dot4(b1.y, ub1.y, dot4(b1.x, ub1.x, dot16(b0, ub0)))));
      float t = (sc0 * dd.x) * (float)dotA;
      t = fmaf(sc1 * dd.y, (float)dotB, t);
      float o = m0 * SS.x;
      o = fmaf(m1, SS.y, o);
      const float p = fmaf(-dmin, o, d * t);
      acc[col] += live ? p : 0.0f;
    }
  } else {  // Q6_K
    const int h = c >> 1, j = c & 1;
    const float d = half_bits_to_float((uint16_t)w.d);
    const unsigned xs = (unsigned)w.sc.x >> (16 * j), ys = (unsigned)w.sc.y >> (16 * j);
    const int sa0 = (int)(int8_t)(xs & 0xff), sa1 = (int)(int8_t)((xs >> 8) & 0xff);
    const int sb0 = (int)(int8_t)(ys & 0xff), sb1 = (int)(int8_t)((ys >> 8) & 0xff);
    const int4 t0 = shr4(w.h0, 2 * j), t1 = shr4(w.h1, 2 * j);
    // A: weights h*128 + j*32 + i  = (ql & 15) | ((qh >> 2j) & 3) << 4 ; B: + 64 = (ql >> 4) | ((qh >> (2j+4)) & 3) << 4
    const int4 a0 = or4(and4(w.l0, 0x0F0F0F0F), and4(shl4(t0, 4), 0x30303030));
    const int4 a1 = or4(and4(w.l1, 0x0F0F0F0F), and4(shl4(t1, 4), 0x30303030));
    const int4 b0 = or4(and4(shr4(w.l0, 4), 0x0F0F0F0F), and4(t0, 0x30303030));
    const int4 b1 = or4(and4(shr4(w.l1, 4), 0x0F0F0F0F), and4(t1, 0x30303030));
    const int ab = blk * 8 + h * 4 + j;  // Q8_1 block of A; B is ab + 2
    const int pa = 2 * ab, pb = pa + 4;
#pragma unroll
    for (int col = 0; col < NCOLS; ++col) {
      const int4 *qq = act.q + col * act.runs;
      const int4 ua0 = qq[swz(pa)], ua1 = qq[swz(pa + 1)], ub0 = qq[swz(pb)], ub1 = qq[swz(pb + 1)];
      const float da = act.d8[col * nblk + ab], db = act.d8[col * nblk + ab + 2];
      const float2 Sa = *(const float2 *)(act.S + col * act.runs + pa), Sb = *(const float2 *)(act.S + col * act.runs + pb);
      // sum (q - 32) u = dot - 32 * sum(u) per 16-weight scale group: dots combine in integers, the -32 term in f32
      const int ia = mad24(sa1, dot16(a1, ua1), __mul24(sa0, dot16(a0, ua0)));
      const int ib = mad24(sb1, dot16(b1, ub1), __mul24(sb0, dot16(b0, ub0)));
      float o = (float)sa0 * Sa.x;
      o = fmaf((float)sa1, Sa.y, o);
      o = fmaf((float)sb0, Sb.x, o);
      o = fmaf((float)sb1, Sb.y, o);
      const float p = d * fmaf(-32.0f, o, fmaf(db, (float)ib, da * (float)ia));
      acc[col] += live ? p : 0.0f;
    }
  }
}

// ------------------------------------------------------------------------------------------ paired rows (Q4_K / Q5_K)
// Two rows A, B are streamed with the coalesced one-16-byte-piece-per-lane loads of the slice path (lanes 2k and 2k+1 hold
// the two pieces of quarter k of a superblock, for BOTH rows).  A DPP lane swap then gives the even lane both pieces of row
// A's quarter and the odd lane both pieces of row B's quarter, so every lane runs the 64-weight "wide" arithmetic (one
// scale/min decode, one set of activation reads per 64 weights) on one row.  Even lanes accumulate row A, odd lanes row B.
template <int TYPE> struct PairQ { static constexpr bool value = (TYPE == T_Q4_K || TYPE == T_Q5_K); };

__device__ __forceinline__ int dpp_swap1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ int4 dpp_swap1(int4 v) { return make_int4(dpp_swap1(v.x), dpp_swap1(v.y), dpp_swap1(v.z), dpp_swap1(v.w)); }
__device__ __forceinline__ int4 sel4(bool c, int4 a, int4 b) { return make_int4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

template <int TYPE> __device__ __forceinline__ RawW<TYPE> make_pair_unit(const Raw<TYPE> &a, const Raw<TYPE> &b, bool odd) {
  RawW<TYPE> w;
  const int4 sa = dpp_swap1(a.qs), sb = dpp_swap1(b.qs);
  w.hdr = sel4(odd, b.hdr, a.hdr);
  w.q0 = sel4(odd, sb, a.qs);   // piece 2k   : even lane's own (row A) / even partner's (row B)
  w.q1 = sel4(odd, b.qs, sa);   // piece 2k+1 : odd partner's (row A) / odd lane's own (row B)
  if constexpr (TYPE == T_Q5_K) {
    const int4 ha = dpp_swap1(a.qh), hb = dpp_swap1(b.qh);
    w.h0 = sel4(odd, hb, a.qh);
    w.h1 = sel4(odd, b.qh, ha);
  }
  return w;
}

// sums over the even / odd lanes of the wave (rows A / B): rotations by 2, 4, 8 inside each row of 16, then the four rows
template <int CTRL> __device__ __forceinline__ float dpp_movf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_sum_parity(float v, float &even, float &odd) {
  v += dpp_movf<0x122>(v);  // row_ror:2
  v += dpp_movf<0x124>(v);  // row_ror:4
  v += dpp_movf<0x128>(v);  // row_ror:8  -> lane l holds the sum over the lanes of its row with l's parity
  const int b = __builtin_bit_cast(int, v);
  auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, l)); };
  even = (rl(0) + rl(16)) + (rl(32) + rl(48));
  odd = (rl(1) + rl(17)) + (rl(33) + rl(49));
}

// ------------------------------------------------------------------------------------------ wave reduction
// on the VALU (DPP), no LDS traffic: quads -> rows of 16 -> the four rows via readlane.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror: every lane of a 16-lane row now holds the row sum
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);  // wave-uniform
}

// ------------------------------------------------------------------------------------------ the streaming core
// One wave walks `nrows` output rows (row ids first_row + i*row_step); a row is ceil(nunits/64) steps of 64 units
// (unit = 64 weights for the wide formats, 32 for the others).  The global loads of the NEXT U steps are issued
// (registers only) before the current U steps are decoded and accumulated, and the very first loads are issued before
// the activation prologue runs, so HBM latency overlaps the prologue, the dot products and the wave reductions.
// DUAL streams two rows per step (gate+up, or a RoPE pair) against the same activation reads.
//   rowptr(r, pA, pB): base pointers of row r (pB only if DUAL)
//   pro(): activation prologue, returns the ActLds view (contains the workgroup barriers)
//   epi(r, acc): called once per finished row, acc[m][c] wave-uniform
template <int TYPE> struct UnitT { using type = Raw<TYPE>; };
template <> struct UnitT<T_Q6_K> { using type = RawW<T_Q6_K>; };
template <int TYPE> __host__ __device__ constexpr int unit_weights() { return Wide<TYPE>::value ? 64 : 32; }

template <int TYPE, int NCOLS, bool DUAL, int U, bool FULL, class RowPtr, class Pro, class Epi>
__device__ __forceinline__ void stream_rows(int first_row, int nrows, int row_step, int nunits, RowPtr rowptr, Pro pro, Epi epi) {
  constexpr int NM = DUAL ? 2 : 1;
  using Unit = typename UnitT<TYPE>::type;
  const int lane = lane_id();
  const int ipr = (nunits + 63) >> 6;
  Unit buf[2][U][NM];
  int li = 0, lj = 0;  // loader position (row index, step inside the row)
  // Branch-free issue: positions past the end are clamped to the last valid row / unit (a re-read of bytes the
  // wave already has in flight).  A load inside ANY conditional block (divergent or uniform) makes hipcc lose its
  // exact vmcnt bookkeeping and wait for everything outstanding, which would serialise the pipeline; callers pick
  // U so that 2*U does not exceed the steps a wave actually has (pick_unroll()).
  // a wave without rows (the tail of the last workgroup) must not touch memory at all: its first_row lies past the tensor -- up to rows_per_wave rows,
  // i.e. megabytes for lm_head-sized launches (found on the MI355X as a page fault with a [128256, 8192] Q6_K head whose allocation had 48 KiB of slack;
  // smaller tensors sit inside larger allocator blocks and the stray read went unnoticed).  It still runs the prologue: that holds the workgroup barriers.
  if (nrows <= 0) { (void)pro(); return; }
  const int total = nrows * ipr;
  const int last_row = nrows - 1;
  auto issue = [&](Unit(&b)[U][NM]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = FULL ? lj * 64 + lane : min(lj * 64 + lane, nunits - 1);  // FULL: nunits % 64 == 0
      const uint8_t *pA, *pB;
      rowptr(first_row + min(li, last_row) * row_step, pA, pB);
      if constexpr (Wide<TYPE>::value) {
        b[u][0] = load_wide<TYPE>(pA, s);
        if constexpr (DUAL) b[u][1] = load_wide<TYPE>(pB, s);
      } else {
        b[u][0] = load_raw<TYPE>(pA, s);
        if constexpr (DUAL) b[u][1] = load_raw<TYPE>(pB, s);
      }
      if (++lj == ipr) { lj = 0; ++li; }
    }
  };
  issue(buf[0]);
  const ActLds act = pro();
  float acc[NM][NCOLS];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) acc[m][c] = 0.0f;
  int ci = 0, cj = 0;  // consumer position
  auto consume = [&](Unit(&b)[U][NM]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ci < nrows) {  // wave-uniform
        const int s = cj * 64 + lane;
        const bool live = FULL ? true : (s < nunits);
        const int sc = FULL ? s : min(s, nunits - 1);
        constexpr bool PAIR = DUAL && PairQ<TYPE>::value;
        if constexpr (PAIR) {
          const RawW<TYPE> w = make_pair_unit<TYPE>(b[u][0], b[u][1], (lane & 1) != 0);
          accumulate_wide<TYPE, NCOLS>(w, sc >> 1, act, acc[0], live);  // acc[0]: this lane's row (A on even lanes, B on odd lanes)
        } else if constexpr (Wide<TYPE>::value) {
#pragma unroll
          for (int m = 0; m < NM; ++m) accumulate_wide<TYPE, NCOLS>(b[u][m], sc, act, acc[m], live);
        } else {
          int ra, rb;
          slice_runs<TYPE>(sc, ra, rb);
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const Slice sl = decode_raw<TYPE>(b[u][m], sc);
            accumulate_slice<TYPE, NCOLS>(sl, ra, rb, act, acc[m], live);
          }
        }
        if (++cj == ipr) {
          if constexpr (PAIR) {
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) { float e, o; wave_sum_parity(acc[0][c], e, o); acc[0][c] = e; acc[1][c] = o; }
          } else {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
              for (int c = 0; c < NCOLS; ++c) acc[m][c] = wave_sum_dpp(acc[m][c]);
          }
          epi(first_row + ci * row_step, acc);
#pragma unroll
          for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[m][c] = 0.0f;
          cj = 0;
          ++ci;
        }
      }
    }
  };
  for (int t = 0; t < total; t += 2 * U) {
    issue(buf[1]);
    consume(buf[0]);
    issue(buf[0]);
    consume(buf[1]);
  }
}

// steps a wave has -> prefetch depth (steps per pipeline stage): never issue much more than the wave needs
__host__ __device__ inline int pick_unroll(int steps, int umax) {
  int u = 1;
  while (u < umax && 2 * (2 * u) <= steps) u *= 2;
  return u;
}

// K -> units per row for TYPE; steps_hint = steps per wave (rows per wave * steps per row), grid-uniform
template <int TYPE, int NCOLS, bool DUAL, class RowPtr, class Pro, class Epi>
__device__ __forceinline__ void stream_rows_auto(int first_row, int nrows, int row_step, int K, int rows_hint, RowPtr rowptr, Pro pro, Epi epi) {
  constexpr int UMAX = Wide<TYPE>::value ? (DUAL ? 1 : 2) : (DUAL ? 2 : 4);
  const int nunits = K / unit_weights<TYPE>();
  const int u = pick_unroll(rows_hint * ((nunits + 63) >> 6), UMAX);  // wave-uniform (same for the whole grid)
  if ((nunits & 63) != 0) {  // ragged rows: lane-predicated path
    stream_rows<TYPE, NCOLS, DUAL, 1, false>(first_row, nrows, row_step, nunits, rowptr, pro, epi);
  } else if (UMAX >= 4 && u >= 4) {
    stream_rows<TYPE, NCOLS, DUAL, (UMAX >= 4 ? 4 : UMAX), true>(first_row, nrows, row_step, nunits, rowptr, pro, epi);
  } else if (UMAX >= 2 && u >= 2) {
    stream_rows<TYPE, NCOLS, DUAL, (UMAX >= 2 ? 2 : UMAX), true>(first_row, nrows, row_step, nunits, rowptr, pro, epi);
  } else {
    stream_rows<TYPE, NCOLS, DUAL, 1, true>(first_row, nrows, row_step, nunits, rowptr, pro, epi);
  }
}

}  // namespace mrs
