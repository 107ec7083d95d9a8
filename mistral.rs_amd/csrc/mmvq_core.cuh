// mmvq_core.cuh -- decode GEMV core: GGUF weight rows x int8-quantized activations, wave64.
//
// Work decomposition (MI355X: 256 CUs x 4 SIMD, HBM-bound):
//   * the grid is a fixed number of workgroups (a multiple of the CU count); each workgroup owns a
//     contiguous chunk of output rows and its waves take rows round-robin, so at any instant a
//     workgroup streams one contiguous span of the packed weight tensor;
//   * one wave computes one output row at a time: lane i handles weight slices i, i+64, ... of the
//     row (32 weights = one 16-byte load of quants + header), so a wave's load instruction covers a
//     contiguous >=1 KiB span of HBM; weights go HBM -> VGPR directly (used once, no LDS round trip);
//   * the int8 activations (shared by every row) are staged ONCE per workgroup in LDS as
//     16-byte runs + f32 block scales + f32 offset sums, and read with ds_read_b128;
//   * integer dot products with v_dot4_i32_i8, f32 scale/accumulate, xor-butterfly wave reduction.
//
// Reference semantics: mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:724-996 (mmvq_core_impl,
// fused_glu, fused_qkv).  Same integer arithmetic; the f32 summation order differs (documented
// tolerance in tests/test_mmvq.py).
#pragma once
#include "gguf_blocks.cuh"

namespace mrs {

// LDS view of the staged activations for NCOLS batch columns
struct ActLds {
  const int4 *q;    // [col][K/16] runs of 16 int8
  const float *d8;  // [col][K/32] block scales
  const float *S;   // [col][K/16] offset sums per run (only if HAS_OFFSET)
  int runs;         // K/16
};

__host__ __device__ inline size_t act_lds_bytes(int K, int ncols, bool has_offset) {
  return (size_t)ncols * ((size_t)K + (size_t)(K / 32) * 4 + (has_offset ? (size_t)(K / 16) * 4 : 0));
}

// Stage Q8_1 blocks (36 B: half d, half sum(x), 32 x int8) from global memory into LDS.
// y: [col][stride_col_y] blocks.  Reference layout: mmvq_gguf.cu:141-146 (block_q8_1).
template <int TYPE, int NCOLS>
__device__ __forceinline__ ActLds stage_q8_1(char *smem, const uint8_t *__restrict__ y, int K, int stride_col_y) {
  const int runs = K / 16, nblk = K / 32;
  int4 *q = (int4 *)smem;
  float *d8 = (float *)(smem + (size_t)NCOLS * K);
  float *S = d8 + (size_t)NCOLS * nblk;
  for (int i = threadIdx.x; i < NCOLS * runs; i += blockDim.x) {
    const int col = i / runs, run = i - col * runs;
    const uint8_t *blk = y + ((size_t)col * stride_col_y + (run >> 1)) * 36;
    const int4 u = ld16_a4(blk + 4 + (run & 1) * 16);
    q[i] = u;
    const unsigned ds = *(const unsigned *)blk;
    const float d = half_bits_to_float((uint16_t)(ds & 0xffff));
    if ((run & 1) == 0) d8[col * nblk + (run >> 1)] = d;
    if constexpr (Fmt<TYPE>::HAS_OFFSET) {
      if constexpr (Fmt<TYPE>::SUM_MODE == 0) {
        const int su = dot16(make_int4(0x01010101, 0x01010101, 0x01010101, 0x01010101), u);
        S[i] = d * (float)su;
      } else {
        S[i] = 0.5f * half_bits_to_float((uint16_t)(ds >> 16));
      }
    }
  }
  return ActLds{q, d8, S, runs};
}

template <int TYPE, int NCOLS>
__device__ __forceinline__ void accumulate_slice(const Slice &sl, int ra, int rb, const ActLds &act, float (&acc)[NCOLS]) {
  const int nblk = act.runs >> 1;
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) {
    const int4 ua = act.q[c * act.runs + ra];
    const int4 ub = act.q[c * act.runs + rb];
    const float da = act.d8[c * nblk + (ra >> 1)];
    const float db = act.d8[c * nblk + (rb >> 1)];
    float p = (sl.sa * da) * (float)dot16(sl.qa, ua) + (sl.sb * db) * (float)dot16(sl.qb, ub);
    if constexpr (Fmt<TYPE>::HAS_OFFSET) p -= sl.oa * act.S[c * act.runs + ra] + sl.ob * act.S[c * act.runs + rb];
    acc[c] += p;
  }
}

// one full row (all slices) for NCOLS columns; result valid in every lane
template <int TYPE, int NCOLS>
__device__ __forceinline__ void row_dot(const uint8_t *__restrict__ wrow, int nslices, const ActLds &act, float (&acc)[NCOLS]) {
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) acc[c] = 0.0f;
  const int lane = lane_id();
#pragma unroll 2
  for (int s = lane; s < nslices; s += 64) {
    const Slice sl = load_slice<TYPE>(wrow, s);
    int ra, rb;
    slice_runs<TYPE>(s, ra, rb);
    accumulate_slice<TYPE, NCOLS>(sl, ra, rb, act, acc);
  }
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) acc[c] = wave_sum(acc[c]);
}

// two rows at once (fused gate/up): shares the activation LDS reads
template <int TYPE, int NCOLS>
__device__ __forceinline__ void row_dot2(const uint8_t *__restrict__ w0, const uint8_t *__restrict__ w1, int nslices,
                                         const ActLds &act, float (&a0)[NCOLS], float (&a1)[NCOLS]) {
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) a0[c] = a1[c] = 0.0f;
  const int lane = lane_id();
  for (int s = lane; s < nslices; s += 64) {
    const Slice s0 = load_slice<TYPE>(w0, s);
    const Slice s1 = load_slice<TYPE>(w1, s);
    int ra, rb;
    slice_runs<TYPE>(s, ra, rb);
    accumulate_slice<TYPE, NCOLS>(s0, ra, rb, act, a0);
    accumulate_slice<TYPE, NCOLS>(s1, ra, rb, act, a1);
  }
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) { a0[c] = wave_sum(a0[c]); a1[c] = wave_sum(a1[c]); }
}

}  // namespace mrs
