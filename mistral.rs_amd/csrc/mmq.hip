// mmq.hip -- the reference's MMQ C ABI (prompt-sized quantized matmul) on gfx950:
//   launch_mmq_quantize_q8_1_{D4,DS4,D2S6}(x, ids, vy, type_x, ne00, s01, s02, s03, ne0, ne1, ne2, ne3, stream)
//   launch_mmq_quantize_glu_q8_1_{D4,DS4,D2S6}[_f32](gate, up, ids, vy, [type_x,] ne00, s01, ne0, ne1, activation, stream)
//   launch_mmq_gguf_<t>(tmp_fixup, x, y, dst, ncols_x, nrows_x, ncols_y, stride_row_x, stride_col_dst, cc, nsm, smpbo, warp_size, type_dst, stream)
//   launch_mmq_gguf_<t>_moe(tmp_fixup, x, y, ids_dst, expert_bounds, dst, ncols_x, nrows_x, ncols_dst, stride_row_x, stride_col_dst,
//                           num_experts, ncols_max, cc, nsm, smpbo, warp_size, stream)
// Reference: mistralrs-quant/src/gguf/ffi.rs:1313-1452 (declarations), src/gguf/fast_mmq.rs:388-447 (caller), kernels/mmq_gguf/mmq_quantize.cu
// (quantizers), kernels/mmq_gguf/mmq_instance_<t>.cu + mmq_gguf.cuh:3960-4010 (launchers), mmq_vecdotq.cuh (arithmetic).
//
// What is kept from the reference: the ABI, the block_q8_1_mmq scratch format (so the caller's workspace sizes and the quantize -> matmul
// hand-off are unchanged), and the arithmetic -- integer dots of the weight ints with the activation ints, weight offsets against the STORED
// partial sums where the layout carries them (DS4: per 32 values; D2S6: per 16 values for the first 96 of 128) and d * SUM(u) elsewhere.
// What is not: the tiling.  The reference's kernel is an int8 tensor-core (mma.sync) tile loop with stream-k fix-up; on MI355X the fast prompt
// path is the fused block-dequant -> bf16 MFMA GEMM (ext_gemm.hip, mrs_gemm_q_*), which needs no activation quantizer at all.
// These launchers exist so that fast_mmq.rs links and runs unmodified: one wave per 1 or 4 weight rows x 8 activation columns, lanes striding over
// 32-weight slices (coalesced 16-byte weight loads, the same per-format decode as the MMVQ kernels: gguf_blocks.cuh load_slice), activations
// read from the block_q8_1_mmq scratch through L2.  tmp_fixup / cc / nsm / smpbo / warp_size are accepted and unused (no stream-k here).
// Scale products (d * sc, dmin * m) stay in f32; the reference's Q2_K tile loader rounds them to half.
#include "common.cuh"
#include "gguf_blocks.cuh"
#include <stdlib.h>

namespace mrs {

enum : int { MMQ_D4 = 0, MMQ_DS4 = 1, MMQ_D2S6 = 2 };  // mmq_q8_1_ds_layout, mmq_gguf.cuh:64-68
// which layout a weight type is paired with (mmq_gguf.cuh:100-135; Rust: fast_mmq.rs ds_layout_for)
template <int TYPE> struct MmqLayout { static constexpr int value = MMQ_D4; };
template <> struct MmqLayout<T_Q4_0> { static constexpr int value = MMQ_DS4; };
template <> struct MmqLayout<T_Q4_1> { static constexpr int value = MMQ_DS4; };
template <> struct MmqLayout<T_Q5_1> { static constexpr int value = MMQ_DS4; };
template <> struct MmqLayout<T_Q4_K> { static constexpr int value = MMQ_DS4; };
template <> struct MmqLayout<T_Q5_K> { static constexpr int value = MMQ_DS4; };
template <> struct MmqLayout<T_Q2_K> { static constexpr int value = MMQ_D2S6; };

constexpr int MMQ_BLOCK_BYTES = 144;  // 16 header bytes + 128 int8 (mmq_gguf.cuh:70-89)

// ------------------------------------------------------------------------------------------------ quantizers
// One thread = 4 consecutive values, 128 threads per workgroup = 512 values of one token (mmq_quantize.cu:104-198).  The exchanges stay inside
// groups of <= 16 lanes, so the wave64 butterflies visit the partners in the reference's order (offsets n/8 .. 1).
template <int LAYOUT>
__device__ __forceinline__ void quantize4_store(float4 xi, uint8_t *__restrict__ y, int64_t ib, int iqs) {
  constexpr int per_scale = LAYOUT == MMQ_D2S6 ? 64 : 32, per_sum = LAYOUT == MMQ_D2S6 ? 16 : 32;
  float amax = fmaxf(fmaxf(fabsf(xi.x), fabsf(xi.y)), fmaxf(fabsf(xi.z), fabsf(xi.w)));
#pragma unroll
  for (int off = per_scale / 8; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  float sum = 0.0f;
  if constexpr (LAYOUT != MMQ_D4) {
    sum = xi.x + xi.y + xi.z + xi.w;
#pragma unroll
    for (int off = per_sum / 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  }
  const bool zero = amax == 0.0f;  // the reference reaches 0 * inf = NaN -> int8 0 and d = 1 / inf = 0 here
  const float d_inv = 127.0f / amax, d = zero ? 0.0f : 1.0f / d_inv;
  uint8_t *blk = y + ib * MMQ_BLOCK_BYTES;
  const uint32_t q = zero ? 0u
                          : ((uint32_t)(uint8_t)(int8_t)roundf(xi.x * d_inv) | ((uint32_t)(uint8_t)(int8_t)roundf(xi.y * d_inv) << 8) |
                             ((uint32_t)(uint8_t)(int8_t)roundf(xi.z * d_inv) << 16) | ((uint32_t)(uint8_t)(int8_t)roundf(xi.w * d_inv) << 24));
  *(uint32_t *)(blk + 16 + iqs) = q;
  if constexpr (LAYOUT == MMQ_D2S6) {
    if (iqs % 16 != 0 || iqs >= 96) return;
    *(uint16_t *)(blk + 4 + 2 * (iqs / 16)) = float_to_half_bits(sum);
    if (iqs % 64 != 0) return;
    *(uint16_t *)(blk + 2 * (iqs / 64)) = float_to_half_bits(d);
  } else {
    if (iqs % 32 != 0) return;
    if constexpr (LAYOUT == MMQ_DS4) {
      *(uint16_t *)(blk + 4 * (iqs / 32)) = float_to_half_bits(d);
      *(uint16_t *)(blk + 4 * (iqs / 32) + 2) = float_to_half_bits(sum);
    } else {
      *(float *)(blk + 4 * (iqs / 32)) = d;
    }
  }
}

template <class T> __device__ __forceinline__ float4 load4(const T *__restrict__ x, int64_t base, int64_t i0, int64_t ne00) {
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (i0 + 0 < ne00) v.x = to_f<T>(x[base + 0]);
  if (i0 + 1 < ne00) v.y = to_f<T>(x[base + 1]);
  if (i0 + 2 < ne00) v.z = to_f<T>(x[base + 2]);
  if (i0 + 3 < ne00) v.w = to_f<T>(x[base + 3]);
  return v;
}

template <class T, int LAYOUT>
__global__ void __launch_bounds__(128) mmq_quantize_kernel(const T *__restrict__ x, const int32_t *__restrict__ ids, uint8_t *__restrict__ y, int64_t ne00,
                                                           int64_t s01, int64_t s02, int64_t s03, int64_t ne0, int ne1, int ne2) {
  const int64_t i0 = ((int64_t)blockDim.x * blockIdx.y + threadIdx.x) * 4;
  if (i0 >= ne0) return;  // ne0 % 128 == 0 at every call site: whole exchange groups leave together
  const int64_t i1 = blockIdx.x, i2 = blockIdx.z % ne2, i3 = blockIdx.z / ne2;
  const int64_t i01 = ids ? ids[i1] : i1;
  const int64_t ib0 = (int64_t)blockIdx.z * ((int64_t)gridDim.x * gridDim.y * blockDim.x / 32);  // first block of the channel
  const int64_t ib = ib0 + (i0 / 128) * ne1 + i1;
  const float4 xi = load4(x, i3 * s03 + i2 * s02 + i01 * s01 + i0, i0, ne00);
  quantize4_store<LAYOUT>(xi, y, ib, (int)(i0 % 128));
}

// activation(gate) * up with the product formed in T (mmq_quantize.cu:229-234: `(input_t)act(gate) * (input_t)up`)
template <class T> __device__ __forceinline__ float glu_product(float gate, float up, int act) {
  return round_to<T>(round_to<T>(glu_act(gate, act)) * up);
}

template <class T, int LAYOUT>
__global__ void __launch_bounds__(128) mmq_quantize_glu_kernel(const T *__restrict__ gate, const T *__restrict__ up, const int32_t *__restrict__ ids,
                                                               uint8_t *__restrict__ y, int64_t ne00, int64_t s01, int64_t ne0, int ne1, int act) {
  const int64_t i0 = ((int64_t)blockDim.x * blockIdx.y + threadIdx.x) * 4;
  if (i0 >= ne0) return;
  const int64_t i1 = blockIdx.x, i01 = ids ? ids[i1] : i1;
  const int64_t ib = (i0 / 128) * ne1 + i1;
  const float4 g = load4(gate, i01 * s01 + i0, i0, ne00), u = load4(up, i01 * s01 + i0, i0, ne00);
  const float4 xi = make_float4(glu_product<T>(g.x, u.x, act), glu_product<T>(g.y, u.y, act), glu_product<T>(g.z, u.z, act), glu_product<T>(g.w, u.w, act));
  quantize4_store<LAYOUT>(xi, y, ib, (int)(i0 % 128));
}

template <int LAYOUT>
static void launch_quantize(const void *x, const int32_t *ids, void *vy, int type_x, int64_t ne00, int64_t s01, int64_t s02, int64_t s03, int64_t ne0,
                            int64_t ne1, int64_t ne2, int64_t ne3, void *stream) {
  if (ne0 <= 0 || ne1 <= 0 || ne2 <= 0 || ne3 <= 0) return;
  const dim3 grid((unsigned)ne1, (unsigned)((ne0 + 511) / 512), (unsigned)(ne2 * ne3));
  hipStream_t s = (hipStream_t)stream;
  switch (type_x) {  // ggml type codes: 0 = f32, 1 = f16, 30 = bf16 (fast_mmq.rs:591-596); anything else: no launch, as the reference
  case 0: hipLaunchKernelGGL((mmq_quantize_kernel<float, LAYOUT>), grid, dim3(128), 0, s, (const float *)x, ids, (uint8_t *)vy, ne00, s01, s02, s03, ne0, (int)ne1, (int)ne2); break;
  case 1: hipLaunchKernelGGL((mmq_quantize_kernel<f16_t, LAYOUT>), grid, dim3(128), 0, s, (const f16_t *)x, ids, (uint8_t *)vy, ne00, s01, s02, s03, ne0, (int)ne1, (int)ne2); break;
  case 30: hipLaunchKernelGGL((mmq_quantize_kernel<bf16_t, LAYOUT>), grid, dim3(128), 0, s, (const bf16_t *)x, ids, (uint8_t *)vy, ne00, s01, s02, s03, ne0, (int)ne1, (int)ne2); break;
  default: break;
  }
}

template <int LAYOUT>
static void launch_quantize_glu(const void *gate, const void *up, const int32_t *ids, void *vy, int type_x, int64_t ne00, int64_t s01, int64_t ne0,
                                int64_t ne1, int act, void *stream) {
  if (ne0 <= 0 || ne1 <= 0) return;
  const dim3 grid((unsigned)ne1, (unsigned)((ne0 + 511) / 512), 1);
  hipStream_t s = (hipStream_t)stream;
  switch (type_x) {
  case 0: hipLaunchKernelGGL((mmq_quantize_glu_kernel<float, LAYOUT>), grid, dim3(128), 0, s, (const float *)gate, (const float *)up, ids, (uint8_t *)vy, ne00, s01, ne0, (int)ne1, act); break;
  case 1: hipLaunchKernelGGL((mmq_quantize_glu_kernel<f16_t, LAYOUT>), grid, dim3(128), 0, s, (const f16_t *)gate, (const f16_t *)up, ids, (uint8_t *)vy, ne00, s01, ne0, (int)ne1, act); break;
  case 30: hipLaunchKernelGGL((mmq_quantize_glu_kernel<bf16_t, LAYOUT>), grid, dim3(128), 0, s, (const bf16_t *)gate, (const bf16_t *)up, ids, (uint8_t *)vy, ne00, s01, ne0, (int)ne1, act); break;
  default: break;
  }
}

// ------------------------------------------------------------------------------------------------ matmul
struct MmqArgs {
  const uint8_t *x;            // weights: [channels][nrows_x] rows of stride_row_x blocks
  const uint8_t *y;            // block_q8_1_mmq [k_padded / 128][ncols_y]
  void *dst;                   // dst[col * nrows_dst + row]
  const int32_t *ids_dst;      // MoE: destination column of y column j (else null)
  const int32_t *expert_bounds; // MoE: y columns [bounds[e], bounds[e + 1]) belong to expert e (else null)
  int64_t ncols_x, nrows_x, ncols_y, stride_row_x, nrows_dst, stride_channel_x;
};

// One 16-value activation run of column `col`: its ints, the scale of its 32- (64-) value group and the partner of the weight offset.
template <int LAYOUT, bool HAS_OFFSET>
__device__ __forceinline__ void act_run(const uint8_t *__restrict__ y, int64_t ncols_y, int64_t col, int run, int4 &u, float &d8, float &so) {
  const int e = run * 16, r = e & 127;
  const uint8_t *blk = y + ((int64_t)(e >> 7) * ncols_y + col) * MMQ_BLOCK_BYTES;
  u = ld16_a16(blk + 16 + r);
  if constexpr (LAYOUT == MMQ_D4) d8 = *(const float *)(blk + 4 * (r >> 5));
  else if constexpr (LAYOUT == MMQ_DS4) d8 = half_bits_to_float(ld2(blk + 4 * (r >> 5)));
  else d8 = half_bits_to_float(ld2(blk + 2 * (r >> 6)));
  so = 0.0f;
  if constexpr (HAS_OFFSET) {
    const int4 ones = make_int4(0x01010101, 0x01010101, 0x01010101, 0x01010101);
    if constexpr (LAYOUT == MMQ_DS4) so = 0.5f * half_bits_to_float(ld2(blk + 4 * (r >> 5) + 2));  // the two runs of a 32-block share its stored sum
    else if constexpr (LAYOUT == MMQ_D2S6) so = r < 96 ? half_bits_to_float(ld2(blk + 4 + 2 * (r >> 4))) : d8 * (float)dot16(ones, u);
    else so = d8 * (float)dot16(ones, u);
  }
}

// R weight rows x NC activation columns per wave: one activation run (ints, scale, stored sum) is loaded once and meets R decoded weight
// slices, one weight slice meets NC columns -- R*NC products per (R weight + NC activation) loads.  The per-(row, column) arithmetic and its
// order (lane partial sums over s = lane, lane + 64, ..., then the wave reduction) do not depend on R, so every R gives bit-identical results.
template <int TYPE, class OUT, int NC, int R>
__global__ void __launch_bounds__(256) mmq_kernel(MmqArgs a) {
  constexpr int LAYOUT = MmqLayout<TYPE>::value;
  constexpr bool OFF = Fmt<TYPE>::HAS_OFFSET;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (row0 >= a.nrows_x) return;  // no workgroup barrier below
  const int64_t col_low = a.expert_bounds ? a.expert_bounds[blockIdx.z] : 0;
  const int64_t col_high = a.expert_bounds ? a.expert_bounds[blockIdx.z + 1] : a.ncols_y;
  const int64_t c0 = col_low + (int64_t)blockIdx.y * NC;
  if (c0 >= col_high) return;
  const uint8_t *wbase = a.x + (int64_t)blockIdx.z * a.stride_channel_x * Fmt<TYPE>::TS;
  const int nslices = (int)(a.ncols_x / 32);
  float acc[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = 0.0f;
  for (int s = lane; s < nslices; s += 64) {
    Slice sl[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row0 + r < a.nrows_x ? row0 + r : a.nrows_x - 1;  // clamped: the surplus rows are computed and dropped
      sl[r] = load_slice<TYPE>(wbase + row * a.stride_row_x * Fmt<TYPE>::TS, s);
    }
    int ra, rb;
    slice_runs<TYPE>(s, ra, rb);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t col = c0 + c < col_high ? c0 + c : col_high - 1;  // clamped likewise
      int4 ua, ub;
      float da, db, soa, sob;
      act_run<LAYOUT, OFF>(a.y, a.ncols_y, col, ra, ua, da, soa);
      act_run<LAYOUT, OFF>(a.y, a.ncols_y, col, rb, ub, db, sob);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float p = (sl[r].sa * da) * (float)dot16(sl[r].qa, ua) + (sl[r].sb * db) * (float)dot16(sl[r].qb, ub);
        if constexpr (OFF) p -= sl[r].oa * soa + sl[r].ob * sob;
        acc[r][c] += p;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float v = wave_sum(acc[r][c]);
      if (lane == 0 && row0 + r < a.nrows_x && c0 + c < col_high) {
        const int64_t dcol = a.ids_dst ? a.ids_dst[c0 + c] : c0 + c;
        ((OUT *)a.dst)[dcol * a.nrows_dst + row0 + r] = from_f<OUT>(v);
      }
    }
}

// ------------------------------------------------------------------------------------------------ matrix-core route (Q4_K / Q5_K, DS4 activations)
// The integer dot of one 32-value sub-block is ONE v_mfma_i32_32x32x32_i8: A = 32 activation columns x 32 ints (from the block_q8_1_mmq scratch,
// staged through LDS once per workgroup), B = 32 ints x 32 weight rows (each lane unpacks the 16 nibbles of ITS row straight from the GGUF block:
// lane l = row l % 32, k half l / 32 -- the same 16 bytes serve sub-blocks 2c (low nibbles) and 2c + 1 (high nibbles)).  The result lane holds
// weight row l % 32 and 16 activation columns, so the per-sub-block fix-up  acc += (d sc_j)(row) d8_j(col) isum - (dmin m_j)(row) s8_j(col)
// has its row factors in two lane registers and its column factors as f32 pairs in LDS (converted from the half2 headers when the tile is staged;
// every lane of a k half reads the same address: broadcast).  Same integers, same stored sums as mmq_kernel; only the f32 summation order differs
// (sub-block order per output instead of lane-strided partial sums).  Arithmetic cost: 16 cvt + 16 mul + 32 fma per MFMA -- the route is VALU-bound
// about 8 : 1, which still is ~5 x the v_dot4 kernel.  Workgroup tile 128 weight rows x 128 columns, wave tile 32 x 128.
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v16i __attribute__((ext_vector_type(16)));
// workgroup tile: NW waves x 32 weight rows by NT x 32 activation columns (4 x 4 = 128 x 128 for launches that fill the chip, 2 x 2 for the others)

template <int TYPE, class OUT, int NW, int NT>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 3) mmq_mfma_kernel(MmqArgs a) {
  constexpr int MM_ROWS = 32 * NW, MM_COLS = 32 * NT, NTHR = 64 * NW;  // 2 workgroups per CU (<= 256 VGPRs: 168 spills ~90 values)
  static_assert(TYPE == T_Q4_K || TYPE == T_Q5_K, "DS4 K-quants with 32-value sub-blocks");
  constexpr int TS = Fmt<TYPE>::TS, QS = TYPE == T_Q4_K ? 16 : 48;  // block bytes, offset of qs[128]
  __shared__ __attribute__((aligned(16))) uint8_t raw[2 * MM_COLS * MMQ_BLOCK_BYTES];  // [k block of 128][column][144 B]
  __shared__ __attribute__((aligned(16))) float hdr[8 * MM_COLS * 2];                  // [sub-block][column]{d8, s8}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
  const int64_t col_low = a.expert_bounds ? a.expert_bounds[blockIdx.z] : 0;
  const int64_t col_high = a.expert_bounds ? a.expert_bounds[blockIdx.z + 1] : a.ncols_y;
  const int64_t c0 = col_low + (int64_t)blockIdx.y * MM_COLS;
  if (c0 >= col_high) return;  // workgroup-uniform
  const int64_t row = (int64_t)blockIdx.x * MM_ROWS + wave * 32 + (lane & 31);
  const int64_t rowc = row < a.nrows_x ? row : a.nrows_x - 1;  // surplus rows are computed on the last row and dropped
  const uint8_t *wrow = a.x + ((int64_t)blockIdx.z * a.stride_channel_x + rowc * a.stride_row_x) * TS;
  const int nsb = (int)(a.ncols_x / 256);
  float acc[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  const mm_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int sb = 0; sb < nsb; ++sb) {
    // this lane's row: header + scales, its 16 bytes of each 64-value chunk (and the fifth bits)
    const uint8_t *b = wrow + (int64_t)sb * TS;
    const int4 h = ld16_a16(b);
    int4 q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = ld16_a16(b + QS + 32 * c + 16 * kh);
    int4 q5 = make_int4(0, 0, 0, 0);
    if constexpr (TYPE == T_Q5_K) q5 = ld16_a16(b + 16 + 16 * kh);
    __syncthreads();  // the previous superblock's tile has been read by every wave
    // stage the activation tile: thread = (k block tid / 128, column tid % 128) copies its 144-byte block (columns past the range re-read the last one)
    for (int p = tid; p < 2 * MM_COLS; p += NTHR) {
      const int kb = p / MM_COLS, cc = p % MM_COLS;
      const int64_t col = c0 + cc < col_high ? c0 + cc : col_high - 1;
      const uint8_t *src = a.y + ((int64_t)(2 * sb + kb) * a.ncols_y + col) * MMQ_BLOCK_BYTES;
      uint8_t *dstb = raw + (kb * MM_COLS + cc) * MMQ_BLOCK_BYTES;
      const int4 v0 = ld16_a16(src);
#pragma unroll
      for (int pc = 1; pc < 9; ++pc) *(int4 *)(dstb + pc * 16) = ld16_a16(src + pc * 16);
      *(int4 *)dstb = v0;
      const int w4[4] = {v0.x, v0.y, v0.z, v0.w};  // the four half2 {d, s} of the block's 32-value groups
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float *o = hdr + ((kb * 4 + g) * MM_COLS + cc) * 2;
        o[0] = half_bits_to_float((uint16_t)((unsigned)w4[g] & 0xffffu));
        o[1] = half_bits_to_float((uint16_t)((unsigned)w4[g] >> 16));
      }
    }
    __syncthreads();
    const float d = half_bits_to_float((uint16_t)((unsigned)h.x & 0xffffu)), dmin = half_bits_to_float((uint16_t)((unsigned)h.x >> 16));
    const unsigned sw[3] = {(unsigned)h.y, (unsigned)h.z, (unsigned)h.w};
    auto sbyte = [&](int i) { return (sw[i >> 2] >> (8 * (i & 3))) & 0xffu; };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned sc, mn;  // get_scale_min_k4
      if (j < 4) { sc = sbyte(j) & 63u; mn = sbyte(j + 4) & 63u; }
      else { sc = (sbyte(j + 4) & 15u) | ((sbyte(j - 4) >> 6) << 4); mn = (sbyte(j + 4) >> 4) | ((sbyte(j) >> 6) << 4); }
      const float dsc = d * (float)sc, ndm = -(dmin * (float)mn);
      __builtin_amdgcn_sched_barrier(0);  // keep the eight sub-blocks apart: hoisting their LDS reads / MFMAs over each other costs 450 VGPRs
      const int c = j >> 1, hi = j & 1;
      const int qq[4] = {q[c].x, q[c].y, q[c].z, q[c].w}, q5w[4] = {q5.x, q5.y, q5.z, q5.w};
      mm_v4i bf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned v = ((unsigned)qq[e] >> (4 * hi)) & 0x0f0f0f0fu;
        if constexpr (TYPE == T_Q5_K) v |= (((unsigned)q5w[e] >> j) & 0x01010101u) << 4;
        bf[e] = (int)v;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t) __builtin_amdgcn_sched_barrier(0);
        const mm_v4i af = *(const mm_v4i *)(raw + ((j >> 2) * MM_COLS + 32 * t + (lane & 31)) * MMQ_BLOCK_BYTES + 16 + 32 * (j & 3) + 16 * kh);
        const mm_v16i is = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, zero, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 h0 = *(const float4 *)(hdr + (j * MM_COLS + 32 * t + 8 * g + 4 * kh) * 2);
          const float4 h1 = *(const float4 *)(hdr + (j * MM_COLS + 32 * t + 8 * g + 4 * kh + 2) * 2);
          acc[t][4 * g + 0] = fmaf(ndm, h0.y, fmaf((float)is[4 * g + 0], dsc * h0.x, acc[t][4 * g + 0]));
          acc[t][4 * g + 1] = fmaf(ndm, h0.w, fmaf((float)is[4 * g + 1], dsc * h0.z, acc[t][4 * g + 1]));
          acc[t][4 * g + 2] = fmaf(ndm, h1.y, fmaf((float)is[4 * g + 2], dsc * h1.x, acc[t][4 * g + 2]));
          acc[t][4 * g + 3] = fmaf(ndm, h1.w, fmaf((float)is[4 * g + 3], dsc * h1.z, acc[t][4 * g + 3]));
        }
      }
    }
  }
  if (row >= a.nrows_x) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t col = c0 + 32 * t + 8 * (i >> 2) + 4 * kh + (i & 3);
      if (col < col_high) {
        const int64_t dcol = a.ids_dst ? a.ids_dst[col] : col;
        ((OUT *)a.dst)[dcol * a.nrows_dst + row] = from_f<OUT>(acc[t][i]);
      }
    }
}

// Q6_K (D4 activations: one f32 scale per 32 values, no sums): the scales change every 16 values, so the unit is v_mfma_i32_32x32x16_i8 -- lane l
// holds weight row l % 32 and the 8 values 8 (l / 32) + j of a 16-run; 6-bit values are re-centred to q - 32 (signed bytes), which removes the offset
// term.  The run of index r = 8 h + 2 q + lh (half, quarter, 16-half) starts at element 128 h + 32 q + 16 lh of the superblock, its low bits are
// nibble (q >> 1) of ql[64 h + 32 (q & 1) + 16 lh ..], its high bits the 2-bit field q of qh[32 h + 16 lh ..].  Twice the fix-ups per weight of
// the Q4_K route (16 cvt + 8 pk_mul + 8 pk_fma per 16 k): about 0.6 x its rate.
template <int TYPE, class OUT, int NW, int NT>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 3) mmq_mfma_k16_kernel(MmqArgs a) {
  static_assert(TYPE == T_Q6_K || TYPE == T_Q3_K || TYPE == T_Q2_K, "K-quants with 16-value scales: Q6_K / Q3_K (no minimum, D4 activations), Q2_K (minimum, D2S6)");
  constexpr bool Q2 = TYPE == T_Q2_K;
  constexpr int MM_ROWS = 32 * NW, MM_COLS = 32 * NT, NTHR = 64 * NW;
  constexpr int TS = Fmt<TYPE>::TS;
  __shared__ __attribute__((aligned(16))) uint8_t raw[2 * MM_COLS * MMQ_BLOCK_BYTES];
  // D4 (Q6_K, Q3_K): [32-value group 8][column] d8.  D2S6 (Q2_K): [16-value run 16][column]{d8 of the run's 64-value half, the partner of the minimum:
  // the stored sum of the run for the first 96 values of a 128-block, d8 * SUM(u) for the last 32 (mmq_gguf.cuh:70-89)}
  __shared__ __attribute__((aligned(16))) float hdr[(Q2 ? 32 : 8) * MM_COLS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
  const int64_t col_low = a.expert_bounds ? a.expert_bounds[blockIdx.z] : 0;
  const int64_t col_high = a.expert_bounds ? a.expert_bounds[blockIdx.z + 1] : a.ncols_y;
  const int64_t c0 = col_low + (int64_t)blockIdx.y * MM_COLS;
  if (c0 >= col_high) return;
  const int64_t row = (int64_t)blockIdx.x * MM_ROWS + wave * 32 + (lane & 31);
  const int64_t rowc = row < a.nrows_x ? row : a.nrows_x - 1;
  const uint8_t *wrow = a.x + ((int64_t)blockIdx.z * a.stride_channel_x + rowc * a.stride_row_x) * TS;
  const int nsb = (int)(a.ncols_x / 256);
  float acc[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  const mm_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int sb = 0; sb < nsb; ++sb) {
    const uint8_t *b = wrow + (int64_t)sb * TS;
    // this lane's 8 bytes of every piece (2-byte aligned blocks).  Q6_K: ql8 [h][q & 1][lh] low nibbles, qh8 [h][lh] 2-bit fields, 16 int8 scales.
    // Q3_K (hmask[32] qs[64] scales[12] d): ql8 [h][0][lh] = qs bytes (2-bit field q), qh8 [0][lh] = hmask bytes (bit 4 h + q), 6-bit scales minus 32.
    int2_a2 ql8[2][2][2], qh8[2][2];
    int sc4[4];
    float d, dmin = 0.0f;
    if constexpr (Q2) {  // scales[16] (scale | min << 4), qs[64], half d, half dmin: 84 B, 4-byte aligned
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int lh = 0; lh < 2; ++lh) { ql8[h][0][lh] = *(const int2_a2 *)(b + 16 + 32 * h + 16 * lh + 8 * kh); ql8[h][1][lh] = ql8[h][0][lh]; qh8[h][lh] = ql8[h][0][lh]; }
      const int4 scw = ld16_a4(b);
      sc4[0] = scw.x; sc4[1] = scw.y; sc4[2] = scw.z; sc4[3] = scw.w;
      const unsigned dd = (unsigned)ld4_a2(b + 80);
      d = half_bits_to_float((uint16_t)(dd & 0xffffu));
      dmin = half_bits_to_float((uint16_t)(dd >> 16));
    } else if constexpr (TYPE == T_Q6_K) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int lh = 0; lh < 2; ++lh) {
          qh8[h][lh] = *(const int2_a2 *)(b + 128 + 32 * h + 16 * lh + 8 * kh);
#pragma unroll
          for (int p = 0; p < 2; ++p) ql8[h][p][lh] = *(const int2_a2 *)(b + 64 * h + 32 * p + 16 * lh + 8 * kh);
        }
      const int4 scw = ld16_a2(b + 192);
      sc4[0] = scw.x; sc4[1] = scw.y; sc4[2] = scw.z; sc4[3] = scw.w;
      d = half_bits_to_float(ld2(b + 208));
    } else {
#pragma unroll
      for (int lh = 0; lh < 2; ++lh) {
        qh8[0][lh] = *(const int2_a2 *)(b + 16 * lh + 8 * kh);
        qh8[1][lh] = qh8[0][lh];
#pragma unroll
        for (int h = 0; h < 2; ++h) { ql8[h][0][lh] = *(const int2_a2 *)(b + 32 + 32 * h + 16 * lh + 8 * kh); ql8[h][1][lh] = ql8[h][0][lh]; }
      }
      const unsigned s0 = (unsigned)ld4_a2(b + 96), s1 = (unsigned)ld4_a2(b + 100), s2 = (unsigned)ld4_a2(b + 104);
      // scale j = (j < 8 ? s[j] & 15 : s[j - 8] >> 4) | ((s[8 + j % 4] >> 2 (j / 4)) & 3) << 4, minus 32: four per dword, as signed bytes
      const unsigned lo[4] = {s0 & 0x0f0f0f0fu, s1 & 0x0f0f0f0fu, (s0 >> 4) & 0x0f0f0f0fu, (s1 >> 4) & 0x0f0f0f0fu};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned v = lo[g] | (((s2 >> (2 * g)) & 0x03030303u) << 4);
        sc4[g] = (int)(((v | 0x80808080u) - 0x20202020u) ^ 0x80808080u);
      }
      d = half_bits_to_float(ld2(b + 108));
    }
    __syncthreads();
    for (int p = tid; p < 2 * MM_COLS; p += NTHR) {
      const int kb = p / MM_COLS, cc = p % MM_COLS;
      const int64_t col = c0 + cc < col_high ? c0 + cc : col_high - 1;
      const uint8_t *src = a.y + ((int64_t)(2 * sb + kb) * a.ncols_y + col) * MMQ_BLOCK_BYTES;
      uint8_t *dstb = raw + (kb * MM_COLS + cc) * MMQ_BLOCK_BYTES;
      const int4 v0 = ld16_a16(src);
#pragma unroll
      for (int pc = 1; pc < 9; ++pc) *(int4 *)(dstb + pc * 16) = ld16_a16(src + pc * 16);
      const int w4[4] = {v0.x, v0.y, v0.z, v0.w};
      if constexpr (Q2) {  // D2S6: two half scales (per 64 values), six half sums (16-runs 0..5); runs 6, 7 sum their ints
        const float dh[2] = {half_bits_to_float((uint16_t)((unsigned)w4[0] & 0xffffu)), half_bits_to_float((uint16_t)((unsigned)w4[0] >> 16))};
        const int4 ones = make_int4(0x01010101, 0x01010101, 0x01010101, 0x01010101);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float so;
          if (r < 6) so = half_bits_to_float((uint16_t)(((unsigned)w4[1 + r / 2] >> (16 * (r & 1))) & 0xffffu));
          else so = dh[1] * (float)dot16(ones, *(const int4 *)(dstb + 16 + 16 * r));
          float *o = hdr + ((kb * 8 + r) * MM_COLS + cc) * 2;
          o[0] = dh[r >> 2];
          o[1] = so;
        }
      } else {  // D4: four f32 scales
#pragma unroll
        for (int g = 0; g < 4; ++g) hdr[(kb * 4 + g) * MM_COLS + cc] = __int_as_float(w4[g]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      __builtin_amdgcn_sched_barrier(0);
      const int h = r >> 3, q = (r >> 1) & 3, lh = r & 1;
      const unsigned scb = ((unsigned)sc4[r >> 2] >> (8 * (r & 3))) & 0xffu;
      const float dsc = Q2 ? d * (float)(scb & 15u) : d * (float)(int)(int8_t)scb;
      const float ndm = Q2 ? -(dmin * (float)(scb >> 4)) : 0.0f;
      const int lw[2] = {ql8[h][q & 1][lh].x, ql8[h][q & 1][lh].y}, hw[2] = {qh8[h][lh].x, qh8[h][lh].y};
      unsigned bw[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if constexpr (Q2) {
          bw[e] = ((unsigned)lw[e] >> (2 * q)) & 0x03030303u;  // 0..3: the minimum is a separate term
        } else if constexpr (TYPE == T_Q6_K) {
          const unsigned v = (((unsigned)lw[e] >> (4 * (q >> 1))) & 0x0f0f0f0fu) | ((((unsigned)hw[e] >> (2 * q)) & 0x03030303u) << 4);
          bw[e] = ((v | 0x80808080u) - 0x20202020u) ^ 0x80808080u;  // per byte q - 32, no borrow across bytes
        } else {
          const unsigned v = (((unsigned)lw[e] >> (2 * q)) & 0x03030303u) | ((((unsigned)hw[e] >> (4 * h + q)) & 0x01010101u) << 2);
          bw[e] = ((v | 0x80808080u) - 0x04040404u) ^ 0x80808080u;  // per byte q - 4
        }
      }
      const long bf = (long)(((unsigned long long)bw[1] << 32) | bw[0]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t) __builtin_amdgcn_sched_barrier(0);
        const long af = *(const long *)(raw + ((r >> 3) * MM_COLS + 32 * t + (lane & 31)) * MMQ_BLOCK_BYTES + 16 + 16 * (r & 7) + 8 * kh);
        const mm_v16i is = __builtin_amdgcn_mfma_i32_32x32x16_i8(af, bf, zero, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if constexpr (Q2) {
            const float4 h0 = *(const float4 *)(hdr + (r * MM_COLS + 32 * t + 8 * g + 4 * kh) * 2);
            const float4 h1 = *(const float4 *)(hdr + (r * MM_COLS + 32 * t + 8 * g + 4 * kh + 2) * 2);
            acc[t][4 * g + 0] = fmaf(ndm, h0.y, fmaf((float)is[4 * g + 0], dsc * h0.x, acc[t][4 * g + 0]));
            acc[t][4 * g + 1] = fmaf(ndm, h0.w, fmaf((float)is[4 * g + 1], dsc * h0.z, acc[t][4 * g + 1]));
            acc[t][4 * g + 2] = fmaf(ndm, h1.y, fmaf((float)is[4 * g + 2], dsc * h1.x, acc[t][4 * g + 2]));
            acc[t][4 * g + 3] = fmaf(ndm, h1.w, fmaf((float)is[4 * g + 3], dsc * h1.z, acc[t][4 * g + 3]));
          } else {
            const float4 h0 = *(const float4 *)(hdr + (r >> 1) * MM_COLS + 32 * t + 8 * g + 4 * kh);
            acc[t][4 * g + 0] = fmaf((float)is[4 * g + 0], dsc * h0.x, acc[t][4 * g + 0]);
            acc[t][4 * g + 1] = fmaf((float)is[4 * g + 1], dsc * h0.y, acc[t][4 * g + 1]);
            acc[t][4 * g + 2] = fmaf((float)is[4 * g + 2], dsc * h0.z, acc[t][4 * g + 2]);
            acc[t][4 * g + 3] = fmaf((float)is[4 * g + 3], dsc * h0.w, acc[t][4 * g + 3]);
          }
        }
      }
    }
  }
  if (row >= a.nrows_x) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t col = c0 + 32 * t + 8 * (i >> 2) + 4 * kh + (i & 3);
      if (col < col_high) {
        const int64_t dcol = a.ids_dst ? a.ids_dst[col] : col;
        ((OUT *)a.dst)[dcol * a.nrows_dst + row] = from_f<OUT>(acc[t][i]);
      }
    }
}

// The 32-value block formats (Q8_0, Q4_0, Q4_1, Q5_0, Q5_1): one MFMA per weight block; the loop runs over 128-value k blocks (one activation
// block per column: 18 KiB of LDS per stage).  w = s q - o per block (gguf_blocks.cuh load_slice): s = d; o = 8 d / -m / 16 d / -m / 0.  The offset
// meets the STORED sum of the 32 activations where the layout has one (DS4: Q4_0, Q4_1, Q5_1) and d8 * SUM(u) elsewhere (D4: Q5_0; Q8_0 has no
// offset) -- the sum of the ints is formed once per (column, block) when the tile is staged.
template <int TYPE, class OUT, int NW, int NT>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 3) mmq_mfma_b32_kernel(MmqArgs a) {
  constexpr int MM_ROWS = 32 * NW, MM_COLS = 32 * NT, NTHR = 64 * NW;
  static_assert(TYPE == T_Q8_0 || TYPE == T_Q4_0 || TYPE == T_Q4_1 || TYPE == T_Q5_0 || TYPE == T_Q5_1, "32-value blocks");
  constexpr int TS = Fmt<TYPE>::TS, LAYOUT = MmqLayout<TYPE>::value;
  constexpr bool OFF = Fmt<TYPE>::HAS_OFFSET;
  __shared__ __attribute__((aligned(16))) uint8_t raw[MM_COLS * MMQ_BLOCK_BYTES];   // [column][144 B]
  __shared__ __attribute__((aligned(16))) float hdr[4 * MM_COLS * 2];               // [32-value group][column]{d8, s8}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
  const int64_t col_low = a.expert_bounds ? a.expert_bounds[blockIdx.z] : 0;
  const int64_t col_high = a.expert_bounds ? a.expert_bounds[blockIdx.z + 1] : a.ncols_y;
  const int64_t c0 = col_low + (int64_t)blockIdx.y * MM_COLS;
  if (c0 >= col_high) return;
  const int64_t row = (int64_t)blockIdx.x * MM_ROWS + wave * 32 + (lane & 31);
  const int64_t rowc = row < a.nrows_x ? row : a.nrows_x - 1;
  const uint8_t *wrow = a.x + ((int64_t)blockIdx.z * a.stride_channel_x + rowc * a.stride_row_x) * TS;
  const int nkb = (int)(a.ncols_x / 128);  // launch_mmq_t sends K % 128 != 0 to the v_dot4 kernel
  float acc[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  const mm_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int kb = 0; kb < nkb; ++kb) {
    // this lane's four weight blocks: scale, offset, its 16 values (k half kh) as signed / unsigned bytes
    mm_v4i bf[4];
    float sc[4], no[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint8_t *blk = wrow + (int64_t)(kb * 4 + j) * TS;
      const float d = half_bits_to_float(ld2(blk));
      sc[j] = d;
      int4 v;
      if constexpr (TYPE == T_Q8_0) {
        v = ld16_a2(blk + 2 + 16 * kh);
        no[j] = 0.0f;
      } else {
        constexpr int QO = TYPE == T_Q4_0 ? 2 : TYPE == T_Q4_1 ? 4 : TYPE == T_Q5_0 ? 6 : 8;
        v = and4(shr4(ld16_a2(blk + QO), 4 * kh), 0x0F0F0F0F);  // values 0..15 are the low nibbles, 16..31 the high nibbles of the same 16 bytes
        if constexpr (TYPE == T_Q5_0 || TYPE == T_Q5_1) {
          const unsigned qh = (unsigned)ld4_a2(blk + (TYPE == T_Q5_0 ? 2 : 4)) >> (16 * kh);
          v.x |= spread4(qh, 4); v.y |= spread4(qh >> 4, 4); v.z |= spread4(qh >> 8, 4); v.w |= spread4(qh >> 12, 4);
        }
        if constexpr (TYPE == T_Q4_0) no[j] = -(8.0f * d);
        else if constexpr (TYPE == T_Q5_0) no[j] = -(16.0f * d);
        else no[j] = half_bits_to_float(ld2(blk + 2));  // w = d q + m: o = -m
      }
      bf[j] = mm_v4i{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
    for (int cc = tid; cc < MM_COLS; cc += NTHR) {
      const int64_t col = c0 + cc < col_high ? c0 + cc : col_high - 1;
      const uint8_t *src = a.y + ((int64_t)kb * a.ncols_y + col) * MMQ_BLOCK_BYTES;
      uint8_t *dstb = raw + cc * MMQ_BLOCK_BYTES;
      const int4 v0 = ld16_a16(src);
      const int w4[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 u0 = ld16_a16(src + 16 + 32 * g), u1 = ld16_a16(src + 32 + 32 * g);
        *(int4 *)(dstb + 16 + 32 * g) = u0;
        *(int4 *)(dstb + 32 + 32 * g) = u1;
        float d8, s8 = 0.0f;
        if constexpr (LAYOUT == MMQ_DS4) {
          d8 = half_bits_to_float((uint16_t)((unsigned)w4[g] & 0xffffu));
          s8 = half_bits_to_float((uint16_t)((unsigned)w4[g] >> 16));
        } else {
          d8 = __int_as_float(w4[g]);
          if constexpr (OFF) {
            const int4 ones = make_int4(0x01010101, 0x01010101, 0x01010101, 0x01010101);
            s8 = d8 * (float)(dot16(ones, u0) + dot16(ones, u1));
          }
        }
        if constexpr (OFF) {
          hdr[(g * MM_COLS + cc) * 2] = d8;
          hdr[(g * MM_COLS + cc) * 2 + 1] = s8;
        } else {
          hdr[g * MM_COLS + cc] = d8;  // no offset term: the scales alone, four columns per 16-byte read
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t) __builtin_amdgcn_sched_barrier(0);
        const mm_v4i af = *(const mm_v4i *)(raw + (32 * t + (lane & 31)) * MMQ_BLOCK_BYTES + 16 + 32 * j + 16 * kh);
        const mm_v16i is = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[j], zero, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if constexpr (OFF) {
            const float4 h0 = *(const float4 *)(hdr + (j * MM_COLS + 32 * t + 8 * g + 4 * kh) * 2);
            const float4 h1 = *(const float4 *)(hdr + (j * MM_COLS + 32 * t + 8 * g + 4 * kh + 2) * 2);
            acc[t][4 * g + 0] = fmaf(no[j], h0.y, fmaf((float)is[4 * g + 0], sc[j] * h0.x, acc[t][4 * g + 0]));
            acc[t][4 * g + 1] = fmaf(no[j], h0.w, fmaf((float)is[4 * g + 1], sc[j] * h0.z, acc[t][4 * g + 1]));
            acc[t][4 * g + 2] = fmaf(no[j], h1.y, fmaf((float)is[4 * g + 2], sc[j] * h1.x, acc[t][4 * g + 2]));
            acc[t][4 * g + 3] = fmaf(no[j], h1.w, fmaf((float)is[4 * g + 3], sc[j] * h1.z, acc[t][4 * g + 3]));
          } else {
            const float4 h0 = *(const float4 *)(hdr + j * MM_COLS + 32 * t + 8 * g + 4 * kh);
            acc[t][4 * g + 0] = fmaf((float)is[4 * g + 0], sc[j] * h0.x, acc[t][4 * g + 0]);
            acc[t][4 * g + 1] = fmaf((float)is[4 * g + 1], sc[j] * h0.y, acc[t][4 * g + 1]);
            acc[t][4 * g + 2] = fmaf((float)is[4 * g + 2], sc[j] * h0.z, acc[t][4 * g + 2]);
            acc[t][4 * g + 3] = fmaf((float)is[4 * g + 3], sc[j] * h0.w, acc[t][4 * g + 3]);
          }
        }
      }
    }
  }
  if (row >= a.nrows_x) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t col = c0 + 32 * t + 8 * (i >> 2) + 4 * kh + (i & 3);
      if (col < col_high) {
        const int64_t dcol = a.ids_dst ? a.ids_dst[col] : col;
        ((OUT *)a.dst)[dcol * a.nrows_dst + row] = from_f<OUT>(acc[t][i]);
      }
    }
}

template <int TYPE> constexpr bool mmq_is_b32() { return TYPE == T_Q8_0 || TYPE == T_Q4_0 || TYPE == T_Q4_1 || TYPE == T_Q5_0 || TYPE == T_Q5_1; }
template <int TYPE> constexpr bool mmq_has_mfma() { return TYPE == T_Q4_K || TYPE == T_Q5_K || TYPE == T_Q6_K || TYPE == T_Q3_K || TYPE == T_Q2_K || mmq_is_b32<TYPE>(); }
// prompt-sized launches of the two DS4 K-quants go to the matrix cores (MRS_MMQ_MFMA=0: keep the v_dot4 kernel, for A/B measurements)
static int g_mmq_small_below = -1;  // < 0: not set yet (MRS_MMQ_SMALL_TILES_BELOW, default 384 = 1.5 workgroups of 128 x 128 per CU)
static int mmq_small_tiles_below() {
  if (g_mmq_small_below < 0) { const char *e = getenv("MRS_MMQ_SMALL_TILES_BELOW"); g_mmq_small_below = e ? atoi(e) : 384; }
  return g_mmq_small_below;
}
static bool mmq_mfma_wanted() {
  static const bool on = [] { const char *e = getenv("MRS_MMQ_MFMA"); return !e || atoi(e) != 0; }();
  return on;
}

template <int TYPE, class OUT> static void launch_mmq_t(const MmqArgs &a, int64_t channels, int64_t ncols_max, void *stream) {
  constexpr int NC = 8;
  if (a.nrows_x <= 0 || ncols_max <= 0 || channels <= 0) return;
  if constexpr (mmq_has_mfma<TYPE>()) {
    if (mmq_mfma_wanted() && ncols_max >= 48 && a.nrows_x >= 32 && a.ncols_x % (mmq_is_b32<TYPE>() ? 128 : 256) == 0) {
      // 128 x 128 tiles when they give every CU its two workgroups, 64 x 64 tiles otherwise (a 4096-row tensor at 512 columns: 128 -> 512 workgroups)
      const int64_t big = ((a.nrows_x + 127) / 128) * ((ncols_max + 127) / 128) * channels;
      if (big >= mmq_small_tiles_below()) {
        const dim3 grid((unsigned)((a.nrows_x + 127) / 128), (unsigned)((ncols_max + 127) / 128), (unsigned)channels);
        if constexpr (TYPE == T_Q6_K || TYPE == T_Q3_K || TYPE == T_Q2_K) hipLaunchKernelGGL((mmq_mfma_k16_kernel<TYPE, OUT, 4, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else if constexpr (mmq_is_b32<TYPE>()) hipLaunchKernelGGL((mmq_mfma_b32_kernel<TYPE, OUT, 4, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((mmq_mfma_kernel<TYPE, OUT, 4, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
      } else {
        const dim3 grid((unsigned)((a.nrows_x + 63) / 64), (unsigned)((ncols_max + 63) / 64), (unsigned)channels);
        if constexpr (TYPE == T_Q6_K || TYPE == T_Q3_K || TYPE == T_Q2_K) hipLaunchKernelGGL((mmq_mfma_k16_kernel<TYPE, OUT, 2, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
        else if constexpr (mmq_is_b32<TYPE>()) hipLaunchKernelGGL((mmq_mfma_b32_kernel<TYPE, OUT, 2, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((mmq_mfma_kernel<TYPE, OUT, 2, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
      }
      return;
    }
  }
  const unsigned gy = (unsigned)((ncols_max + NC - 1) / NC), gz = (unsigned)channels;
  // 4 rows per wave once that still leaves >= 2 workgroups per CU (256 CUs); small launches keep one row per wave to fill the chip
  if (((a.nrows_x + 15) / 16) * (int64_t)gy * gz >= 512) {
    hipLaunchKernelGGL((mmq_kernel<TYPE, OUT, NC, 4>), dim3((unsigned)((a.nrows_x + 15) / 16), gy, gz), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((mmq_kernel<TYPE, OUT, NC, 1>), dim3((unsigned)((a.nrows_x + 3) / 4), gy, gz), dim3(256), 0, (hipStream_t)stream, a);
  }
}

// dense: dst column stride = nrows_x -- the reference's launcher puts nrows_x into mmq_args.nrows_dst and never reads its stride_col_dst
// parameter (mmq_instance_<t>.cu:216-252; the caller passes nrows for both, fast_mmq.rs:426-430)
template <int TYPE>
static void launch_mmq_dense(const void *x, const void *y, void *dst, int64_t ncols_x, int64_t nrows_x, int64_t ncols_y, int64_t stride_row_x, int type_dst,
                             void *stream) {
  const MmqArgs a{(const uint8_t *)x, (const uint8_t *)y, dst, nullptr, nullptr, ncols_x, nrows_x, ncols_y, stride_row_x, nrows_x, 0};
  switch (type_dst) {
  case 0: launch_mmq_t<TYPE, float>(a, 1, ncols_y, stream); break;
  case 1: launch_mmq_t<TYPE, f16_t>(a, 1, ncols_y, stream); break;
  case 30: launch_mmq_t<TYPE, bf16_t>(a, 1, ncols_y, stream); break;
  default: break;
  }
}

// MoE: y holds the routes in expert-sorted order, expert e owns columns [bounds[e], bounds[e + 1]), its weights are channel e of x, the result
// of column j goes to dst column ids_dst[j]; f32 output (DEFINE_MMQ_MOE_LAUNCHER, mmq_gguf.cuh:3968-4010)
template <int TYPE>
static void launch_mmq_moe(const void *x, const void *y, const int32_t *ids_dst, const int32_t *expert_bounds, void *dst, int64_t ncols_x, int64_t nrows_x,
                           int64_t ncols_dst, int64_t stride_row_x, int64_t stride_col_dst, int64_t num_experts, int64_t ncols_max, void *stream) {
  const MmqArgs a{(const uint8_t *)x, (const uint8_t *)y, dst, ids_dst, expert_bounds, ncols_x, nrows_x, ncols_dst, stride_row_x, stride_col_dst,
                  nrows_x * stride_row_x};
  launch_mmq_t<TYPE, float>(a, num_experts, ncols_max, stream);
}

}  // namespace mrs

// tile policy of the matrix-core route (tests / measurements): launches with fewer than `n` 128 x 128 tiles take 64 x 64 tiles; 0 = always 128 x 128
extern "C" void mrs_mmq_set_small_tiles_below(int n) { mrs::g_mmq_small_below = n < 0 ? 0 : n; }

#define MRS_MMQ_QUANTIZE(NAME, LAYOUT)                                                                                                            \
  extern "C" void launch_mmq_quantize_q8_1_##NAME(const void *x, const int32_t *ids, void *vy, int type_x, int64_t ne00, int64_t s01, int64_t s02, \
                                                  int64_t s03, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, void *stream) {                 \
    mrs::launch_quantize<LAYOUT>(x, ids, vy, type_x, ne00, s01, s02, s03, ne0, ne1, ne2, ne3, stream);                                             \
  }                                                                                                                                                \
  extern "C" void launch_mmq_quantize_glu_q8_1_##NAME##_f32(const float *gate, const float *up, const int32_t *ids, void *vy, int64_t ne00,        \
                                                            int64_t s01, int64_t ne0, int64_t ne1, int activation, void *stream) {                 \
    mrs::launch_quantize_glu<LAYOUT>(gate, up, ids, vy, 0, ne00, s01, ne0, ne1, activation, stream);                                               \
  }                                                                                                                                                \
  extern "C" void launch_mmq_quantize_glu_q8_1_##NAME(const void *gate, const void *up, const int32_t *ids, void *vy, int type_x, int64_t ne00,    \
                                                      int64_t s01, int64_t ne0, int64_t ne1, int activation, void *stream) {                       \
    mrs::launch_quantize_glu<LAYOUT>(gate, up, ids, vy, type_x, ne00, s01, ne0, ne1, activation, stream);                                          \
  }
MRS_MMQ_QUANTIZE(D4, mrs::MMQ_D4)
MRS_MMQ_QUANTIZE(DS4, mrs::MMQ_DS4)
MRS_MMQ_QUANTIZE(D2S6, mrs::MMQ_D2S6)

#define MRS_MMQ_LAUNCHERS(TAG, TYPE)                                                                                                               \
  extern "C" void launch_mmq_gguf_##TAG(void *tmp_fixup, const void *x, const void *y, void *dst, int64_t ncols_x, int64_t nrows_x, int64_t ncols_y, \
                                        int64_t stride_row_x, int64_t stride_col_dst, int cc, int nsm, int64_t smpbo, int warp_size, int type_dst, \
                                        void *stream) {                                                                                            \
    (void)tmp_fixup; (void)stride_col_dst; (void)cc; (void)nsm; (void)smpbo; (void)warp_size;                                                      \
    mrs::launch_mmq_dense<TYPE>(x, y, dst, ncols_x, nrows_x, ncols_y, stride_row_x, type_dst, stream);                                             \
  }                                                                                                                                                \
  extern "C" void launch_mmq_gguf_##TAG##_moe(void *tmp_fixup, const void *x, const void *y, const int32_t *ids_dst, const int32_t *expert_bounds, \
                                              void *dst, int64_t ncols_x, int64_t nrows_x, int64_t ncols_dst, int64_t stride_row_x,               \
                                              int64_t stride_col_dst, int64_t num_experts, int64_t ncols_max, int cc, int nsm, int64_t smpbo,      \
                                              int warp_size, void *stream) {                                                                       \
    (void)tmp_fixup; (void)cc; (void)nsm; (void)smpbo; (void)warp_size;                                                                            \
    mrs::launch_mmq_moe<TYPE>(x, y, ids_dst, expert_bounds, dst, ncols_x, nrows_x, ncols_dst, stride_row_x, stride_col_dst, num_experts, ncols_max, \
                              stream);                                                                                                             \
  }
MRS_MMQ_LAUNCHERS(q4_0, mrs::T_Q4_0)
MRS_MMQ_LAUNCHERS(q4_1, mrs::T_Q4_1)
MRS_MMQ_LAUNCHERS(q5_0, mrs::T_Q5_0)
MRS_MMQ_LAUNCHERS(q5_1, mrs::T_Q5_1)
MRS_MMQ_LAUNCHERS(q8_0, mrs::T_Q8_0)
MRS_MMQ_LAUNCHERS(q2_k, mrs::T_Q2_K)
MRS_MMQ_LAUNCHERS(q3_k, mrs::T_Q3_K)
MRS_MMQ_LAUNCHERS(q4_k, mrs::T_Q4_K)
MRS_MMQ_LAUNCHERS(q5_k, mrs::T_Q5_K)
MRS_MMQ_LAUNCHERS(q6_k, mrs::T_Q6_K)
