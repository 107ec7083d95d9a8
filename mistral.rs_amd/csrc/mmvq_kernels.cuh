// mmvq_kernels.cuh -- __global__ kernels + host launch helpers behind the reference's
// `launch_mmvq_gguf_<type>_<dst>_{plain,fused_glu,fused_qkv}` C ABI
// (declared in mistralrs-quant/src/gguf/ffi.rs, defined in kernels/mmvq_gguf/mmvq_gguf.cu:1322-1600).
#pragma once
#include "mmvq_core.cuh"
#include <stdlib.h>

namespace mrs {

enum : int { DST_F32 = 0, DST_F16 = 1, DST_BF16 = 2 };
enum : int { MODE_PLAIN = 0, MODE_GLU = 1, MODE_QKV = 2 };

struct MmvqArgs {
  const uint8_t *w[3];  // plain: w[0]; glu: gate, up; qkv: q, k, v
  void *dst[3];
  int nrows[3];
  const uint8_t *y;  // Q8_1 blocks [b][stride_col_y]
  int ncols_x;       // K
  int stride_col_y;  // Q8_1 blocks per batch column
  int stride_col_dst;
  int activation;
  int dst_kind;
  int rows_per_wave;
  // indexed MoE forward (launch_indexed_moe_forward_<t>_q8_1, kernels/indexed_moe/indexed_moe.cu:806-890): blockIdx.y = task = token * topk + slot;
  // weights of expert indices[task] (stride expert_stride bytes), Q8_1 row (input_dim1 == 1 ? token : task), f32 output row `task`
  const uint32_t *indices;  // nullptr = dense launch
  size_t expert_stride;
  int topk, input_dim1;
  // fused MoE decode pair (launch_moe_gemv_fused_gate_up_<t>_q8_1 / launch_moe_gemv_down_aggregate_<t>_q8_1, indexed_moe.cu:1336-1477):
  // MOE_GATE_UP: MODE_GLU on expert indices[task], Q8_1 row of the TOKEN (task / topk), f32 out[task][n] = up * act(gate), act_type 0 =
  // gelu_pytorch_tanh, anything else = silu;  MOE_DOWN_AGG: MODE_PLAIN on expert indices[task], Q8_1 row `task`,
  // atomicAdd(out[token][row], dot * topk_weights[task]) -- the caller zero-fills out (gguf/cuda.rs fused decode path)
  int moe_kind;
  const float *topk_weights;
};
enum : int { MOE_FORWARD = 0, MOE_GATE_UP = 1, MOE_DOWN_AGG = 2 };

__device__ __forceinline__ void store_dst(void *dst, size_t idx, float v, int kind) {
  if (kind == DST_F32) ((float *)dst)[idx] = v;
  else if (kind == DST_F16) ((f16_t *)dst)[idx] = (f16_t)v;
  else ((uint16_t *)dst)[idx] = float_to_bf16_bits(v);
}
__device__ __forceinline__ float round_kind(float v, int kind) {
  if (kind == DST_F32) return v;
  if (kind == DST_F16) return (float)(f16_t)v;
  return bf16_bits_to_float(float_to_bf16_bits(v));
}

template <int TYPE, int NCOLS, int MODE>
__global__ void __launch_bounds__(256) mmvq_kernel(const MmvqArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = a.ncols_x;
  const size_t row_bytes = (size_t)(K / Fmt<TYPE>::BLK) * Fmt<TYPE>::TS;
  const int total_rows = (MODE == MODE_QKV) ? a.nrows[0] + a.nrows[1] + a.nrows[2] : a.nrows[0];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int rpw = a.rows_per_wave;  // contiguous rows per wave: each wave streams one contiguous byte range
  const int first = (blockIdx.x * 4 + wave) * rpw;
  const int nrows = max(0, min(rpw, total_rows - first));
  const uint8_t *ybase = a.y, *w0 = a.w[0], *w1 = a.w[1];
  void *dst0 = a.dst[0];
  float moe_weight = 1.0f;
  bool aggregate = false;
  if constexpr (MODE != MODE_QKV && NCOLS == 1) {
    if (a.indices) {  // wave-uniform
      const int task = blockIdx.y, token = task / a.topk;
      const size_t expert_off = (size_t)a.indices[task] * a.expert_stride;
      w0 += expert_off;
      if constexpr (MODE == MODE_GLU) w1 += expert_off;
      const bool per_token_input = (a.moe_kind == MOE_FORWARD) ? a.input_dim1 == 1 : a.moe_kind == MOE_GATE_UP;
      ybase += (size_t)(per_token_input ? token : task) * a.stride_col_y * 36;
      aggregate = a.moe_kind == MOE_DOWN_AGG;
      dst0 = (float *)a.dst[0] + (size_t)(aggregate ? token : task) * a.nrows[0];
      if (aggregate) moe_weight = a.topk_weights[task];
    }
  }
  auto pro = [&]() {
    const ActLds act = stage_q8_1<TYPE, NCOLS>(smem, ybase, K, a.stride_col_y);
    __syncthreads();
    return act;
  };
  if constexpr (MODE == MODE_GLU) {
    // reference: mmvq_core_fused_glu_impl (mmvq_gguf.cu:794-873): both projections are rounded
    // to dst_t, the activation runs in f32 on the rounded gate, is rounded again, then multiplied.
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = w0 + (size_t)r * row_bytes; pB = w1 + (size_t)r * row_bytes; };
    auto epi = [&](int r, float(&acc)[2][NCOLS]) {
      if (lane == 0) {
        if (NCOLS == 1 && a.indices) {  // fused MoE gate/up: f32, no intermediate rounding (indexed_moe.cu:1400-1405)
          ((float *)dst0)[r] = acc[1][0] * glu_act(acc[0][0], a.activation == 0 ? 1 : 0);
          return;
        }
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
          const float gv = round_kind(acc[0][c], a.dst_kind), uv = round_kind(acc[1][c], a.dst_kind);
          const float av = round_kind(glu_act(gv, a.activation), a.dst_kind);
          store_dst(a.dst[0], (size_t)c * a.stride_col_dst + r, av * uv, a.dst_kind);
        }
      }
    };
    stream_rows_auto<TYPE, NCOLS, true>(first, nrows, 1, K, rpw, rowptr, pro, epi);
  } else {
    // MODE_QKV: mmvq_core_fused_qkv_impl (:875-996): virtual row r -> (matrix, local row), dst[j*nrows_x + row]
    auto locate = [&](int r, int &m, int &lr) {
      m = 0; lr = r;
      if constexpr (MODE == MODE_QKV) {
        if (r >= a.nrows[0] + a.nrows[1]) { m = 2; lr = r - a.nrows[0] - a.nrows[1]; }
        else if (r >= a.nrows[0]) { m = 1; lr = r - a.nrows[0]; }
      }
    };
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) {
      int m, lr;
      locate(r, m, lr);
      pA = (MODE == MODE_PLAIN ? w0 : a.w[m]) + (size_t)lr * row_bytes;
      pB = pA + row_bytes;  // row pairs (paired-row path): r + 1 lies in the same matrix (every row count is even there)
    };
    auto store_row = [&](int r, const float(&v)[NCOLS]) {
      int m, lr;
      locate(r, m, lr);
      const int stride = (MODE == MODE_QKV) ? a.nrows[m] : a.stride_col_dst;
      if (MODE == MODE_PLAIN && NCOLS == 1 && aggregate) {  // indexed_moe.cu:1473-1475
        atomicAdd((float *)dst0 + lr, v[0] * moe_weight);
        return;
      }
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) store_dst(MODE == MODE_PLAIN ? dst0 : a.dst[m], (size_t)c * stride + lr, v[c], a.dst_kind);
    };
    bool paired = false;
    if constexpr (PairQ<TYPE>::value) {  // Q4_K / Q5_K: two rows per step, 64-weight arithmetic per lane (mmvq_core.cuh)
      paired = ((rpw | a.nrows[0] | a.nrows[1] | a.nrows[2]) & 1) == 0;
      if (paired) {
        auto epi2 = [&](int r, float(&acc)[2][NCOLS]) { if (lane == 0) { store_row(r, acc[0]); store_row(r + 1, acc[1]); } };
        stream_rows_auto<TYPE, NCOLS, true>(first, nrows / 2, 2, K, rpw / 2, rowptr, pro, epi2);
      }
    }
    if (!paired) {
      auto epi = [&](int r, float(&acc)[1][NCOLS]) { if (lane == 0) store_row(r, acc[0]); };
      stream_rows_auto<TYPE, NCOLS, false>(first, nrows, 1, K, rpw, rowptr, pro, epi);
    }
  }
}

// launch_moe_grouped_gemm_<t> (kernels/moe_grouped/moe_grouped.cu:704-865,1180-1235): for every expert e the assignments
// sorted_token_ids[expert_bounds[e] .. expert_bounds[e+1]) are multiplied with W[e].  The reference walks them in 64-token tiles of an
// int8 MMQ-style block; here a workgroup owns a slice of rows of ONE expert (blockIdx.y) -- a few KiB of packed weights that stay
// L1/L2-resident -- and streams its expert's assignments through the batch-8 MMVQ core, 8 gathered Q8_1 rows per pass (expert_bounds
// lives on the device, so the pass count is data-dependent and cannot be a grid dimension).  Same per-assignment dot arithmetic as the
// plain MMVQ launch; sorted position ti, flat = sorted_token_ids[ti]:
//   input row  = input_dim1 == 0 ? ti : input_dim1 == 1 ? flat / topk : flat
//   output     = topk_weights ? atomicAdd(out[flat / topk][row], acc * topk_weights[flat]) : out[ti][row] = acc
struct MoeGroupedArgs {
  const uint8_t *w, *y;
  const int *expert_bounds, *sorted_token_ids;
  const float *topk_weights;
  float *out;
  int N, K, stride_col_y, num_experts, topk, input_dim1, rows_per_wave;
  size_t expert_stride;
};

template <int TYPE, int NC>
__global__ void __launch_bounds__(256) moe_grouped_kernel(const MoeGroupedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = a.K, N = a.N;
  int *s_rows = (int *)(smem + act_lds_bytes(K, NC)), *s_flat = s_rows + NC;  // behind the activation view (launch adds 64 B)
  const size_t row_bytes = (size_t)(K / Fmt<TYPE>::BLK) * Fmt<TYPE>::TS;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int rpw = a.rows_per_wave;
  const int first = (blockIdx.x * 4 + wave) * rpw;
  const int nrows = max(0, min(rpw, N - first));
  const int expert = blockIdx.y;
  const int t_begin = a.expert_bounds[expert], t_end = a.expert_bounds[expert + 1];  // workgroup-uniform
  const uint8_t *w0 = a.w + (size_t)expert * a.expert_stride;
  for (int t0 = t_begin; t0 < t_end; t0 += NC) {
    const int cnt = min(NC, t_end - t0);
    __syncthreads();  // every wave is done with the previous pass's LDS
    if (threadIdx.x < NC) {
      const int ti = t0 + min((int)threadIdx.x, cnt - 1);  // short last pass: the spare columns repeat the last assignment (never stored)
      const int flat = a.sorted_token_ids[ti];
      s_flat[threadIdx.x] = flat;
      s_rows[threadIdx.x] = a.input_dim1 == 0 ? ti : (a.input_dim1 == 1 ? flat / a.topk : flat);
    }
    __syncthreads();
    auto pro = [&]() {
      const ActLds act = stage_q8_1<TYPE, NC>(smem, a.y, K, a.stride_col_y, s_rows);
      __syncthreads();
      return act;
    };
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = w0 + (size_t)r * row_bytes; pB = pA + row_bytes; };
    auto store_row = [&](int r, const float(&v)[NC]) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c < cnt) {
          const int flat = s_flat[c];
          if (a.topk_weights) atomicAdd(a.out + (size_t)(flat / a.topk) * N + r, v[c] * a.topk_weights[flat]);
          else a.out[(size_t)(t0 + c) * N + r] = v[c];
        }
      }
    };
    bool paired = false;
    if constexpr (PairQ<TYPE>::value) {
      paired = ((rpw | N) & 1) == 0;
      if (paired) {
        auto epi2 = [&](int r, float(&acc)[2][NC]) { if (lane == 0) { store_row(r, acc[0]); store_row(r + 1, acc[1]); } };
        stream_rows_auto<TYPE, NC, true>(first, nrows / 2, 2, K, rpw / 2, rowptr, pro, epi2);
      }
    }
    if (!paired) {
      auto epi = [&](int r, float(&acc)[1][NC]) { if (lane == 0) store_row(r, acc[0]); };
      stream_rows_auto<TYPE, NC, false>(first, nrows, 1, K, rpw, rowptr, pro, epi);
    }
  }
}

template <int TYPE, int NC> inline void moe_grouped_launch(const MoeGroupedArgs &a, int grid, hipStream_t s) {
  const size_t lds = act_lds_bytes(a.K, NC) + 64;
  static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
  if (lds > 65536 && !attr_done) {
    hipFuncSetAttribute((const void *)moe_grouped_kernel<TYPE, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((moe_grouped_kernel<TYPE, NC>), dim3(grid, a.num_experts), dim3(256), lds, s, a);
}

// rows per wave so that the grid is ~16 waves on each of the 256 CUs (all resident at once: no tail wave)
inline int mmvq_target_waves() {
  static int v = 0;
  if (!v) { const char *e = getenv("MRS_MMVQ_WAVES"); v = e ? atoi(e) : 4096; if (v < 256) v = 256; }
  return v;
}
inline int mmvq_rows_per_wave(int total_rows, bool want_even) {
  const int t = mmvq_target_waves();
  int r = (total_rows + t - 1) / t;
  if (r < 1) r = 1;
  if (want_even && (r & 1)) ++r;  // paired-row formats walk two rows per step
  return r;
}

template <int TYPE, int MODE> struct MmvqLaunch {
  template <int NCOLS> static void go(const MmvqArgs &a, int total_rows, hipStream_t s, int tasks = 1) {
    const size_t lds = act_lds_bytes(a.ncols_x, NCOLS, Fmt<TYPE>::HAS_OFFSET);
    static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (lds > 65536 && !attr_done) {
      hipFuncSetAttribute((const void *)mmvq_kernel<TYPE, NCOLS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done = true;
    }
    MmvqArgs b = a;
    b.rows_per_wave = mmvq_rows_per_wave(total_rows, PairQ<TYPE>::value && MODE != MODE_GLU);
    const int grid = (total_rows + 4 * b.rows_per_wave - 1) / (4 * b.rows_per_wave);
    hipLaunchKernelGGL((mmvq_kernel<TYPE, NCOLS, MODE>), dim3(grid, tasks), dim3(256), lds, s, b);
  }
  static void run(const MmvqArgs &a, int b_size, hipStream_t s) {
    const int total_rows = (MODE == MODE_QKV) ? a.nrows[0] + a.nrows[1] + a.nrows[2] : a.nrows[0];
    if (total_rows <= 0) return;
    switch (b_size) {
    case 1: go<1>(a, total_rows, s); break;
    case 2: go<2>(a, total_rows, s); break;
    case 3: go<3>(a, total_rows, s); break;
    case 4: go<4>(a, total_rows, s); break;
    case 5: go<5>(a, total_rows, s); break;
    case 6: go<6>(a, total_rows, s); break;
    case 7: go<7>(a, total_rows, s); break;
    case 8: go<8>(a, total_rows, s); break;
    default: break;  // the reference launcher silently ignores b_size outside 1..8 as well
    }
  }
};

}  // namespace mrs

// Defines the nine C-ABI launchers of one GGUF type (3 dst dtypes x plain / fused_glu / fused_qkv).
#define MRS_MMVQ_LAUNCHERS_DST(tag, TYPE, dtag, DKIND)                                                              \
  extern "C" void launch_mmvq_gguf_##tag##_##dtag##_plain(const void *vx, const void *vy, void *dst, int ncols_x,   \
                                                         int nrows_x, int stride_col_y, int stride_col_dst,        \
                                                         int b_size, void *stream) {                               \
    mrs::MmvqArgs a{};                                                                                              \
    a.w[0] = (const uint8_t *)vx; a.dst[0] = dst; a.nrows[0] = nrows_x; a.y = (const uint8_t *)vy;                  \
    a.ncols_x = ncols_x; a.stride_col_y = stride_col_y; a.stride_col_dst = stride_col_dst; a.dst_kind = DKIND;      \
    mrs::MmvqLaunch<TYPE, mrs::MODE_PLAIN>::run(a, b_size, (hipStream_t)stream);                                    \
  }                                                                                                                 \
  extern "C" void launch_mmvq_gguf_##tag##_##dtag##_fused_glu(const void *vx_gate, const void *vx_up,              \
                                                             const void *vy, void *dst, int ncols_x, int nrows_x,  \
                                                             int stride_col_y, int stride_col_dst, int b_size,     \
                                                             int activation, void *stream) {                       \
    mrs::MmvqArgs a{};                                                                                              \
    a.w[0] = (const uint8_t *)vx_gate; a.w[1] = (const uint8_t *)vx_up; a.dst[0] = dst; a.nrows[0] = nrows_x;       \
    a.y = (const uint8_t *)vy; a.ncols_x = ncols_x; a.stride_col_y = stride_col_y;                                  \
    a.stride_col_dst = stride_col_dst; a.activation = activation; a.dst_kind = DKIND;                               \
    mrs::MmvqLaunch<TYPE, mrs::MODE_GLU>::run(a, b_size, (hipStream_t)stream);                                      \
  }                                                                                                                 \
  extern "C" void launch_mmvq_gguf_##tag##_##dtag##_fused_qkv(                                                      \
      const void *vx_q, const void *vx_k, const void *vx_v, const void *vy, void *q_dst, void *k_dst, void *v_dst, \
      int ncols_x, int nrows_q, int nrows_k, int nrows_v, int stride_col_y, int b_size, void *stream) {            \
    mrs::MmvqArgs a{};                                                                                              \
    a.w[0] = (const uint8_t *)vx_q; a.w[1] = (const uint8_t *)vx_k; a.w[2] = (const uint8_t *)vx_v;                 \
    a.dst[0] = q_dst; a.dst[1] = k_dst; a.dst[2] = v_dst;                                                           \
    a.nrows[0] = nrows_q; a.nrows[1] = nrows_k; a.nrows[2] = nrows_v;                                               \
    a.y = (const uint8_t *)vy; a.ncols_x = ncols_x; a.stride_col_y = stride_col_y; a.dst_kind = DKIND;              \
    mrs::MmvqLaunch<TYPE, mrs::MODE_QKV>::run(a, b_size, (hipStream_t)stream);                                      \
  }

#define MRS_MMVQ_LAUNCHERS(tag, TYPE)                    \
  MRS_MMVQ_LAUNCHERS_DST(tag, TYPE, f32, mrs::DST_F32)   \
  MRS_MMVQ_LAUNCHERS_DST(tag, TYPE, f16, mrs::DST_F16)   \
  MRS_MMVQ_LAUNCHERS_DST(tag, TYPE, bf16, mrs::DST_BF16)


// launch_indexed_moe_forward_<moe tag>_q8_1 (mistralrs-quant/src/gguf/ffi.rs:100-260; kernels/indexed_moe/indexed_moe.cu:806-1157):
// all_weights [E][n][k / blk] packed, all_inputs Q8_1 rows of k_padded / 32 blocks ([batch] when input_dim1 == 1, else [batch * topk]),
// indices [batch * topk] expert ids, all_outputs f32 [batch * topk][n].  Caller: GgufMatMul::gather_forward_raw -> qmatmul_indexed_moe_forward
// (gguf/mod.rs:485-516, gguf/cuda.rs:514-588).  Same dot-product arithmetic as the plain MMVQ launch of one expert.
#define MRS_INDEXED_MOE_LAUNCHER(moetag, TYPE) MRS_INDEXED_MOE_LAUNCHER_(moetag, TYPE)  /* expand the tag macro before pasting */
#define MRS_INDEXED_MOE_LAUNCHER_(moetag, TYPE)                                                                                        \
  extern "C" void launch_indexed_moe_forward_##moetag##_q8_1(const void *all_weights, const void *all_inputs, const unsigned int *indices, \
                                                             float *all_outputs, int n, int k, int batch, int topk, int k_padded,       \
                                                             int input_dim1, void *stream) {                                          \
    if (n <= 0 || batch <= 0 || topk <= 0) return;                                                                                    \
    mrs::MmvqArgs a{};                                                                                                                \
    a.w[0] = (const uint8_t *)all_weights; a.dst[0] = all_outputs; a.nrows[0] = n; a.y = (const uint8_t *)all_inputs;                  \
    a.ncols_x = k; a.stride_col_y = k_padded / 32; a.stride_col_dst = n; a.dst_kind = mrs::DST_F32;                                    \
    a.indices = indices; a.topk = topk; a.input_dim1 = input_dim1;                                                                    \
    a.expert_stride = (size_t)n * (size_t)(k / mrs::Fmt<TYPE>::BLK) * mrs::Fmt<TYPE>::TS;                                              \
    mrs::MmvqLaunch<TYPE, mrs::MODE_PLAIN>::go<1>(a, n, (hipStream_t)stream, batch * topk);                                           \
  }                                                                                                                                   \
  /* launch_moe_grouped_gemm_<t> (gguf/ffi.rs:338-500; kernels/moe_grouped/moe_grouped.cu:1180-1235): see moe_grouped_kernel */      \
  extern "C" void launch_moe_grouped_gemm_##moetag(const void *all_weights, const void *all_inputs, const int32_t *expert_bounds,     \
                                                   const int32_t *sorted_token_ids, const float *topk_weights, float *all_outputs,    \
                                                   int N, int K, int K_padded, int num_experts, int topk, int input_dim1,              \
                                                   void *stream) {                                                                    \
    if (N <= 0 || num_experts <= 0) return;                                                                                           \
    mrs::MoeGroupedArgs a{};                                                                                                          \
    a.w = (const uint8_t *)all_weights; a.y = (const uint8_t *)all_inputs; a.expert_bounds = expert_bounds;                            \
    a.sorted_token_ids = sorted_token_ids; a.topk_weights = topk_weights; a.out = all_outputs; a.N = N; a.K = K;                       \
    a.stride_col_y = K_padded / 32; a.num_experts = num_experts; a.topk = topk; a.input_dim1 = input_dim1;                             \
    a.expert_stride = (size_t)N * (size_t)(K / mrs::Fmt<TYPE>::BLK) * mrs::Fmt<TYPE>::TS;                                              \
    a.rows_per_wave = mrs::mmvq_rows_per_wave(N, mrs::PairQ<TYPE>::value);                                                            \
    const int grid = (N + 4 * a.rows_per_wave - 1) / (4 * a.rows_per_wave);                                                           \
    /* assignments per pass: as many Q8_1 rows as fit the 160 KiB LDS (8 up to K = 14848, 4 up to 29696, else 2) */                  \
    if (mrs::act_lds_bytes(K, 8) + 64 <= 160 * 1024) mrs::moe_grouped_launch<TYPE, 8>(a, grid, (hipStream_t)stream);                   \
    else if (mrs::act_lds_bytes(K, 4) + 64 <= 160 * 1024) mrs::moe_grouped_launch<TYPE, 4>(a, grid, (hipStream_t)stream);              \
    else mrs::moe_grouped_launch<TYPE, 2>(a, grid, (hipStream_t)stream);                                                              \
  }                                                                                                                                   \
  /* launch_moe_gemv_fused_gate_up_<t>_q8_1 (gguf/ffi.rs:520-700; kernels/indexed_moe/indexed_moe.cu:1336-1407,1618-1645):            \
     out[task][row] = (up_w[e][row] . y[token]) * act(gate_w[e][row] . y[token]), e = indices[task], task = token * topk + slot */    \
  extern "C" void launch_moe_gemv_fused_gate_up_##moetag##_q8_1(const void *gate_weights, const void *up_weights, const void *all_inputs, \
                                                                const unsigned int *indices, float *all_outputs, int n, int k, int batch, \
                                                                int topk, int k_padded, int act_type, void *stream) {                  \
    if (n <= 0 || batch <= 0 || topk <= 0) return;                                                                                    \
    mrs::MmvqArgs a{};                                                                                                                \
    a.w[0] = (const uint8_t *)gate_weights; a.w[1] = (const uint8_t *)up_weights; a.dst[0] = all_outputs; a.nrows[0] = n;               \
    a.y = (const uint8_t *)all_inputs; a.ncols_x = k; a.stride_col_y = k_padded / 32; a.stride_col_dst = n; a.dst_kind = mrs::DST_F32; \
    a.indices = indices; a.topk = topk; a.moe_kind = mrs::MOE_GATE_UP; a.activation = act_type;                                        \
    a.expert_stride = (size_t)n * (size_t)(k / mrs::Fmt<TYPE>::BLK) * mrs::Fmt<TYPE>::TS;                                              \
    mrs::MmvqLaunch<TYPE, mrs::MODE_GLU>::go<1>(a, n, (hipStream_t)stream, batch * topk);                                             \
  }                                                                                                                                   \
  /* launch_moe_gemv_down_aggregate_<t>_q8_1 (gguf/ffi.rs:716-890; indexed_moe.cu:1409-1477,1699-1714):                               \
     out[token][row] += topk_weights[task] * (w[e][row] . y[task]) with f32 atomics; out is zero-filled by the caller */              \
  extern "C" void launch_moe_gemv_down_aggregate_##moetag##_q8_1(const void *all_weights, const void *all_inputs,                      \
                                                                 const unsigned int *indices, const float *topk_weights,              \
                                                                 float *all_outputs, int n, int k, int batch, int topk, int k_padded,  \
                                                                 void *stream) {                                                      \
    if (n <= 0 || batch <= 0 || topk <= 0) return;                                                                                    \
    mrs::MmvqArgs a{};                                                                                                                \
    a.w[0] = (const uint8_t *)all_weights; a.dst[0] = all_outputs; a.nrows[0] = n; a.y = (const uint8_t *)all_inputs;                  \
    a.ncols_x = k; a.stride_col_y = k_padded / 32; a.stride_col_dst = n; a.dst_kind = mrs::DST_F32;                                    \
    a.indices = indices; a.topk = topk; a.moe_kind = mrs::MOE_DOWN_AGG; a.topk_weights = topk_weights;                                 \
    a.expert_stride = (size_t)n * (size_t)(k / mrs::Fmt<TYPE>::BLK) * mrs::Fmt<TYPE>::TS;                                              \
    mrs::MmvqLaunch<TYPE, mrs::MODE_PLAIN>::go<1>(a, n, (hipStream_t)stream, batch * topk);                                           \
  }
