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
};

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
  const uint8_t *ybase = a.y, *w0 = a.w[0];
  void *dst0 = a.dst[0];
  if constexpr (MODE == MODE_PLAIN && NCOLS == 1) {
    if (a.indices) {  // wave-uniform
      const int task = blockIdx.y;
      w0 += (size_t)a.indices[task] * a.expert_stride;
      ybase += (size_t)(a.input_dim1 == 1 ? task / a.topk : task) * a.stride_col_y * 36;
      dst0 = (float *)a.dst[0] + (size_t)task * a.nrows[0];
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
    auto rowptr = [&](int r, const uint8_t *&pA, const uint8_t *&pB) { pA = a.w[0] + (size_t)r * row_bytes; pB = a.w[1] + (size_t)r * row_bytes; };
    auto epi = [&](int r, float(&acc)[2][NCOLS]) {
      if (lane == 0) {
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
  }
