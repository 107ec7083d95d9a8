// moe.hip -- MoE prefill plumbing behind the reference's grouped path (gfx950): route dispatch (counting sort of the
// flattened top-k ids by expert) and the weighted reduce of per-route expert outputs.
//   replaces kernels/moe_grouped/moe_grouped.cu:630-702 (kernels) and :1104-1176 (launchers);
//   Rust: mistralrs-quant/src/gguf/ffi.rs:285-336; callers gguf/cuda.rs:590-640 (moe_dispatch_build), :640-930
//   (moe_weighted_reduce_flat*), used by FastExpertsWeights::forward_* (moe/experts/backends.rs:969-1100) for prompts.
// The grouped GEMM that consumes the dispatch tables lives with the MMVQ core (mmvq_kernels.cuh: moe_grouped_kernel).
//
// Dispatch on MI355X: the reference scatters with atomic cursors (three launches + a memset + a D2D copy; the order inside an
// expert's segment depends on the atomics).  Here one workgroup per expert makes the segment STABLE (ascending flat index -- one of
// the orders the reference can produce, and the one its kernels give when run sequentially): pass 1 counts the expert's routes
// with wave ballots, pass 2 writes them at `bounds[e] + rank`.  Every workgroup re-reads topk_ids (4 B per route) from L2;
// nothing else is touched, no atomics, two launches, graph-capturable, deterministic.
#include "common.cuh"

namespace mrs {

// launch 1: expert_counts[e] = #{i : topk_ids[i] == e}
__global__ void __launch_bounds__(256) moe_dispatch_count_kernel(const int *__restrict__ topk_ids, int *__restrict__ expert_counts, int total) {
  __shared__ int s_part[4];
  const int e = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int n = 0;
  for (int base = wave * 64; base < total; base += 256) {
    const int i = base + lane;
    const bool hit = i < total && topk_ids[i] == e;
    n += __popcll(__ballot(hit));
  }
  if (lane == 0) s_part[wave] = n;
  __syncthreads();
  if (threadIdx.x == 0) expert_counts[e] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// launch 2: bounds (exclusive prefix of the counts), final cursors (= bounds[e + 1], what the reference's atomics leave behind),
// and the stable scatter of expert e's routes.
__global__ void __launch_bounds__(256) moe_dispatch_scatter_kernel(const int *__restrict__ topk_ids, const int *__restrict__ expert_counts,
                                                                   int *__restrict__ expert_bounds, int *__restrict__ expert_cursors,
                                                                   int *__restrict__ sorted_token_ids, int *__restrict__ sorted_source_ids,
                                                                   int total, int num_experts, int topk) {
  __shared__ int s_red[4], s_wave_n[4], s_base;
  const int e = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // bounds[e] = sum of counts[0 .. e)
  int part = 0;
  for (int j = threadIdx.x; j < e; j += 256) part += expert_counts[j];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int b = s_red[0] + s_red[1] + s_red[2] + s_red[3], n = expert_counts[e];
    s_base = b;
    expert_bounds[e] = b;
    expert_cursors[e] = b + n;
    if (e == num_experts - 1) expert_bounds[num_experts] = b + n;
  }
  // wave w owns the contiguous quarter [w * span, (w + 1) * span) of the routes: count, then rank + write
  const int span = ((total + 3) / 4 + 63) & ~63;
  const int lo = wave * span, hi = min(total, lo + span);
  int n = 0;
  for (int base = lo; base < hi; base += 64) {
    const int i = base + lane;
    n += __popcll(__ballot(i < hi && topk_ids[i] == e));
  }
  if (lane == 0) s_wave_n[wave] = n;
  __syncthreads();
  int pos = s_base;
  for (int w = 0; w < wave; ++w) pos += s_wave_n[w];
  for (int base = lo; base < hi; base += 64) {
    const int i = base + lane;
    const bool hit = i < hi && topk_ids[i] == e;
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int p = pos + __popcll(m & ((1ull << lane) - 1ull));
      sorted_token_ids[p] = i;  // flat index into topk_ids: token = i / topk
      if (sorted_source_ids) sorted_source_ids[p] = i / topk;
    }
    pos += __popcll(m);
  }
}

// outputs[token][h] = OutT( sum_slot float(inputs[token][slot][h]) * topk_weights[token][slot] ), slots in order, f32 accumulate
// (moe_grouped.cu:678-702).  One thread per h: a wave reads / writes one contiguous run per slot; weights broadcast from LDS.
template <class InT, class OutT>
__global__ void __launch_bounds__(256) moe_weighted_reduce_flat_kernel(const InT *__restrict__ inputs, const float *__restrict__ topk_weights,
                                                                       OutT *__restrict__ outputs, int num_tokens, int hidden, int topk) {
  extern __shared__ float s_w[];
  const int token = blockIdx.x;
  for (int slot = threadIdx.x; slot < topk; slot += blockDim.x) s_w[slot] = topk_weights[(size_t)token * topk + slot];
  __syncthreads();
  const int h = blockIdx.y * blockDim.x + threadIdx.x;
  if (h >= hidden) return;
  const InT *in = inputs + (size_t)token * topk * hidden + h;
  float acc = 0.0f;
  for (int slot = 0; slot < topk; ++slot) acc += to_f(in[(size_t)slot * hidden]) * s_w[slot];
  outputs[(size_t)token * hidden + h] = from_f<OutT>(acc);
}

template <class InT, class OutT>
static int weighted_reduce(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden, int topk, void *stream) {
  if (num_tokens <= 0 || hidden <= 0) return (int)hipSuccess;
  (void)hipGetLastError();
  const int threads = 256;
  hipLaunchKernelGGL((moe_weighted_reduce_flat_kernel<InT, OutT>), dim3(num_tokens, 1 + (hidden - 1) / threads), dim3(threads),
                     (size_t)topk * sizeof(float), (hipStream_t)stream, (const InT *)inputs, topk_weights, (OutT *)outputs, num_tokens,
                     hidden, topk);
  return (int)hipGetLastError();
}

}  // namespace mrs

extern "C" void launch_moe_dispatch(const int32_t *topk_ids, int32_t *expert_bounds, int32_t *sorted_token_ids, int32_t *sorted_source_ids,
                                    int total_assignments, int num_experts, int topk, int32_t *expert_counts, int32_t *expert_cursors,
                                    void *stream) {
  if (num_experts <= 0) return;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mrs::moe_dispatch_count_kernel, dim3(num_experts), dim3(256), 0, s, topk_ids, expert_counts, total_assignments);
  hipLaunchKernelGGL(mrs::moe_dispatch_scatter_kernel, dim3(num_experts), dim3(256), 0, s, topk_ids, expert_counts, expert_bounds,
                     expert_cursors, sorted_token_ids, sorted_source_ids, total_assignments, num_experts, topk);
}

extern "C" int launch_moe_weighted_reduce_flat(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden,
                                               int topk, void *stream) {
  return mrs::weighted_reduce<float, float>(inputs, topk_weights, outputs, num_tokens, hidden, topk, stream);
}
extern "C" int launch_moe_weighted_reduce_flat_bf16(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden,
                                                    int topk, void *stream) {
  return mrs::weighted_reduce<float, mrs::bf16_t>(inputs, topk_weights, outputs, num_tokens, hidden, topk, stream);
}
extern "C" int launch_moe_weighted_reduce_flat_f16_input(const void *inputs, const float *topk_weights, void *outputs, int num_tokens,
                                                         int hidden, int topk, void *stream) {
  return mrs::weighted_reduce<mrs::f16_t, mrs::f16_t>(inputs, topk_weights, outputs, num_tokens, hidden, topk, stream);
}
extern "C" int launch_moe_weighted_reduce_flat_bf16_input(const void *inputs, const float *topk_weights, void *outputs, int num_tokens,
                                                          int hidden, int topk, void *stream) {
  return mrs::weighted_reduce<mrs::bf16_t, mrs::bf16_t>(inputs, topk_weights, outputs, num_tokens, hidden, topk, stream);
}
