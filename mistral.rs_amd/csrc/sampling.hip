// sampling.hip -- libmistralrscuda, sampling subset: top-k of one logits row over a 100k+ vocabulary + the pieces of the full-softmax normaliser the host
// sampler needs (Sampler::sample_topk_on_device, mistralrs-core/src/sampler.rs:1171-1260: top-p / min-p / the multinomial draw stay on the host).
//   top1_large_f32_packed, top1_large_f32_packed_batched                     sort.cu:1825-1912,2071-2143,2207-2238 ; ffi.rs:643-665 ; caller ops.rs:1232-2000 (greedy)
//   topk_large_f32, topk_large_f32_packed, topk_large_f32_packed_batched      mistralrs-core/src/cuda/sort.cu:1502-1823,2146-2206 ; ffi.rs:583-624 ;
//                                                                              caller ops.rs:691-828 (cuda_topk_logits_f32_packed)
// Contract kept from the reference (the caller owns every buffer):
//   stage 1, one workgroup per `chunk_size` logits: block_values / block_indices [chunk][k] = the chunk's k largest logits in (value descending, index
//   ascending) order, NaN and -inf never selected, missing entries (-inf, 0); block_maxes[chunk] = top value * inv_temperature (-inf for an empty chunk);
//   block_sums[chunk] = sum over the chunk of expf(x * inv_temperature - block_max) (NaN if the chunk holds one);
//   stage 2, one workgroup per row: global max of block_maxes, denom = sum_b block_sums[b] * expf(block_maxes[b] - max), and the k best candidates in the
//   same order; packed_out = [k values][k indices as f32][denom][max].
// MI355X design (not the reference's k rounds of scan-the-chunk + two block barriers each):
//   stage 1: a chunk lives in REGISTERS (<= 16 logits per thread), every wave extracts the top-k of its quarter on its own -- one 64-bit key per candidate
//   (order-preserving float bits << 32 | ~index, so ONE max reduction per round settles value and tie; DPP + v_readlane, no LDS crossbar), each lane's keys
//   sorted once so that its best is keys[0], no barrier -- then wave 0 merges the four sorted lists by heads (one lane per list).  stage 2 is the same head merge over the nblocks sorted lists (a lane per list, lists beyond
//   64 share lanes): k rounds of one wave-wide max instead of k scans of nblocks * k candidates.
//   The f32 sums keep the reference's association (per-thread strided partials, 32-lane shuffle-down trees, warp sums through LDS) so that only expf's last
//   ulp separates the normaliser from the CUDA build's.
#include "common.cuh"
#include <stdint.h>
#include <algorithm>

namespace mrs {
namespace sampling {

constexpr int NT = 256;       // threads per workgroup (the reference's block size: the strided partial sums depend on it)
constexpr int MAXV = 16;      // logits per thread kept in registers: chunks up to 4096
constexpr int MAX_K = 128;    // CUDA_TOPK_MAX_K (ops.rs:18)

// One 64-bit key per candidate: [63:32] the value's bits mapped to an order-preserving unsigned (with -0.0 == +0.0, as the reference's `>` sees them),
// [31:1] ~index (so the LOWER index wins a tie; indices < 2^31), [0] "the value was -0.0" (to give back the original bits).  0 = not a candidate.
__device__ __forceinline__ unsigned long long key_of(float v, unsigned idx) {
  // NaN and -inf are never candidates (sort.cu:1552: candidate == candidate && candidate > -INFINITY)
  if (!(v == v) || v == -INFINITY) return 0ull;
  const unsigned u = __float_as_uint(v + 0.0f);  // -0.0 -> +0.0
  const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  const unsigned low = ((~idx) << 1) | (__float_as_uint(v) == 0x80000000u ? 1u : 0u);
  return ((unsigned long long)o << 32) | (unsigned long long)low;
}
__device__ __forceinline__ unsigned idx_of(unsigned long long key) { return (~((unsigned)key >> 1)) & 0x7fffffffu; }
__device__ __forceinline__ float val_of(unsigned long long key) {  // the candidate's original bits
  const unsigned o = (unsigned)(key >> 32);
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(((unsigned)key & 1u) ? 0x80000000u : u);
}
// max over the wave, every lane gets it: DPP inside rows of 16 (xor 1, xor 2, half-row mirror, row mirror), then the four row results through v_readlane -- no LDS crossbar
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long k, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  unsigned long long o;
  o = dpp_u64<0xB1>(k); k = o > k ? o : k;
  o = dpp_u64<0x4E>(k); k = o > k ? o : k;
  o = dpp_u64<0x141>(k); k = o > k ? o : k;
  o = dpp_u64<0x140>(k); k = o > k ? o : k;
  const unsigned long long r0 = readlane_u64(k, 0), r1 = readlane_u64(k, 16), r2 = readlane_u64(k, 32), r3 = readlane_u64(k, 48);
  const unsigned long long a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}
// the reference's reductions, association for association (sort.cu:1470-1497): 32-lane shuffle-down trees, warp sums through shared memory, first warp again
__device__ __forceinline__ float warp32_sum_down(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
  return v;
}
__device__ __forceinline__ float block_sum_ref_order(float v, float *warp_sums /* [32] */) {
  const int tid = threadIdx.x, warp_id = tid / 32, lane_id = tid % 32;
  v = warp32_sum_down(v);
  if (lane_id == 0) warp_sums[warp_id] = v;
  __syncthreads();
  v = tid < NT / 32 ? warp_sums[tid] : 0.0f;
  if (tid < 64) v = warp32_sum_down(v);  // the whole first wave walks the tree (lanes 32..63 reduce zeros): lane 0 sees the reference's first-warp tree
  __syncthreads();
  return v;  // valid in thread 0
}

struct Stage1Args {
  const float *input;
  float *block_values;
  uint32_t *block_indices;
  float *block_maxes, *block_sums;
  const float *inv_temperatures;  // batched: one per row
  float inv_temperature;
  int ncols, k, chunk_size, nblocks;
};

// merge `nl` lists sorted by key (descending) into the k best: lane l walks list l (l, l + 64, ... when nl > 64) by its head and keeps the best of its heads in a
// register; one wave-wide max per round, and only the WINNING lane does anything else (win(ki, list, pos), advance its head, refresh its best) -- no broadcast.
// `get(list, pos)` returns the key (0 = exhausted); without a winner lane 0 calls win(ki, -1, 0).  One wave.
template <class Get, class Win>
__device__ __forceinline__ void head_merge(int nl, int k, int *heads /* LDS [nl] */, Get get, Win win) {
  const int lane = threadIdx.x & 63;
  for (int l = lane; l < nl; l += 64) heads[l] = 0;  // a lane only ever touches the heads of its own lists: no cross-lane traffic through LDS
  int bl = -1;
  auto my_best = [&]() {
    unsigned long long b = 0ull;
    bl = -1;
    for (int l = lane; l < nl; l += 64) {
      const int h = heads[l];
      const unsigned long long key = h < k ? get(l, h) : 0ull;
      if (key > b) { b = key; bl = l; }
    }
    return b;
  };
  unsigned long long best = my_best();
  for (int ki = 0; ki < k; ++ki) {
    const unsigned long long w = wave_max_u64(best);
    if (w != 0ull && best == w) {  // keys are unique (they carry the index): exactly one lane
      const int pos = heads[bl];
      win(ki, bl, pos);
      heads[bl] = pos + 1;
      best = my_best();
    } else if (w == 0ull && lane == 0) {
      win(ki, -1, 0);
    }
  }
}

template <bool BATCHED>
__global__ void __launch_bounds__(NT) topk_stage1_kernel(Stage1Args a) {
  __shared__ unsigned long long s_keys[4][MAX_K];  // each wave's sorted candidates
  __shared__ int s_heads[4];
  __shared__ float s_warp_sums[32];
  __shared__ float s_block_max;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = BATCHED ? blockIdx.y : 0;
  const int chunk = blockIdx.x, k = a.k;
  const float *input = a.input + row * (size_t)a.ncols;
  float *bv = a.block_values + (row * a.nblocks + chunk) * (size_t)k;
  uint32_t *bi = a.block_indices + (row * a.nblocks + chunk) * (size_t)k;
  const float inv_t = BATCHED ? a.inv_temperatures[row] : a.inv_temperature;
  const int start = chunk * a.chunk_size, end = min(start + a.chunk_size, a.ncols), width = max(0, end - start);
  // the chunk in registers, in the reference's thread-strided assignment (local = tid + 256 j): needed as it is for the partial sums below
  float v[MAXV];
  unsigned long long keys[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int local = tid + j * NT;
    v[j] = local < width ? input[start + local] : -INFINITY;
    keys[j] = local < width ? key_of(v[j], (unsigned)(start + local)) : 0ull;
  }
  // ---- every wave: the k best of its 64 x MAXV logits, sorted.  Each lane first sorts ITS keys (descending, a bitonic network in registers), so that its
  // best unused key is always keys[0]; the round's winner shifts its array down by one.  A round = one wave-wide 64-bit max + 2 * MAXV moves in one lane.
#pragma unroll
  for (int size = 2; size <= MAXV; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1)
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        const int p = j ^ stride;
        if (p > j) {
          const bool desc = (j & size) == 0;  // direction of this bitonic block
          const unsigned long long a = keys[j], b = keys[p];
          const bool sw = desc ? a < b : a > b;
          keys[j] = sw ? b : a;
          keys[p] = sw ? a : b;
        }
      }
  for (int ki = 0; ki < k; ++ki) {
    const unsigned long long win = wave_max_u64(keys[0]);
    if (win != 0ull && keys[0] == win) {  // keys carry the index: exactly one lane
      s_keys[wave][ki] = win;
#pragma unroll
      for (int j = 0; j + 1 < MAXV; ++j) keys[j] = keys[j + 1];
      keys[MAXV - 1] = 0ull;
    } else if (win == 0ull && lane == 0) {
      s_keys[wave][ki] = 0ull;
    }
  }
  __syncthreads();
  // ---- wave 0: merge the four lists
  if (wave == 0) {
    head_merge(4, k, s_heads, [&](int l, int h) { return s_keys[l][h]; },
               [&](int ki, int wl, int wp) {  // the winning lane (or lane 0 when nothing is left)
                 const unsigned long long key = wl >= 0 ? s_keys[wl][wp] : 0ull;
                 bv[ki] = wl >= 0 ? val_of(key) : -INFINITY;
                 bi[ki] = wl >= 0 ? idx_of(key) : 0u;
                 if (ki == 0) s_block_max = width > 0 ? (wl >= 0 ? val_of(key) : -INFINITY) * inv_t : -INFINITY;
               });
  }
  __syncthreads();
  // ---- the chunk's share of the softmax normaliser (sort.cu:1580-1597)
  const float block_max = s_block_max;
  float local_sum = 0.0f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int local = tid + j * NT;
    if (local < width) {
      const float c = v[j];
      if (c != c) local_sum = NAN;
      else if (block_max != -INFINITY) local_sum += expf(c * inv_t - block_max);
    }
  }
  const float block_sum = block_sum_ref_order(local_sum, s_warp_sums);
  if (tid == 0) {
    a.block_maxes[row * a.nblocks + chunk] = block_max;
    a.block_sums[row * a.nblocks + chunk] = block_sum;
  }
}

struct Stage2Args {
  const float *block_values;
  const uint32_t *block_indices;
  const float *block_maxes, *block_sums;
  float *packed_out;      // [2k + 2] per row, or NULL
  float *values_out;      // unpacked variant
  uint32_t *indices_out;
  float *softmax_info_out;
  int nblocks, k, depth;
};

template <bool BATCHED>
__global__ void __launch_bounds__(NT) topk_stage2_kernel(Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *heads = (int *)smem;  // [nblocks]
  __shared__ float s_warp_max[32], s_warp_sums[32];
  __shared__ float s_global_max;
  const int tid = threadIdx.x, lane = tid & 63, k = a.k, nb = a.nblocks;
  const size_t row = BATCHED ? blockIdx.x : 0;
  const float *bv = a.block_values + row * (size_t)nb * k;
  const uint32_t *bi = a.block_indices + row * (size_t)nb * k;
  const float *bm = a.block_maxes + row * (size_t)nb, *bs = a.block_sums + row * (size_t)nb;
  float *packed = a.packed_out ? a.packed_out + row * (size_t)(2 * k + 2) : nullptr;
  // global max (a max is association-free)
  float gm = -INFINITY;
  for (int b = tid; b < nb; b += NT) gm = fmaxf(gm, bm[b]);
  gm = wave_max(gm);
  if (lane == 0) s_warp_max[tid >> 6] = gm;
  __syncthreads();
  if (tid == 0) s_global_max = fmaxf(fmaxf(s_warp_max[0], s_warp_max[1]), fmaxf(s_warp_max[2], s_warp_max[3]));
  __syncthreads();
  const float global_max = s_global_max;
  float local_denom = 0.0f;
  if (global_max != -INFINITY)
    for (int b = tid; b < nb; b += NT) local_denom += bs[b] * expf(bm[b] - global_max);
  const float denom = block_sum_ref_order(local_denom, s_warp_sums);
  if (tid == 0) {
    if (packed) { packed[2 * k] = denom; packed[2 * k + 1] = global_max; }
    else { a.softmax_info_out[0] = denom; a.softmax_info_out[1] = global_max; }
  }
  // the heads of every list, as keys, in LDS: `depth` entries per list (all k of them for vocabularies up to ~130k tokens); deeper entries are read from memory
  unsigned long long *skeys = (unsigned long long *)(smem + (((size_t)nb * sizeof(int) + 7) & ~(size_t)7));
  const int depth = a.depth;
  for (int i = tid; i < nb * depth; i += NT) {
    const int l = i / depth, h = i - l * depth;
    skeys[i] = key_of(bv[(size_t)l * k + h], bi[(size_t)l * k + h]);
  }
  __syncthreads();
  if (tid >= 64) return;
  // winners are only RECORDED in the loop (list, position); values and indices are gathered afterwards by 64 lanes at once -- a load per round on the
  // critical path would cost more than the round itself
  __shared__ int s_win[MAX_K];
  head_merge(nb, k, heads, [&](int l, int h) { return h < depth ? skeys[l * depth + h] : key_of(bv[(size_t)l * k + h], bi[(size_t)l * k + h]); },
             [&](int ki, int wl, int wp) { s_win[ki] = wl >= 0 ? wl * k + wp : -1; });
  MRS_WAVE_SYNC();
  for (int ki = lane; ki < k; ki += 64) {
    const int pos = s_win[ki];
    const float val = pos >= 0 ? bv[pos] : -INFINITY;
    const uint32_t idx = pos >= 0 ? bi[pos] : 0u;
    if (packed) { packed[ki] = val; packed[k + ki] = (float)idx; }
    else { a.values_out[ki] = val; a.indices_out[ki] = idx; }
  }
}

static bool shape_ok(int ncols, int k, int chunk_size, int nblocks) {
  return ncols > 0 && k >= 1 && k <= MAX_K && chunk_size >= 1 && chunk_size <= NT * MAXV && nblocks >= 1 && (long long)nblocks * chunk_size >= ncols;
}
static void run(const float *input, const float *inv_temperatures, float inv_temperature, float *block_values, uint32_t *block_indices, float *block_maxes,
                float *block_sums, float *packed_out, float *values_out, uint32_t *indices_out, float *softmax_info_out, int nrows, int ncols, int k,
                int chunk_size, int nblocks, bool batched, int64_t stream) {
  if (!shape_ok(ncols, k, chunk_size, nblocks) || nrows < 1) return;  // the reference's host wrapper validates before it calls (ops.rs:699-731)
  hipStream_t s = (hipStream_t)stream;
  Stage1Args a1{input, block_values, block_indices, block_maxes, block_sums, inv_temperatures, inv_temperature, ncols, k, chunk_size, nblocks};
  const size_t heads_bytes = ((size_t)nblocks * sizeof(int) + 7) & ~(size_t)7;
  if (heads_bytes > 56 * 1024) return;  // > 14 k chunks (a 29 M-token vocabulary at the reference's chunk size): outside this kernel
  const int depth = (int)std::min<size_t>((size_t)k, (60 * 1024 - heads_bytes) / 8 / (size_t)nblocks);  // keys staged per list (60 KiB of LDS in all; 0 = read from memory)
  Stage2Args a2{block_values, block_indices, block_maxes, block_sums, packed_out, values_out, indices_out, softmax_info_out, nblocks, k, depth};
  const size_t lds2 = heads_bytes + (size_t)nblocks * depth * 8;
  if (batched) {
    hipLaunchKernelGGL(topk_stage1_kernel<true>, dim3(nblocks, nrows), dim3(NT), 0, s, a1);
    hipLaunchKernelGGL(topk_stage2_kernel<true>, dim3(nrows), dim3(NT), lds2, s, a2);
  } else {
    hipLaunchKernelGGL(topk_stage1_kernel<false>, dim3(nblocks), dim3(NT), 0, s, a1);
    hipLaunchKernelGGL(topk_stage2_kernel<false>, dim3(1), dim3(NT), lds2, s, a2);
  }
}

// ---------------------------------------------------------------- greedy: top1_large_f32_packed[_batched] (sort.cu:1825-1912, 2071-2143, 2207-2238)
// stage 1: per chunk the largest logit and its (lowest) index; a chunk that holds a NaN reports (NaN, 0); no finite or +inf value -> (-inf, 0).
// stage 2: per row the best chunk (lowest position on ties); any NaN chunk -> token id UINT32_MAX and packed (NaN, NaN); nothing selectable -> token 0.
struct Top1Args {
  const float *input;
  float *block_values;
  uint32_t *block_indices;
  float *packed_out;
  uint32_t *token_ids_out;
  int ncols, chunk_size, nblocks;
};
// block-wide (key, value) arg-max + NaN flag: returns in thread 0
__device__ __forceinline__ void block_argmax(unsigned long long best, float val, bool nan, unsigned long long &okey, float &oval, bool &onan) {
  __shared__ unsigned long long s_k[4];
  __shared__ float s_v[4];
  __shared__ int s_n[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long w = wave_max_u64(best);
  const bool any_nan = __ballot(nan) != 0ull;
  if (w != 0ull && best == w) { s_k[wave] = w; s_v[wave] = val; }  // keys carry the position: one lane
  if (w == 0ull && lane == 0) { s_k[wave] = 0ull; s_v[wave] = -INFINITY; }
  if (lane == 0) s_n[wave] = any_nan;
  __syncthreads();
  okey = 0ull; oval = -INFINITY; onan = false;
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) {
      if (s_k[i] > okey) { okey = s_k[i]; oval = s_v[i]; }
      onan = onan || s_n[i] != 0;
    }
  }
  __syncthreads();
}
template <bool BATCHED>
__global__ void __launch_bounds__(NT) top1_stage1_kernel(Top1Args a) {
  const int tid = threadIdx.x, chunk = blockIdx.x;
  const size_t row = BATCHED ? blockIdx.y : 0;
  const float *input = a.input + row * (size_t)a.ncols;
  const int start = chunk * a.chunk_size, end = min(start + a.chunk_size, a.ncols);
  unsigned long long best = 0ull;
  float val = -INFINITY;
  bool nan = false;
  for (int idx = start + tid; idx < end; idx += NT) {
    const float c = input[idx];
    if (c != c) nan = true;
    const unsigned long long key = key_of(c, (unsigned)idx);
    if (key > best) { best = key; val = c; }
  }
  unsigned long long k; float v; bool n;
  block_argmax(best, val, nan, k, v, n);
  if (tid == 0) {
    a.block_values[row * a.nblocks + chunk] = n ? NAN : v;
    a.block_indices[row * a.nblocks + chunk] = n || k == 0ull ? 0u : idx_of(k);
  }
}
template <bool BATCHED>
__global__ void __launch_bounds__(NT) top1_stage2_kernel(Top1Args a) {
  const int tid = threadIdx.x;
  const size_t row = BATCHED ? blockIdx.x : 0;
  const float *bv = a.block_values + row * (size_t)a.nblocks;
  const uint32_t *bi = a.block_indices + row * (size_t)a.nblocks;
  unsigned long long best = 0ull;
  float val = -INFINITY;
  bool nan = false;
  for (int pos = tid; pos < a.nblocks; pos += NT) {
    const float c = bv[pos];
    if (c != c) nan = true;
    const unsigned long long key = key_of(c, (unsigned)pos);
    if (key > best) { best = key; val = c; }
  }
  unsigned long long k; float v; bool n;
  block_argmax(best, val, nan, k, v, n);
  if (tid == 0) {
    const uint32_t token = n ? 0xffffffffu : (k != 0ull ? bi[idx_of(k)] : 0u);
    if (a.packed_out) { a.packed_out[row * 2] = n ? NAN : v; a.packed_out[row * 2 + 1] = n ? NAN : (float)token; }
    if (a.token_ids_out) a.token_ids_out[row] = token;
  }
}
static void run_top1(const float *input, float *block_values, uint32_t *block_indices, float *packed_out, uint32_t *token_ids_out, int nrows, int ncols, int chunk_size,
                     int nblocks, bool batched, int64_t stream) {
  if (ncols <= 0 || chunk_size <= 0 || nblocks <= 0 || nrows <= 0 || (long long)nblocks * chunk_size < ncols) return;
  hipStream_t s = (hipStream_t)stream;
  Top1Args a{input, block_values, block_indices, packed_out, token_ids_out, ncols, chunk_size, nblocks};
  if (batched) {
    hipLaunchKernelGGL(top1_stage1_kernel<true>, dim3(nblocks, nrows), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(top1_stage2_kernel<true>, dim3(nrows), dim3(NT), 0, s, a);
  } else {
    hipLaunchKernelGGL(top1_stage1_kernel<false>, dim3(nblocks), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(top1_stage2_kernel<false>, dim3(1), dim3(NT), 0, s, a);
  }
}

// ---------------------------------------------------------------- logits pre-processing of the sampler (sort.cu:8-110; callers sampler.rs:1113-1169)
// dst = x, then the listed tokens are updated in place: penalties (frequency / presence / repetition, counts from the context) or additive biases.
__global__ void __launch_bounds__(NT) copy_f32_kernel(const float *__restrict__ x, float *__restrict__ dst, int n) {
  const int i = (blockIdx.x * NT + threadIdx.x) * 4;
  if (i + 3 < n && ((((uintptr_t)x | (uintptr_t)dst) & 15) == 0)) *(float4 *)(dst + i) = *(const float4 *)(x + i);
  else
    for (int j = i; j < n && j < i + 4; ++j) dst[j] = x[j];
}
__global__ void __launch_bounds__(NT) sparse_penalties_kernel(float *__restrict__ logits, const uint32_t *__restrict__ token_ids, const float *__restrict__ counts, int n,
                                                              int n_tokens, float frequency_penalty, float presence_penalty, float repetition_penalty) {
  const int idx = blockIdx.x * NT + threadIdx.x;
  if (idx >= n_tokens) return;
  const uint32_t token_id = token_ids[idx];
  if (token_id >= (uint32_t)n) return;
  const float count = counts[idx];
  if (count <= 0.0f) return;
  float value = logits[token_id];
  value -= fmaf(count, frequency_penalty, presence_penalty);  // `count * f + p` as nvcc contracts it (one rounding); this build has -ffp-contract=off, hence explicit
  if (repetition_penalty != 1.0f) value = value > 0.0f ? value / repetition_penalty : value * repetition_penalty;
  logits[token_id] = value;
}
__global__ void __launch_bounds__(NT) sparse_bias_kernel(float *__restrict__ logits, const uint32_t *__restrict__ token_ids, const float *__restrict__ biases, int n,
                                                         int n_tokens) {
  const int idx = blockIdx.x * NT + threadIdx.x;
  if (idx >= n_tokens) return;
  const uint32_t token_id = token_ids[idx];
  if (token_id >= (uint32_t)n) return;
  logits[token_id] += biases[idx];  // token ids are unique in the caller's map (sampler.rs:1145-1160), as in the reference
}

}  // namespace sampling
}  // namespace mrs

extern "C" void topk_large_f32(const float *input, float *block_values, uint32_t *block_indices, float *block_maxes, float *block_sums, float *values_out,
                               uint32_t *indices_out, float *softmax_info_out, int ncols, int k, int chunk_size, int nblocks, float inv_temperature,
                               int64_t stream) {
  mrs::sampling::run(input, nullptr, inv_temperature, block_values, block_indices, block_maxes, block_sums, nullptr, values_out, indices_out, softmax_info_out, 1,
                     ncols, k, chunk_size, nblocks, false, stream);
}
extern "C" void topk_large_f32_packed(const float *input, float *block_values, uint32_t *block_indices, float *block_maxes, float *block_sums, float *packed_out,
                                      int ncols, int k, int chunk_size, int nblocks, float inv_temperature, int64_t stream) {
  mrs::sampling::run(input, nullptr, inv_temperature, block_values, block_indices, block_maxes, block_sums, packed_out, nullptr, nullptr, nullptr, 1, ncols, k,
                     chunk_size, nblocks, false, stream);
}
extern "C" void topk_large_f32_packed_batched(const float *input, const float *inv_temperatures, float *block_values, uint32_t *block_indices, float *block_maxes,
                                              float *block_sums, float *packed_out, int nrows, int ncols, int k, int chunk_size, int nblocks, int64_t stream) {
  mrs::sampling::run(input, inv_temperatures, 0.0f, block_values, block_indices, block_maxes, block_sums, packed_out, nullptr, nullptr, nullptr, nrows, ncols, k,
                     chunk_size, nblocks, true, stream);
}
extern "C" void top1_large_f32_packed(const float *input, float *block_values, uint32_t *block_indices, float *packed_out, uint32_t *token_ids_out, int ncols,
                                      int chunk_size, int nblocks, int64_t stream) {
  mrs::sampling::run_top1(input, block_values, block_indices, packed_out, token_ids_out, 1, ncols, chunk_size, nblocks, false, stream);
}
extern "C" void top1_large_f32_packed_batched(const float *input, float *block_values, uint32_t *block_indices, float *packed_out, uint32_t *token_ids_out, int nrows,
                                              int ncols, int chunk_size, int nblocks, int64_t stream) {
  mrs::sampling::run_top1(input, block_values, block_indices, packed_out, token_ids_out, nrows, ncols, chunk_size, nblocks, true, stream);
}
extern "C" void apply_sparse_penalties_f32(const void *x, void *dst, const uint32_t *token_ids, const float *counts, const int n, const int n_tokens,
                                           const float frequency_penalty, const float presence_penalty, const float repetition_penalty, int64_t stream) {
  using namespace mrs::sampling;
  if (n <= 0) return;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((n + NT * 4 - 1) / (NT * 4)), dim3(NT), 0, s, (const float *)x, (float *)dst, n);
  if (n_tokens <= 0) return;
  hipLaunchKernelGGL(sparse_penalties_kernel, dim3((n_tokens + NT - 1) / NT), dim3(NT), 0, s, (float *)dst, token_ids, counts, n, n_tokens, frequency_penalty,
                     presence_penalty, repetition_penalty);
}
extern "C" void apply_sparse_logits_bias_f32(const void *x, void *dst, const uint32_t *token_ids, const float *biases, const int n, const int n_tokens, int64_t stream) {
  using namespace mrs::sampling;
  if (n <= 0) return;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((n + NT * 4 - 1) / (NT * 4)), dim3(NT), 0, s, (const float *)x, (float *)dst, n);
  if (n_tokens <= 0) return;
  hipLaunchKernelGGL(sparse_bias_kernel, dim3((n_tokens + NT - 1) / NT), dim3(NT), 0, s, (float *)dst, token_ids, biases, n, n_tokens);
}

