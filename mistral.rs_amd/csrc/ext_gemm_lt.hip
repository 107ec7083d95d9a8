// ext_gemm_lt.hip -- the SELECTABLE bf16 prompt path on a bf16 shadow copy of the weights (round 6).
//
// Role in the reference: the prompt GEMMs of GgufMatMul::forward / fast_mmq (mistralrs-quant/src/gguf/fast_mmq.rs:528-635).  north_star asks for "MFMA only on the bf16
// prefill GEMM"; rounds 2-5 fused the block dequant INTO that GEMM (ext_gemm.hip) and stayed VALU-bound at 0.21-0.31 of the bf16 MFMA peak (the K-quant decode runs on the
// vector ALU next to the matrix cores).  288 GB of HBM make the other design possible: dequantize every dense linear ONCE at load time to bf16 (mrs_dequantize, 2 bytes per
// weight: 14 GB for Llama-3-8B) and run the prompt GEMMs as PLAIN bf16 x bf16 -> f32 library GEMMs (hipBLASLt: "hipBLASLt/rocBLAS only for plain library GEMMs").  Same
// arithmetic as the fused kernel -- weights rounded to bf16 once, activations rounded to bf16, f32 accumulation -- in the library's summation order; measured on the
// MI355X (profiles/experiments/gemm_lib_probe.py): 0.335 of the peak per layer at 512 tokens, 0.455 at 2048, 0.547 at Llama-3-70B shapes, against 0.21 / 0.31 / 0.30.
// This path is NOT the default prompt path (that one keeps the reference CPU arithmetic: ext_gemm_qi.hip); Llama.set_prefill_mode(0) selects it, and it is used only when
// every dense linear of the model has its shadow copy (Llama(bf16_shadow=...): off for models whose copy would not fit).
#include "common.cuh"
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>

namespace mrs {
namespace lt {
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)float_to_bf16_bits(a) | ((unsigned)float_to_bf16_bits(b) << 16); }
// f32 rows -> bf16 rows (row-major, contiguous): the activations of the library GEMM
__global__ void __launch_bounds__(256) rows_bf16_kernel(const float *__restrict__ x, uint16_t *__restrict__ y, int ldx, int M, int K) {
  const int k8 = K / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * k8) return;
  const int m = (int)(i / k8), k = (int)(i % k8) * 8;
  const float4 a = *(const float4 *)(x + (size_t)m * ldx + k), b = *(const float4 *)(x + (size_t)m * ldx + k + 4);
  *(int4 *)(y + (size_t)m * K + k) = make_int4((int)pack2(a.x, a.y), (int)pack2(a.z, a.w), (int)pack2(b.x, b.y), (int)pack2(b.z, b.w));
}
// silu(g) * u -> bf16 rows (fused_glu's expression, mistralrs-quant/src/utils/ops.rs:2601-2620)
__global__ void __launch_bounds__(256) glu_rows_bf16_kernel(const float *__restrict__ g, const float *__restrict__ u, uint16_t *__restrict__ y, int ld, int M, int N) {
  const int n8 = N / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * n8) return;
  const int m = (int)(i / n8), k = (int)(i % n8) * 8;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float gv = g[(size_t)m * ld + k + j]; o[j] = gv / (1.0f + expf(-gv)) * u[(size_t)m * ld + k + j]; }
  *(int4 *)(y + (size_t)m * N + k) = make_int4((int)pack2(o[0], o[1]), (int)pack2(o[2], o[3]), (int)pack2(o[4], o[5]), (int)pack2(o[6], o[7]));
}
// RMSNorm (x * rsqrt(mean(x^2) + eps) * w: the arithmetic and reduction order of mrs_rms_norm_f32) -> bf16 rows; one workgroup per row
__global__ void __launch_bounds__(256) rms_norm_rows_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w, uint16_t *__restrict__ y, int M, int K, float eps) {
  __shared__ float red[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float *xr = x + (size_t)m * K;
  float sum = 0.f;
  for (int v = tid; v < K / 4; v += 256) {
    const float4 t = *(const float4 *)(xr + v * 4);
    sum = fmaf(t.x, t.x, sum); sum = fmaf(t.y, t.y, sum); sum = fmaf(t.z, t.z, sum); sum = fmaf(t.w, t.w, sum);
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
  for (int v = tid; v < K / 8; v += 256) {
    const int k = v * 8;
    const float4 a = *(const float4 *)(xr + k), b = *(const float4 *)(xr + k + 4), wa = *(const float4 *)(w + k), wb = *(const float4 *)(w + k + 4);
    *(int4 *)(y + (size_t)m * K + k) = make_int4((int)pack2(a.x * inv * wa.x, a.y * inv * wa.y), (int)pack2(a.z * inv * wa.z, a.w * inv * wa.w),
                                                 (int)pack2(b.x * inv * wb.x, b.y * inv * wb.y), (int)pack2(b.z * inv * wb.z, b.w * inv * wb.w));
  }
}

struct Plan { hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t a, b, c; hipblasLtMatmulAlgo_t algo; size_t ws; bool ok; };
struct State {
  hipblasLtHandle_t h = nullptr;
  void *ws = nullptr;
  size_t ws_bytes = 0;
  std::map<std::tuple<int, int, int, int>, Plan> plans;
  std::mutex mu;
  bool failed = false;
};
static State &state() { static State s; return s; }
}  // namespace lt
}  // namespace mrs

using namespace mrs;

extern "C" int mrs_rows_f32_to_bf16(const float *x, int ldx, int M, int K, void *y, void *stream) {
  if (K <= 0 || K % 8 || (ldx & 3)) return -1;
  if (M <= 0) return 0;
  const size_t n = (size_t)M * (K / 8);
  hipLaunchKernelGGL(lt::rows_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t *)y, ldx, M, K);
  return 0;
}
extern "C" int mrs_rows_glu_bf16(const float *g, const float *u, int ld, int M, int N, void *y, void *stream) {
  if (N <= 0 || N % 8) return -1;
  if (M <= 0) return 0;
  const size_t n = (size_t)M * (N / 8);
  hipLaunchKernelGGL(lt::glu_rows_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, u, (uint16_t *)y, ld, M, N);
  return 0;
}
extern "C" int mrs_rows_rms_norm_bf16(const float *x, const float *w, int M, int K, float eps, void *y, void *stream) {
  if (K <= 0 || K % 8) return -1;
  if (M <= 0) return 0;
  hipLaunchKernelGGL(lt::rms_norm_rows_bf16_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, w, (uint16_t *)y, M, K, eps);
  return 0;
}

// out f32 [T][ldo] (+)= x bf16 [T][K] . w bf16 [N][K]^T, f32 accumulation (hipBLASLt).  One-time set-up (handle, 64 MiB workspace, one plan per shape) happens at the
// first call: call it once outside any stream capture.  Returns 0; -1 bad arguments; -5 the library has no kernel for the shape / is unavailable.
extern "C" int mrs_lt_gemm_bf16(const void *w_bf16, const void *x_bf16, float *out, int ldo, int N, int K, int T, int accumulate, void *stream) {
  if (!w_bf16 || !x_bf16 || !out || N <= 0 || K <= 0 || T <= 0 || ldo < N) return -1;
  lt::State &st = lt::state();
  std::lock_guard<std::mutex> g(st.mu);
  if (st.failed) return -5;
  if (!st.h) {
    if (hipblasLtCreate(&st.h) != HIPBLAS_STATUS_SUCCESS) { st.failed = true; return -5; }
    st.ws_bytes = (size_t)64 << 20;
    if (hipMalloc(&st.ws, st.ws_bytes) != hipSuccess) { st.ws = nullptr; st.ws_bytes = 0; }
  }
  const auto key = std::make_tuple(N, K, T, ldo);
  auto it = st.plans.find(key);
  if (it == st.plans.end()) {
    lt::Plan p{};
    // column-major view: D [N x T] (ld = ldo) = op(A) [N x K] . B [K x T];  A = the weights, row-major [N][K] = column-major [K x N] (lda = K), transposed;
    // B = the activations, row-major [T][K] = column-major [K x T] (ldb = K)
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    bool ok = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16BF, K, N, K) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16BF, K, T, K) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && hipblasLtMatrixLayoutCreate(&p.c, HIP_R_32F, N, T, ldo) == HIPBLAS_STATUS_SUCCESS;
    if (ok) {
      hipblasLtMatmulPreference_t pref;
      ok = hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
      if (ok) {
        uint64_t wsb = st.ws_bytes;
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
        constexpr int NC = 8;
        hipblasLtMatmulHeuristicResult_t res[NC];
        int n = 0;
        ok = hipblasLtMatmulAlgoGetHeuristic(st.h, p.desc, p.a, p.b, p.c, p.c, pref, NC, res, &n) == HIPBLAS_STATUS_SUCCESS && n > 0;
        if (ok) {
          // the heuristic's first answer is not always the fastest kernel at prompt shapes (512 x 4096 x 4096: 54 us against 32): time the candidates once, on a
          // scratch output, and keep the best (plan time only: the first call per shape, never inside a stream capture)
          int best = 0;
          float *scratch = nullptr;
          hipEvent_t e0, e1;
          if (n > 1 && hipMalloc(&scratch, (size_t)T * ldo * 4) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            const float one = 1.0f, zerof = 0.0f;
            float best_ms = 1e30f;
            for (int c = 0; c < n; ++c) {
              if (res[c].state != HIPBLAS_STATUS_SUCCESS || res[c].workspaceSize > st.ws_bytes) continue;
              bool good = true;
              for (int rep = 0; rep < 4 && good; ++rep) {
                if (rep == 1) hipEventRecord(e0, (hipStream_t)stream);
                good = hipblasLtMatmul(st.h, p.desc, &one, w_bf16, p.a, x_bf16, p.b, &zerof, scratch, p.c, scratch, p.c, &res[c].algo, st.ws, res[c].workspaceSize,
                                       (hipStream_t)stream) == HIPBLAS_STATUS_SUCCESS;
              }
              hipEventRecord(e1, (hipStream_t)stream);
              hipEventSynchronize(e1);
              float ms = 1e30f;
              if (good) hipEventElapsedTime(&ms, e0, e1);
              if (good && ms < best_ms) { best_ms = ms; best = c; }
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
          }
          if (scratch) hipFree(scratch);
          p.algo = res[best].algo; p.ws = res[best].workspaceSize;
        }
        hipblasLtMatmulPreferenceDestroy(pref);
      }
    }
    p.ok = ok;
    it = st.plans.emplace(key, p).first;
  }
  const lt::Plan &p = it->second;
  if (!p.ok) return -5;
  const float alpha = 1.0f, beta = accumulate ? 1.0f : 0.0f;
  const hipblasStatus_t rc = hipblasLtMatmul(st.h, p.desc, &alpha, w_bf16, p.a, x_bf16, p.b, &beta, out, p.c, out, p.c, &p.algo, st.ws, p.ws <= st.ws_bytes ? p.ws : 0,
                                             (hipStream_t)stream);
  return rc == HIPBLAS_STATUS_SUCCESS ? 0 : -5;
}
