// ext_dec_gemv.hip -- instantiations of the decode engine's GEMV phase kernel (dec_gemv.cuh) for MRS_DEC_NC activation columns; build.py compiles this file
// once per column count 1 .. 8.
#include "dec_gemv.cuh"
#ifndef MRS_DEC_NC
#error "compile with -DMRS_DEC_NC=<1..8>"
#endif
namespace mrs {
namespace dec {
template <int EPI, int NC, int TMASK = TM_ALL, bool RING2 = false> static void go1(int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  auto kern = dec_gemv_kernel<NC, EPI, TMASK, RING2>;
  lds_attr_once((const void *)kern, 158 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
}
// one or two format bodies per instantiation where the launch is latency-bound (batch 1); every format in one kernel for the batched launches
// (batch 1 also per ring depth: dec_core2.cuh stream() RING2)
template <int EPI, int NC, int TMASK> static void go_ring(bool ring2, int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  if (ring2) go1<EPI, NC, TMASK, true>(grid, lds, a, s); else go1<EPI, NC, TMASK, false>(grid, lds, a, s);
}
template <int EPI, int NC> static int go_masked(int tmask, bool ring2, int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  if constexpr (NC == 1) {
    switch (tmask) {
    case TM_Q4K: go_ring<EPI, 1, TM_Q4K>(ring2, grid, lds, a, s); return 0;
    case TM_Q6K: go_ring<EPI, 1, TM_Q6K>(ring2, grid, lds, a, s); return 0;
    case TM_Q80: go_ring<EPI, 1, TM_Q80>(ring2, grid, lds, a, s); return 0;
    case TM_Q5K: go_ring<EPI, 1, TM_Q5K>(ring2, grid, lds, a, s); return 0;
    case TM_Q4K | TM_Q6K: if constexpr (EPI == EPI_QKV) { go_ring<EPI, 1, TM_Q4K | TM_Q6K>(ring2, grid, lds, a, s); return 0; } break;
    case TM_Q5K | TM_Q6K: if constexpr (EPI == EPI_QKV) { go_ring<EPI, 1, TM_Q5K | TM_Q6K>(ring2, grid, lds, a, s); return 0; } break;
    default: break;
    }
  }
  if constexpr (NC > 1 || EPI == EPI_QKV) { go1<EPI, NC>(grid, lds, a, s); return 0; }  // any mix of formats
  else return -4;  // a single-tensor launch has a single format
}
template <> int gemv_launch<MRS_DEC_NC>(int epi, int tmask, bool ring2, int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  constexpr int NC = MRS_DEC_NC;
  switch (epi) {
  case EPI_STORE: return go_masked<EPI_STORE, NC>(tmask, ring2, grid, lds, a, s);
  case EPI_RESID: return go_masked<EPI_RESID, NC>(tmask, ring2, grid, lds, a, s);
  case EPI_GLU: return go_masked<EPI_GLU, NC>(tmask, ring2, grid, lds, a, s);
  case EPI_QKV: return go_masked<EPI_QKV, NC>(tmask, ring2, grid, lds, a, s);
  case EPI_RESID2: if constexpr (NC == 1) return go_masked<EPI_RESID2, 1>(tmask, ring2, grid, lds, a, s); return -1;
  default: return -1;
  }
}
}  // namespace dec
}  // namespace mrs
