// ext_dec_gemv.hip -- instantiations of the decode engine's GEMV phase kernel (dec_gemv.cuh) for MRS_DEC_NC activation columns; build.py compiles this file
// once per column count 1 .. 8.  The SPEC schedule (dec_core2.cuh) exists for one column only: batched launches use ALL.
#include "dec_gemv.cuh"
#ifndef MRS_DEC_NC
#error "compile with -DMRS_DEC_NC=<1..8>"
#endif
namespace mrs {
namespace dec {
template <int EPI, bool SPEC, int NC, int TMASK = TM_ALL> static void go1(int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  auto kern = dec_gemv_kernel<NC, EPI, SPEC, TMASK>;
  lds_attr_once((const void *)kern, 158 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
}
template <> int gemv_launch<MRS_DEC_NC>(int epi, bool spec, int tmask, int grid, size_t lds, const GemvArgs &a, hipStream_t s) {
  constexpr int NC = MRS_DEC_NC;
  if constexpr (NC == 1) {
    if (epi == EPI_QKV) {  // per format set (ALL schedule: the launch is small)
      switch (tmask) {
      case TM_Q4K: go1<EPI_QKV, false, 1, TM_Q4K>(grid, lds, a, s); return 0;
      case TM_Q4K | TM_Q6K: go1<EPI_QKV, false, 1, TM_Q4K | TM_Q6K>(grid, lds, a, s); return 0;
      case TM_Q6K: go1<EPI_QKV, false, 1, TM_Q6K>(grid, lds, a, s); return 0;
      case TM_Q80: go1<EPI_QKV, false, 1, TM_Q80>(grid, lds, a, s); return 0;
      case TM_Q5K: go1<EPI_QKV, false, 1, TM_Q5K>(grid, lds, a, s); return 0;
      case TM_Q5K | TM_Q6K: go1<EPI_QKV, false, 1, TM_Q5K | TM_Q6K>(grid, lds, a, s); return 0;
      default: go1<EPI_QKV, false, 1>(grid, lds, a, s); return 0;
      }
    }
    if (spec) {
      switch (epi) {
      case EPI_STORE: go1<EPI_STORE, true, 1>(grid, lds, a, s); return 0;
      case EPI_RESID: go1<EPI_RESID, true, 1>(grid, lds, a, s); return 0;
      case EPI_GLU: go1<EPI_GLU, true, 1>(grid, lds, a, s); return 0;
      case EPI_RESID2: go1<EPI_RESID2, true, 1>(grid, lds, a, s); return 0;
      default: return -1;
      }
    }
  }
  switch (epi) {
  case EPI_STORE: go1<EPI_STORE, false, NC>(grid, lds, a, s); return 0;
  case EPI_RESID: go1<EPI_RESID, false, NC>(grid, lds, a, s); return 0;
  case EPI_GLU: go1<EPI_GLU, false, NC>(grid, lds, a, s); return 0;
  case EPI_QKV: go1<EPI_QKV, false, NC>(grid, lds, a, s); return 0;
  case EPI_RESID2: if constexpr (NC == 1) { go1<EPI_RESID2, false, 1>(grid, lds, a, s); return 0; } return -1;
  default: return -1;
  }
}
}  // namespace dec
}  // namespace mrs
