// oracle/ref_shim/moe_shim.h -- TEST INFRASTRUCTURE ONLY.  Extra host stand-ins for the reference's MoE kernels
// (kernels/indexed_moe/indexed_moe.cu:1157-1615 fused decode pair, kernels/moe_grouped/moe_grouped.cu:629-1102 dispatch / weighted
// reduce / grouped GEMM): the fibers of a block and the blocks of a grid run one after the other, so an atomic is a plain
// read-modify-write -- the result is the one the device produces when its atomics happen to retire in launch order.
#pragma once
template <class T> static inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
static inline __half __float2half_rn(float f) { return __half(f); }
static inline __nv_bfloat16 __float2bfloat16_rn(float f) { return __nv_bfloat16(f); }
using std::max;
using std::min;
