#!/bin/sh
# oracle/build_ref.sh <reference root> -- TEST INFRASTRUCTURE ONLY.
# Compiles the reference's OWN device functions for the host, from the sources where they lie under <reference root>:
#   _ref/libref_mmvq.so   <- head of mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu (block structs + vec_dot_*_q8_1,
#                            up to the "Core mat-vec-q template" marker) + ref_shim/mmvq_driver.inc
#   _ref/libref_affine.so <- head of kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu (block structs, get_quant,
#                            get_affine_params: the in-tree GGUF format spec) + ref_shim/affine_driver.inc
#   _ref/libref_cache.so  <- kernels/rotary/rotary.cu (namespace vllm), reshape_and_cache_kernel.cu, gather_kv_cache_kernel.cu,
#                            copy_blocks_kernel.cu kernels (f32 instantiations), run by ref_shim/cache_driver.inc
#   _ref/libref_pa.so     <- paged_attention_v1 / v2 / v2_reduce kernels (pagedattention.cuh:56-667 + attention/*.cuh, f32 path) executed
#                            on host fibers with block barriers and warp shuffles (ref_shim/fiber_shim.h, pa_driver.inc)
#   _ref/libref_q8_1.so   <- mmvq_gguf_quantize_q8_1_f32 (mmvq_gguf.cu) on host fibers: the reference's Q8_1 activation quantizer
#   _ref/libref_rms.so    <- add_rms_norm_* / rms_norm_residual_* kernels (mistralrs-core/src/cuda/sort.cu:148-428) on host fibers
#   _ref/libref_mmvq_kernel.so <- the whole mmvq_gguf.cu kernel set (mmvq_core_impl, fused GLU; f32 destinations) on host fibers
#   _ref/libref_glu.so    <- fused_glu_kernel / fused_glu_kernel_vec4 (mistralrs-quant/kernels/ops/ops.cu) in f32 / f16 / bf16
#   _ref/libref_router.so <- moe_router_topk_kernel (mistralrs-core/src/cuda/sort.cu:1097-1357), f32 logits, all option combinations
#   _ref/libref_imoe.so   <- indexed_moe_forward_<t>_q8_1 kernels (kernels/indexed_moe/indexed_moe.cu) on host fibers
#   _ref/libref_moe_decode.so  <- moe_gemv_fused_gate_up_<t> / moe_gemv_down_aggregate_<t> (indexed_moe.cu:1157-1615) on host fibers
#   _ref/libref_moe_grouped.so <- moe_dispatch_* / moe_weighted_reduce_flat / moe_grouped_gemm_<t> kernels (kernels/moe_grouped/moe_grouped.cu) on host fibers
#   _ref/libref_gemv.so   <- gemv_kernel_batched (kernels/gemv/gemv.cu:50-160) in f32 / f16 / bf16, batch 1..8, on host fibers
#   _ref/libref_half.so   <- f16 / bf16 instantiations of the HQQ dequantize kernels and of the rotary kernels (arithmetic in the tensor dtype)
#   _ref/libref_hqq.so    <- the __global__ kernel templates of kernels/hqq/hqq.cu (dequantize_*) and hqq_bitpack.cu (pack_*),
#                            run one thread at a time by ref_shim/hqq_driver.inc
# The reference text is STREAMED into g++ (stdin); nothing from /root/reference is written into this repo.
# Outputs only into oracle/_ref/ (git-ignored, shipped to the GPU box by gpurun like the other built .so files).
set -e
REF="$1"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
mkdir -p "$OUT"
CXX="${CXX:-g++}"
FLAGS="-x c++ -std=c++17 -O2 -fPIC -shared -fno-fast-math -ffp-contract=off -w"
FIB="-DSHIM_FIBERS"
MMVQ="$REF/mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu"
AFF_DIR="$REF/mistralrs-quant/kernels/gguf_affine_packed"
( cat "$HERE/ref_shim/cuda_shim.h"
  awk '/Core mat-vec-q template/{exit} {print}' "$MMVQ" | grep -v '#include "cuda_'
  cat "$HERE/ref_shim/mmvq_driver.inc" ) | $CXX $FLAGS -o "$OUT/libref_mmvq.so" -
( cat "$HERE/ref_shim/cuda_shim.h"
  awk '/^template <typename T> struct Scalar;/{exit} {print}' "$AFF_DIR/marlin_gguf_affine_repack.cu" | grep -v '#include <cuda'
  cat "$HERE/ref_shim/affine_driver.inc" ) | $CXX $FLAGS -I"$AFF_DIR" -o "$OUT/libref_affine.so" -
# HQQ: only the live kernel templates (a line starting with "__global__ void", plus its "template <typename T>" line, up to the
# closing brace in column 0) -- the launchers use <<<>>> and are not needed
HQQ_DIR="$REF/mistralrs-quant/kernels/hqq"
KERNELS='/^template <typename T>$/{t=$0; next} /^__global__ void /{if (t != "") print t; p=1} {t=""} p{print} p && /^}$/{p=0}'
( cat "$HERE/ref_shim/cuda_shim.h"
  awk "$KERNELS" "$HQQ_DIR/hqq.cu"
  awk "$KERNELS" "$HQQ_DIR/hqq_bitpack.cu"
  cat "$HERE/ref_shim/hqq_driver.inc" ) | $CXX $FLAGS -o "$OUT/libref_hqq.so" -
# RoPE + paged-cache data movement: the rotary namespace block as a whole (no launch syntax inside), the three cache kernels by name
PA_DIR="$REF/mistralrs-paged-attn/src/cuda"
ONE='$0 ~ start {p=1; if (t != "") print t} /^template </{t=$0; if (!p) next} {if (!p) t=""} p{print} p && /^}$/{exit}'
( cat "$HERE/ref_shim/cuda_shim.h"
  echo '#define VLLM_LDG(arg) *(arg)'
  awk '/^namespace vllm \{/{p=1} p{print} /^} \/\/ namespace vllm/{exit}' "$REF/mistralrs-quant/kernels/rotary/rotary.cu"
  echo 'namespace vllm { enum class Fp8KVCacheDataType { kAuto, kFp8E4M3, kFp8E5M2 };'
  echo 'namespace fp8 { template <class O, class I, Fp8KVCacheDataType K> static O scaled_convert(const I &, float) { return O{}; } }'
  awk -v start='^__global__ void reshape_and_cache_kernel' "$ONE" "$PA_DIR/reshape_and_cache_kernel.cu"
  awk -v start='^__global__ void gather_kv_cache_kernel' "$ONE" "$PA_DIR/gather_kv_cache_kernel.cu"
  echo '}'
  awk -v start='^copy_blocks_internal_kernel' '/^template </{t=$0; next} /^__device__ void$/{d=$0; next} $0 ~ start {p=1; print t; print d} p{print} p && /^}$/{exit}' "$PA_DIR/copy_blocks_kernel.cu"
  cat "$HERE/ref_shim/cache_driver.inc" ) | $CXX $FLAGS -o "$OUT/libref_cache.so" -
# paged attention v1 / v2 / v2_reduce: the f32 path of the reference kernels on host fibers (ref_shim/fiber_shim.h).  Streamed: the
# generic vector templates, the float32 dtype header, the fp8 enum header, the Q.K helper and pagedattention.cuh:56-667 (namespace vllm
# up to the launch macros); includes / pragmas dropped, the dynamic shared-memory declaration pointed at the shim's buffer.
ATT="$PA_DIR/attention"
STRIP='/^#include/d; /^#pragma once/d'
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  sed "$STRIP" "$ATT/attention_generic.cuh"; echo ''
  sed "$STRIP" "$ATT/dtype_float32.cuh"; echo ''
  sed "$STRIP" "$ATT/dtype_fp8.cuh"
  echo ''  # the header has no trailing newline
  echo 'namespace vllm { namespace fp8 { template <class O, class I, Fp8KVCacheDataType K> static O scaled_convert(const I &, float) { return O{}; } } }'
  sed "$STRIP" "$ATT/attention_utils.cuh"; echo ''
  sed -n '56,667p' "$PA_DIR/pagedattention.cuh" | sed 's/extern __shared__ char shared_mem\[\];/char *shared_mem = shim_fiber::dyn_smem;/'
  cat "$HERE/ref_shim/pa_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_pa.so" -
# Q8_1 activation quantizer: the reference's mmvq_gguf_quantize_q8_1_f32 kernel (mmvq_gguf.cu, after the fused-QKV type sets) with its
# real warp reductions (butterfly order), on top of the same streamed head as libref_mmvq
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  awk '/Core mat-vec-q template/{exit} {print}' "$MMVQ" | grep -v '#include "cuda_' | grep -v '^#define WARP_SIZE'
  awk '/^mmvq_gguf_quantize_q8_1_f32\(/{p=1; print "extern \"C\" void"} p{print} p && /^}$/{exit}' "$MMVQ"
  cat "$HERE/ref_shim/quantize_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_q8_1.so" -
# the complete MMVQ kernels (mmvq_core_impl + fused GLU, all 10 formats x batch 1..8): mmvq_gguf.cu up to its host-side launchers
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  awk '/^\/\/ Host-side launchers/{exit} {print}' "$MMVQ" | grep -v '#include "cuda_' | grep -v '^#define WARP_SIZE'
  cat "$HERE/ref_shim/mmvq_kernel_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_mmvq_kernel.so" -
# indexed MoE forward (mistralrs-quant/kernels/indexed_moe/indexed_moe.cu): everything except the <<<>>> launchers
IMOE="$REF/mistralrs-quant/kernels/indexed_moe/indexed_moe.cu"
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  awk '/^\/\/ Launch wrapper for BF16 quantize/{p=0} /^\/\/ indexed_moe_forward template/{p=1} /^\/\/ =+ C wrapper functions/{exit} BEGIN{p=1} p{print}' "$IMOE" \
    | grep -v '#include "cuda_' | grep -v '^#define WARP_SIZE'
  cat "$HERE/ref_shim/imoe_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_imoe.so" -
# fused MoE decode pair (indexed_moe.cu:1157-1615): the file's head (block structs, vec_dot_*, quantize kernels) + the fused section up to its launchers
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h" "$HERE/ref_shim/moe_shim.h"
  awk '/^\/\/ Launch wrapper for BF16 quantize/{p=0} /^\/\/ =+ Fused MoE decode kernels/{p=1} /^\/\/ =+ Fused gate\+up launcher functions/{exit} BEGIN{p=1} p{print}' "$IMOE" \
    | grep -v '#include "cuda_' | grep -v '^#define WARP_SIZE'
  cat "$HERE/ref_shim/moe_decode_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_moe_decode.so" -
# MoE prompt path (moe_grouped.cu:1-1102): dispatch kernels, weighted reduce, tiled grouped GEMM; launchers restated in the driver
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h" "$HERE/ref_shim/moe_shim.h"
  awk '/^\/\/ =+ C wrapper functions for FFI/{exit} {print}' "$REF/mistralrs-quant/kernels/moe_grouped/moe_grouped.cu" \
    | grep -v '#include "cuda_' | grep -v '^#define WARP_SIZE' \
    | sed 's/extern __shared__ float weights\[\];/float *weights = (float *)shim_fiber::dyn_smem;/; s/extern __shared__ char smem\[\];/char *smem = shim_fiber::dyn_smem;/; s/extern __shared__ char smem_q8\[\];/char *smem_q8 = shim_fiber::dyn_smem;/'
  cat "$HERE/ref_shim/moe_grouped_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_moe_grouped.so" -
# dense decode GEMV (kernels/gemv/gemv.cu): reduction helper, converters and the batched kernel template, up to the block-size selection
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  echo '#define __CUDA_ARCH__ 800'
  echo '#define __ldg(p) (*(p))'
  echo 'struct __nv_bfloat162 { __nv_bfloat16 x, y; };'
  awk '/^\/\/ Warp-level reduction sum/{p=1} /^\/\/ Block Size Selection/{exit} p{if (held != "") print held; held=$0}' "$REF/mistralrs-quant/kernels/gemv/gemv.cu"
  cat "$HERE/ref_shim/gemv_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_gemv.so" -
# 16-bit instantiations whose arithmetic runs in the tensor dtype (every operation rounds to half / bf16): HQQ dequantize and RoPE, with
# the shim's half / bf16 operator set (-DSHIM_HALF_OPS)
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  echo '#define VLLM_LDG_DONE'
  awk "$KERNELS" "$HQQ_DIR/hqq.cu"
  awk '/^namespace vllm \{/{p=1} p{print} /^} \/\/ namespace vllm/{exit}' "$REF/mistralrs-quant/kernels/rotary/rotary.cu"
  cat "$HERE/ref_shim/half_driver.inc" ) | $CXX $FLAGS $FIB -DSHIM_HALF_OPS -o "$OUT/libref_half.so" -
# MoE router (mistralrs-core/src/cuda/sort.cu): constants, warp reductions and moe_router_topk_kernel up to its launcher
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  echo 'static inline float max(float a, float b) { return fmaxf(a, b); }'
  awk '/^constexpr int MOE_ROUTER_SCORE_RAW/{p=1} /^void launch_moe_router_topk/{exit} p{if (held != "") print held; held=$0}' "$REF/mistralrs-core/src/cuda/sort.cu"
  cat "$HERE/ref_shim/router_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_router.so" -
# sampling top-k over a large vocabulary (mistralrs-core/src/cuda/sort.cu): warp_reduce_max_with_idx (:925-940), the sum reductions (:1470-1497),
# topk_large_stage1_f32 (:1502-1600) and topk_large_stage2_f32_packed (:1708-1823); launch sequence of :2165-2206 in topk_driver.inc
grep -q '^__global__ void topk_large_stage1_f32($' "$REF/mistralrs-core/src/cuda/sort.cu" || { echo "build_ref.sh: topk_large_stage1_f32 moved"; exit 1; }
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  S="$REF/mistralrs-core/src/cuda/sort.cu"
  echo 'static inline int min(int a, int b) { return a < b ? a : b; }'
  echo 'static inline int max(int a, int b) { return a > b ? a : b; }'
  awk '/^__device__ __forceinline__ T warp_reduce_max_with_idx/{print "template <typename T>"; p=1} p{print} p&&/^}/{exit}' "$S"
  awk '/^__device__ __forceinline__ float warp_reduce_sum_f32/{p=1} /^\/\/ Large-vocabulary top-k for token sampling/{exit} p{print}' "$S"
  awk '/^__global__ void topk_large_stage1_f32\($/{print "template <bool BATCHED>"; p=1} /^__global__ void topk_large_stage2_f32\($/{exit} p{print}' "$S" | sed 's/extern __shared__ char smem\[\];/char *smem = shim_fiber::dyn_smem;/'
  awk '/^__global__ void topk_large_stage2_f32_packed\($/{print "template <bool BATCHED>"; p=1} /^template <bool BATCHED, bool COMPUTE_SUMS>/{exit} p{print}' "$S" | sed 's/extern __shared__ char smem\[\];/char *smem = shim_fiber::dyn_smem;/'
  # greedy: top1_large_stage1_f32 (:1825-1912) and top1_large_stage2_f32_packed (:2071-2143); __syncthreads_or is three block barriers around a flag
  echo 'static int shim_or_flag; static inline int __syncthreads_or(int p) { __syncthreads(); if (p) shim_or_flag = 1; __syncthreads(); const int r = shim_or_flag; __syncthreads(); if (threadIdx.x == 0) shim_or_flag = 0; return r; }'
  awk '/^__global__ void top1_large_stage1_f32\($/{print "template <bool BATCHED, bool COMPUTE_SUMS>"; p=1} /^__global__ void categorical_large_stage2_f32_packed\($/{exit} p{print}' "$S"
  awk '/^top1_large_stage2_f32_packed\(/{print "template <bool BATCHED>"; print "__global__ void"; p=1} /^extern "C" void topk_large_f32\(/{exit} p{print}' "$S"
  cat "$HERE/ref_shim/topk_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_topk.so" -
# fused_glu (mistralrs-quant/kernels/ops/ops.cu): activation enum + functions, scalar and vec4 kernels; f32 / f16 / bf16
OPS="$REF/mistralrs-quant/kernels/ops/ops.cu"
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  awk '/^enum GluActivation/{p=1} /^void launch_fused_glu/{exit} p{if (held != "") print held; held=$0}' "$OPS"
  cat "$HERE/ref_shim/glu_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_glu.so" -
# add_rms_norm / rms_norm_residual (mistralrs-core/src/cuda/sort.cu): helpers + block reduction (:148-243), the residual kernels (:244-318),
# the add kernels (:351-428); f32, f16 and bf16 instantiations through the shim's half / bf16 stand-ins
SORT="$REF/mistralrs-core/src/cuda/sort.cu"
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  sed -n '148,318p' "$SORT" | awk '/^template <typename T>$/{t=$0; next} /^void launch_/{skip=1} !skip{if (t != "") print t; print} {t=""} skip && /^}$/{skip=0}'
  sed -n '351,428p' "$SORT"
  cat "$HERE/ref_shim/rms_driver.inc" ) | $CXX $FLAGS $FIB -o "$OUT/libref_rms.so" -
# MMQ activation quantizer (mistralrs-quant/kernels/mmq_gguf/mmq_quantize.cu): block_q8_1_mmq + layout enum streamed from mmq_gguf.cuh, the two
# kernels (plain, fused GLU) streamed from mmq_quantize.cu up to its launchers; f32 / f16 / bf16 inputs, D4 / DS4 / D2S6 layouts
MMQ_DIR="$REF/mistralrs-quant/kernels/mmq_gguf"
( cat "$HERE/ref_shim/cuda_shim.h" "$HERE/ref_shim/fiber_shim.h"
  echo '#define QK8_1 32'
  echo '#define MATRIX_ROW_PADDING 512'
  echo 'struct char4 { signed char x, y, z, w; };'
  echo 'struct __nv_bfloat162 { __nv_bfloat16 x, y; };'
  echo 'static inline half2 make_half2(float a, float b) { half2 r; r.x = __half(a); r.y = __half(b); return r; }'
  echo 'static inline float2 __bfloat1622float2(__nv_bfloat162 v) { return float2{(float)v.x, (float)v.y}; }'
  awk '/^enum mmq_q8_1_ds_layout/{p=1} /^struct block_fp4_mmq/{exit} p{print}' "$MMQ_DIR/mmq_gguf.cuh"
  awk '/^#define CUDA_QUANTIZE_BLOCK_SIZE_MMQ/{p=1} /^template <mmq_q8_1_ds_layout ds_layout>$/{exit} p{print}' "$MMQ_DIR/mmq_quantize.cu"
  cat "$HERE/ref_shim/mmq_quantize_driver.inc" ) | $CXX $FLAGS $FIB -DSHIM_HALF_OPS -o "$OUT/libref_mmq_quantize.so" -
echo "oracle/_ref: built libref_mmvq.so libref_affine.so libref_hqq.so libref_cache.so libref_pa.so libref_q8_1.so libref_rms.so libref_mmvq_kernel.so libref_glu.so libref_router.so libref_imoe.so libref_moe_decode.so libref_moe_grouped.so libref_gemv.so libref_half.so libref_mmq_quantize.so libref_topk.so from $REF"
