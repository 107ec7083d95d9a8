#!/bin/sh
# oracle/build_ref.sh <reference root> -- TEST INFRASTRUCTURE ONLY.
# Compiles the reference's OWN device functions for the host, from the sources where they lie under <reference root>:
#   _ref/libref_mmvq.so   <- head of mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu (block structs + vec_dot_*_q8_1,
#                            up to the "Core mat-vec-q template" marker) + ref_shim/mmvq_driver.inc
#   _ref/libref_affine.so <- head of kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu (block structs, get_quant,
#                            get_affine_params: the in-tree GGUF format spec) + ref_shim/affine_driver.inc
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
echo "oracle/_ref: built libref_mmvq.so libref_affine.so libref_hqq.so from $REF"
