#!/bin/sh
# Round-2 experiment (run HERE, in the CPU container: the .so travels to the GPU box with the snapshot): a second build of libmrs_hip_ext.so whose
# decode GEMV kernels are capped at 128 VGPRs (MRS_DECODE_MIN_WAVES=4, csrc/ext_decode.hip) so that two 512-thread workgroups share a CU.
# Output: mistral.rs_amd/lib/libmrs_hip_ext_occ4.so, selected at run time with MRS_EXT_LIB=libmrs_hip_ext_occ4.so (mistral.rs_amd/_lib.py).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/mistral.rs_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -I$CSRC -I$ROOT/include"
mkdir -p "$CSRC/build"
/opt/rocm/bin/hipcc $FLAGS -DMRS_DECODE_MIN_WAVES=${1:-4} -c "$CSRC/ext_decode.hip" -o "$CSRC/build/ext_decode_occ4.o"
objs=""
for o in ext_gemm ext_attn_prefill ext_comm ext_isq runtime kv_cache_manager; do objs="$objs $CSRC/build/$o.o"; done
# same link line as mistral.rs_amd/build.py (ext links against the three ABI libraries next to it)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/mistral.rs_amd/lib/libmrs_hip_ext_occ4.so" "$CSRC/build/ext_decode_occ4.o" $objs \
  -L"$ROOT/mistral.rs_amd/lib" -lmistralrsquant -lmistralrspagedattention -lmistralrscuda '-Wl,-rpath,$ORIGIN' -ldl
echo "built mistral.rs_amd/lib/libmrs_hip_ext_occ4.so"
