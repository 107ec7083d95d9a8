#!/bin/sh
# oracle/build_hip_host.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the PRODUCT kernel sources of libmistralrsquant.so (mistral.rs_amd/csrc: the MMVQ core with every launcher of the 10 GGUF types,
# the Q8_1 quantizer, moe.hip) for the HOST on top of oracle/hip_host/hip/hip_runtime.h (wave64 fibers), so the launchers can be executed
# and compared with the oracle without a GPU (tests/test_hip_host_emulation.py).  Output: oracle/_hiphost/libhiphost_quant.so (git-ignored).
# The sources are copied into oracle/_hiphost/src with ONE textual change: `extern __shared__ ... name[];` (dynamic LDS) becomes a pointer to
# the shim's LDS buffer.  Same flags that pin the arithmetic in the product build: -ffp-contract=off, no fast-math.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
CSRC="$HERE/../mistral.rs_amd/csrc"
OUT="$HERE/_hiphost"
CXX="${HIPHOST_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
mkdir -p "$OUT/src" "$OUT/obj"
for f in common.cuh gguf_blocks.cuh mmvq_core.cuh mmvq_kernels.cuh mmvq_inst.hip mmvq_quantize.hip moe.hip; do
  sed -E 's/extern __shared__( __attribute__\(\(aligned\(16\)\)\))? ([A-Za-z_0-9]+) ([A-Za-z_0-9]+)\[\];/\2 *\3 = (\2 *)hiphost::dyn_lds;/' "$CSRC/$f" > "$OUT/src/$f"
done
FLAGS="-x c++ -std=c++17 -O1 -fPIC -march=native -fno-fast-math -ffp-contract=off -w -I$HERE/hip_host -I$OUT/src -I$HERE/../include"
pids=""
for spec in q4_0:2:q4_0 q4_1:3:q4_1 q5_0:6:q5_0 q5_1:7:q5_1 q8_0:8:q8_0 q2_k:10:q2k q3_k:11:q3k q4_k:12:q4k q5_k:13:q5k q6_k:14:q6k; do
  tag=${spec%%:*}; rest=${spec#*:}; tid=${rest%%:*}; moe=${rest#*:}
  $CXX $FLAGS -DMRS_TAG=$tag -DMRS_TYPE=$tid -DMRS_MOE_TAG=$moe -c "$OUT/src/mmvq_inst.hip" -o "$OUT/obj/mmvq_$tag.o" &
  pids="$pids $!"
done
$CXX $FLAGS -c "$OUT/src/mmvq_quantize.hip" -o "$OUT/obj/mmvq_quantize.o" & pids="$pids $!"
$CXX $FLAGS -c "$OUT/src/moe.hip" -o "$OUT/obj/moe.o" & pids="$pids $!"
for p in $pids; do wait $p; done
$CXX -shared -o "$OUT/libhiphost_quant.so" "$OUT"/obj/*.o
echo "oracle/_hiphost: built libhiphost_quant.so (product kernels of libmistralrsquant.so on wave64 host fibers)"
