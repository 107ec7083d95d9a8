#!/bin/bash
# experiment build of libmrs_hip_ext.so: profiles/experiments/build_variant.sh <name> <source.hip> "<extra hipcc flags>"  ->  mistral.rs_amd/lib/libmrs_hip_ext_<name>.so
# (selected at run time with MRS_EXT_LIB=libmrs_hip_ext_<name>.so, mistral.rs_amd/_lib.py; run HERE, the .so travels to the GPU box)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../mistral.rs_amd"
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -Icsrc -I../include "$@" -c csrc/$src -o /tmp/${base}_$name.o
objs=""
for o in ext_decode ext_dec ext_dec2 ext_prefetch ext_gemm ext_attn_prefill ext_comm ext_p2p ext_hqq_gemv ext_isq runtime kv_cache_manager; do
  if [ "$o" = "$base" ]; then objs="$objs /tmp/${base}_$name.o"; else objs="$objs csrc/build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmrs_hip_ext_$name.so $objs -Llib -lmistralrsquant -lmistralrspagedattention -lmistralrscuda '-Wl,-rpath,$ORIGIN' -ldl
echo "built lib/libmrs_hip_ext_$name.so"
