#!/bin/bash
# experiment builds of libmrs_hip_ext.so with -DMRS_GEMM_ABLATE=<n> (mistral.rs_amd/lib/libmrs_hip_ext_abl<n>.so, selected with MRS_EXT_LIB)
set -e
cd "$(dirname "$0")/../../mistral.rs_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -Icsrc -I../include -DMRS_GEMM_ABLATE=$n -c csrc/ext_gemm.hip -o /tmp/ext_gemm_abl$n.o &
done
wait
for n in "$@"; do
  objs="csrc/build/ext_decode.o csrc/build/ext_dec.o csrc/build/ext_attn_prefill.o csrc/build/ext_comm.o csrc/build/ext_p2p.o csrc/build/ext_hqq_gemv.o csrc/build/ext_isq.o csrc/build/runtime.o csrc/build/kv_cache_manager.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmrs_hip_ext_abl$n.so $objs /tmp/ext_gemm_abl$n.o -Llib -lmistralrsquant -lmistralrspagedattention -lmistralrscuda '-Wl,-rpath,$ORIGIN' -ldl
done
ls -la lib/ | grep abl
