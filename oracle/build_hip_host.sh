#!/bin/sh
# oracle/build_hip_host.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the PRODUCT kernel sources (mistral.rs_amd/csrc: the MMVQ core with every launcher of the 10 GGUF types, the Q8_1 quantizer,
# mmq.hip, moe.hip, gemv.hip, quant_ops / core_ops / hqq / ext_isq, the fused decode kernels of ext_decode.hip, kv_cache_ops and the paged-attention
# instantiations, the MFMA prefill GEMM / attention, the C++ runner and KV manager) for the HOST on top of oracle/hip_host/hip/hip_runtime.h (wave64 fibers), so the C-ABI launchers can be executed and
# compared with the oracle without a GPU (tests/test_hip_host_emulation.py).  Output: oracle/_hiphost/libhiphost.so (git-ignored).
# v_mfma_f32_32x32x16_bf16 and the raw buffer loads of ext_gemm.hip / ext_attn_prefill.hip are modelled too (lane layout calibrated against the
# GPU-green tests).  ext_comm.hip (RCCL) is replaced by hip_host/comm_shim.c: the all-reduce is a callback (gloo in tests/test_distributed.py).
# The sources are copied into oracle/_hiphost/src with TWO textual changes (the second: two wave syncs in decode_attn_wave_kernel, see below); the first: `extern __shared__ ... name[];` (dynamic LDS) becomes a pointer to
# the shim's LDS buffer.  Same flags that pin the arithmetic in the product build: -ffp-contract=off, no fast-math.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
CSRC="$HERE/../mistral.rs_amd/csrc"
OUT="${HIPHOST_OUT:-$HERE/_hiphost}"
CXX="${HIPHOST_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
rm -rf "$OUT/obj"
mkdir -p "$OUT/src" "$OUT/obj"
for f in "$CSRC"/*.cuh "$CSRC"/*.hip; do
  sed -E 's/extern __shared__( __attribute__\(\(aligned\(16\)\)\))? ([A-Za-z_0-9]+) ([A-Za-z_0-9]+)\[\];/\2 *\3 = (\2 *)hiphost::dyn_lds;/' "$f" > "$OUT/src/$(basename "$f")"
done
# second textual change: decode_attn_wave_kernel passes q and the probabilities between the lanes of a wave through LDS without a barrier (lockstep
# execution on the device); fibers are not in lockstep, so a wave-level sync goes in front of the two read sites
sed -i -E 's|^( *)// ---- scores for token t, all G heads|\1hiphost::wave_sync();  // host emulation only|; s|^( *)// ---- P\.V for rows d = lane, lane \+ 64.*|\1hiphost::wave_sync();  // host emulation only|' "$OUT/src/paged_attention.cuh"
grep -c "hiphost::wave_sync" "$OUT/src/paged_attention.cuh" | grep -qx 2 || { echo "build_hip_host.sh: the lockstep markers of paged_attention.cuh moved"; exit 1; }
# same for the per-query rescale factors that prefill_attn_kernel broadcasts through LDS inside a wave (ext_attn_prefill.hip: `bc[wave][ql] = ...`)
sed -i -E 's|^( *if \(kh == 0\) bc\[wave\]\[ql\] = .*;)$|\1 hiphost::wave_sync();  // host emulation only|' "$OUT/src/ext_attn_prefill.hip"
grep -c "hiphost::wave_sync" "$OUT/src/ext_attn_prefill.hip" | grep -qx 2 || { echo "build_hip_host.sh: the lockstep markers of ext_attn_prefill.hip moved"; exit 1; }
FLAGS="-x c++ -std=c++17 -O1 -fPIC -march=native -fno-fast-math -ffp-contract=off -w -I$HERE/hip_host -I$OUT/src -I$HERE/../include"
pids=""
cc() { obj="$1"; src="$2"; shift 2; $CXX $FLAGS "$@" -c "$OUT/src/$src" -o "$OUT/obj/$obj.o" & pids="$pids $!"; }
for spec in q4_0:2:q4_0 q4_1:3:q4_1 q5_0:6:q5_0 q5_1:7:q5_1 q8_0:8:q8_0 q2_k:10:q2k q3_k:11:q3k q4_k:12:q4k q5_k:13:q5k q6_k:14:q6k; do
  tag=${spec%%:*}; rest=${spec#*:}; tid=${rest%%:*}; moe=${rest#*:}
  cc mmvq_$tag mmvq_inst.hip -DMRS_TAG=$tag -DMRS_TYPE=$tid -DMRS_MOE_TAG=$moe
done
for nc in 1 2 3 4 5 6 7 8; do cc ext_dec_gemv_nc$nc ext_dec_gemv.hip -DMRS_DEC_NC=$nc; done
cc mmvq_q8_1 mmvq_inst.hip -DMRS_TAG=q8_1 -DMRS_TYPE=9 -DMRS_MOE_TAG=q8_1 -DMRS_MOE_ONLY  # Q8_1 weights: MoE launchers only
for f in mmvq_quantize mmq moe gemv quant_ops core_ops sampling hqq ext_isq ext_decode ext_dec ext_dec_mm ext_p2p ext_hqq_gemv ext_gemm ext_gemm_qi ext_attn_prefill kv_cache_ops; do cc $f $f.hip; done
# the C++ runner (plain host code: it finds the launchers with dlsym(RTLD_DEFAULT), so it only works in a process that loaded THIS library
# RTLD_GLOBAL and not the product libraries -- `pytest --host-emulation`); the RCCL entry points are hip_host/comm_shim.c
mkdir -p "$OUT/src/host"
cp "$CSRC/host/runtime.cpp" "$CSRC/host/kv_cache_manager.cpp" "$OUT/src/host/"
$CXX $FLAGS -c "$OUT/src/host/kv_cache_manager.cpp" -o "$OUT/obj/kv_cache_manager.o" & pids="$pids $!"
$CXX $FLAGS -I"$HERE/../include" -c "$OUT/src/host/runtime.cpp" -o "$OUT/obj/runtime.o" & pids="$pids $!"
gcc -O1 -fPIC -w -c "$HERE/hip_host/comm_shim.c" -o "$OUT/obj/comm_shim.o"
# paged attention: the instantiations of mistral.rs_amd/build.py
cc pa_f16 paged_attention.hip -DMRS_PA_TAG=f16 -DMRS_PA_T=mrs::f16_t -DMRS_PA_CT=mrs::f16_t -DMRS_PA_EXPORT_ABI
cc pa_bf16 paged_attention.hip -DMRS_PA_TAG=bf16 -DMRS_PA_T=mrs::bf16_t -DMRS_PA_CT=mrs::bf16_t -DMRS_PA_EXPORT_ABI
cc pa_f32 paged_attention.hip -DMRS_PA_TAG=f32 -DMRS_PA_T=float -DMRS_PA_CT=float -DMRS_PA_EXPORT_ABI
cc pa_f32_bf16 paged_attention.hip -DMRS_PA_TAG=f32_bf16 -DMRS_PA_T=float -DMRS_PA_CT=mrs::bf16_t -DMRS_PA_DECODE_Q8_1
cc pa_f16_fp8 paged_attention.hip -DMRS_PA_TAG=f16 -DMRS_PA_T=mrs::f16_t -DMRS_PA_CT=mrs::fp8_t -DMRS_PA_FP8
cc pa_bf16_fp8 paged_attention.hip -DMRS_PA_TAG=bf16 -DMRS_PA_T=mrs::bf16_t -DMRS_PA_CT=mrs::fp8_t -DMRS_PA_FP8
cc pa_f32_fp8 paged_attention.hip -DMRS_PA_TAG=f32 -DMRS_PA_T=float -DMRS_PA_CT=mrs::fp8_t -DMRS_PA_FP8
for p in $pids; do wait $p; done
# -Bsymbolic: inline / template symbols shared with the product libraries (loaded RTLD_GLOBAL in the same process) must bind to THIS library
$CXX -shared -Wl,-Bsymbolic -o "$OUT/libhiphost.so" "$OUT"/obj/*.o
rm -f "$OUT/libhiphost_quant.so"
echo "oracle/_hiphost: built libhiphost.so (product kernels on wave64 host fibers)"
