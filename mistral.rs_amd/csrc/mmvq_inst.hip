// mmvq_inst.hip -- one translation unit per GGUF type (build passes -DMRS_TAG=q4_k -DMRS_TYPE=12 ...),
// so the 10 types compile in parallel.  Exports launch_mmvq_gguf_<tag>_{f32,f16,bf16}_{plain,fused_glu,fused_qkv}
// and launch_indexed_moe_forward_<moe tag>_q8_1 (moe tag: q4k for q4_k, ..., q4_0 unchanged).
#include "mmvq_kernels.cuh"
#ifndef MRS_TAG
#error "compile with -DMRS_TAG=<type tag> -DMRS_TYPE=<ggml type id>"
#endif
#ifndef MRS_MOE_ONLY  // Q8_1 weights exist only behind the MoE launchers (gguf/ffi.rs:268,424,601,800)
MRS_MMVQ_LAUNCHERS(MRS_TAG, MRS_TYPE)
#endif
#ifdef MRS_MOE_TAG
MRS_INDEXED_MOE_LAUNCHER(MRS_MOE_TAG, MRS_TYPE)
#endif
