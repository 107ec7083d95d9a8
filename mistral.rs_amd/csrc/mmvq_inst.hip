// mmvq_inst.hip -- one translation unit per GGUF type (build passes -DMRS_TAG=q4_k -DMRS_TYPE=12 ...),
// so the 10 types compile in parallel.  Exports launch_mmvq_gguf_<tag>_{f32,f16,bf16}_{plain,fused_glu,fused_qkv}.
#include "mmvq_kernels.cuh"
#ifndef MRS_TAG
#error "compile with -DMRS_TAG=<type tag> -DMRS_TYPE=<ggml type id>"
#endif
MRS_MMVQ_LAUNCHERS(MRS_TAG, MRS_TYPE)
