/* oracle/hip_host/mfma_stubs.c -- TEST INFRASTRUCTURE ONLY.  The entry points of ext_gemm.hip / ext_attn_prefill.hip / ext_comm.hip that the
 * C++ runner references directly: MFMA and RCCL are not modelled by the host emulation, so these refuse (-1 / 0 bytes) and the runner
 * reports the failure.  Decode never reaches them. */
#include <stddef.h>
int mrs_comm_all_reduce_sum_f32() { return -1; }
int mrs_convert_f32_bf16_slabs() { return -1; }
int mrs_gemm_q_bf16_multi() { return -1; }
size_t mrs_gemm_q_bf16_workspace_bytes() { return 0; }
int mrs_gemm_q_f32() { return -1; }
int mrs_gemm_q_f32_multi() { return -1; }
int mrs_glu_bf16_slabs() { return -1; }
int mrs_prefill_attention_f32_bf16() { return -1; }
int mrs_rms_norm_bf16_slabs() { return -1; }
